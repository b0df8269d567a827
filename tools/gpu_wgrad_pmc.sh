#!/bin/bash
# PMC passes over tools/wgrad_probe.py (the dense weight-gradient kernel alone) -> gpurun_out/<tag>/pmc_wgrad.txt
TAG=${1:-wgpmc}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
: > $O/pmc_wgrad.txt
for SET in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM" \
           "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  WGRAD_ONLY=1 timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc$i -- python $R/tools/wgrad_probe.py > $O/pmc$i.log 2>&1
  echo "== $SET" >> $O/pmc_wgrad.txt
  python $R/tools/pmc_summary.py $(dirname $(find $O/pmc$i -name "*counter_collection.csv" | head -1)) wgrad3x3 >> $O/pmc_wgrad.txt 2>&1
  rm -rf $O/pmc$i
done
cat $O/pmc_wgrad.txt
