#!/bin/bash
# PMC passes over tools/pp_conv_microbench.py (every conv layer shape of the PointPillars RPN, eager launches) -> gpurun_out/<tag>/pmc_pp_conv.txt
TAG=${1:-pp_pmc}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
: > $O/pmc_pp_conv.txt
for SET in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  GRAPH=0 timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc$i -- python $R/tools/pp_conv_microbench.py > $O/pmc$i.log 2>&1
  echo "== $SET" >> $O/pmc_pp_conv.txt
  python $R/tools/pmc_summary.py $(dirname $(find $O/pmc$i -name "*counter_collection.csv" | head -1)) k_conv2d >> $O/pmc_pp_conv.txt 2>&1
  rm -rf $O/pmc$i
done
cat $O/pmc_pp_conv.txt
