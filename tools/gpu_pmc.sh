#!/bin/bash
# PMC passes of tools/pmc_workload.py (every sparse conv layer + the RPN conv) -> gpurun_out/<tag>/<tag>_pmc.txt, _traffic.json
TAG=${1:-r04_pmc}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for SET in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  PMC_META=$O/pmc_meta.json timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc$i -- python $R/tools/pmc_workload.py > $O/pmc$i.log 2>&1
done
cd $R
python tools/pmc_report.py $O/pmc_meta.json $O/$TAG $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 > $O/pmc_report.log 2>&1
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5
tail -5 $O/pmc1.log; grep -c "==" $O/${TAG}_pmc.txt
