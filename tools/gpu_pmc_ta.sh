#!/bin/bash
# one PMC pass of tools/pmc_workload.py (every sparse conv layer + the RPN conv): texture-addresser busy + L1 request counters
TAG=${1:-pmc_ta}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
: > $O/pmc_ta.txt
for SET in "GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max TCP_TCC_READ_REQ_sum"; do
  PMC_META=$O/pmc_meta.json timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc -- python $R/tools/pmc_workload.py > $O/pmc.log 2>&1
  echo "== $SET" >> $O/pmc_ta.txt
  python $R/tools/pmc_summary.py $(dirname $(find $O/pmc -name "*counter_collection.csv" | head -1)) k_conv >> $O/pmc_ta.txt 2>&1
  rm -rf $O/pmc
done
grep -v "^  GRBM\|effective clock" $O/pmc_ta.txt | cut -c1-150
