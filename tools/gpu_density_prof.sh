#!/bin/bash
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r04_dp}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for F in skip dense; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$F -- python $R/tools/rpn_tiles_density.py 1.0 $F > $O/prof_$F.log 2>&1
  db=$(find $O/prof_$F -name "*.db" | head -1); python $R/tools/rocprof_summary.py $db --steps 1 2>&1 | head -14 | cut -c1-90,100-170 > $O/stats_$F.txt
  echo "== $F"; cat $O/stats_$F.txt; rm -rf $O/prof_$F
done
