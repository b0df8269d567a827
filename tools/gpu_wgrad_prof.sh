#!/bin/bash
# rocprofv3 kernel statistics of tools/wgrad_probe.py (dense weight-gradient kernel + reduce alone)
TAG=${1:-wgrad}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof1 -- python $R/tools/wgrad_probe.py > $O/probe.log 2>&1
cd $R
db=$(find $O/prof1 -name "*.db" | head -1); python tools/rocprof_summary.py $db > $O/kernel_stats_wgrad_probe.txt 2>&1
rm -rf $O/prof1; tail -4 $O/probe.log; grep -E "wgrad|kernel  " $O/kernel_stats_wgrad_probe.txt | cut -c1-60,110-175
