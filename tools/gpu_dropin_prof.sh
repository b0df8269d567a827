#!/bin/bash
# rocprofv3 kernel statistics of the accelerated drop-in call (bf16, synchronous): which launches a net(example) call is made of
TAG=${1:-r06_o}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/tools/dropin_profile.py bf16 60 > $O/prof.log 2>&1
cd $R
db=$(find $O/prof -name "*.db" | head -1); python tools/rocprof_summary.py $db --last-steps 40 --marker k_chain_prep > $O/kernel_stats_dropin_bf16.txt 2>&1 || python tools/rocprof_summary.py $db > $O/kernel_stats_dropin_bf16.txt 2>&1
rm -rf $O/prof; head -75 $O/kernel_stats_dropin_bf16.txt | cut -c1-150
