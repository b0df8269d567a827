#!/bin/bash
# rocprofv3 kernel trace of the bench command (one step in flight) -> per-kernel statistics + one step launch by launch
TAG=${1:-r04_bs}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 60 rocprofv3 --kernel-trace --stats -d $O/prof1 -- python $R/bench.py --profile-run --steps 50 --warmup 10 --inflight 1 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs > $O/prof1.log 2>&1
cd $R
db=$(find $O/prof1 -name "*.db" | head -1); python tools/rocprof_summary.py $db --last-steps 40 --marker k_vox_init > $O/kernel_stats_bench_bs8_inflight1.txt 2>&1
python tools/rocprof_summary.py $db --timeline k_vox_init > $O/step_timeline_inflight1.txt 2>&1
rm -rf $O/prof1; head -14 $O/kernel_stats_bench_bs8_inflight1.txt | cut -c1-70,110-175; tail -3 $O/step_timeline_inflight1.txt
