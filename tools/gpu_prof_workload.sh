#!/bin/bash
# kernel statistics of one inference workload of bench.py, one step at a time: bash tools/gpu_prof_workload.sh <tag> <workload>
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r04_w}; W=${2:-nusc.pp}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/prof -- python $R/bench.py --profile-run --workload $W --steps 30 --warmup 5 --inflight 1 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs > $O/prof.log 2>&1
cd $R
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocprof_summary.py $db --timeline k_vox_init > $O/step_timeline_$W.txt 2>&1
rm -rf $O/prof; cat $O/step_timeline_$W.txt | cut -c1-150
