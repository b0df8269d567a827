#!/bin/bash
# launch-by-launch timeline of ONE replayed step (inflight 1): where the single-step latency goes
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_r}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/prof1 -- python $R/bench.py --profile-run --steps 30 --warmup 10 --inflight 1 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs > $O/prof1.log 2>&1
cd $R
db=$(find $O/prof1 -name "*.db" | head -1); python tools/rocprof_summary.py $db --timeline k_vox_init > $O/step_timeline.txt 2>&1
rm -rf $O/prof1; cat $O/step_timeline.txt | cut -c1-130
