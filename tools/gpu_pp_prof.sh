#!/bin/bash
# rocprofv3 kernel statistics + one-step timeline of the PointPillars (config 4) inference step, one step in flight
TAG=${1:-pp_prof}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof1 -- python $R/bench.py --profile-run --workload nusc.pp --steps 30 --warmup 10 --inflight 1 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs > $O/prof1.log 2>&1
cd $R
db=$(find $O/prof1 -name "*.db" | head -1); python tools/rocprof_summary.py $db --steps 40 > $O/kernel_stats_nusc_pp_inflight1.txt 2>&1
python tools/rocprof_summary.py $db --timeline k_vox_init > $O/step_timeline_nusc_pp_inflight1.txt 2>&1
rm -rf $O/prof1; head -30 $O/step_timeline_nusc_pp_inflight1.txt | cut -c1-130; tail -2 $O/step_timeline_nusc_pp_inflight1.txt
