#!/bin/bash
# kernel statistics of the CAPTURED training step (car.fhd.train bf16, one hipGraph per step): last 20 steps
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r04_train}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/prof -- python $R/bench.py --profile-run --workload car.fhd.train --dtype bf16 --steps 30 --warmup 5 > $O/prof.log 2>&1
cd $R
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocprof_summary.py $db --last-steps 20 --marker k_vox_init > $O/kernel_stats_train_graph.txt 2>&1
python tools/rocprof_summary.py $db --timeline k_vox_init > $O/step_timeline_train_graph.txt 2>&1
rm -rf $O/prof; head -60 $O/kernel_stats_train_graph.txt | cut -c1-90,111-170
