#!/bin/bash
# kernel statistics of the captured nuscenes/all.fhd training step (fp16): last 20 steps
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r04_nusc_train}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace -d $O/prof -- python $R/bench.py --profile-run --workload nusc.fhd.train --steps 30 --warmup 5 > $O/prof.log 2>&1
cd $R
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocprof_summary.py $db --last-steps 20 --marker k_vox_init > $O/kernel_stats_nusc_train_graph.txt 2>&1
rm -rf $O/prof; head -50 $O/kernel_stats_nusc_train_graph.txt | cut -c1-90,111-170
