#!/bin/bash
# kernel stats of the config-3 training step (and of the PointPillars one) as shipped
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_aa}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for WL in car.fhd.train; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$WL -- python $R/bench.py --workload $WL --dtype bf16 --steps 40 --warmup 10 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs > $O/prof_$WL.log 2>&1
  db=$(find $O/prof_$WL -name "*.db" | head -1); python $R/tools/rocprof_summary.py $db --last-steps 20 --marker k_vox_init > $O/kernel_stats_$WL.txt 2>&1
  rm -rf $O/prof_$WL; grep "^{" $O/prof_$WL.log | cut -c1-200; head -50 $O/kernel_stats_$WL.txt | cut -c1-90,110-175
done
