#!/bin/bash
export PYTHONUNBUFFERED=1
R=$PWD
O=$PWD/gpurun_out/${1:-r03_h}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "voxel or pillar or pointpillars or nuscenes" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_pp -- python $R/bench.py --workload nusc.pp --steps 20 --warmup 5 --inflight 1 --no-kernel-table > $O/prof_pp.log 2>&1
cd $R
db=$(find $O/prof_pp -name "*.db" | head -1); python tools/rocprof_summary.py $db --steps 25 > $O/kernel_stats_nusc_pp.txt 2>&1
rm -rf $O/prof_pp; head -22 $O/kernel_stats_nusc_pp.txt | cut -c1-100,110-175
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-lines 2>/dev/null > $O/bench_nms.json; python -c "
import json
d=json.load(open('$O/bench_nms.json')); k=d.pop('kernels')
print(d['value'], d['config']['single_step_latency_ms'], d['roofline']['launch_us'])
for e in k:
    if e['op'] in ('nms_sorted','predict_select','voxelize'): print(e)"
timeout 300 python bench.py --workload nusc.pp --steps 50 --warmup 10 --no-kernel-table 2>/dev/null | cut -c1-700
