#!/bin/bash
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_x}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-lines --no-other-configs 2>$O/bench.err > $O/bench.json
python - <<PY
import json
d=json.load(open("$O/bench.json")); c=d['config']
print(d['value'], d['ms_per_step'], 'single', c.get('single_step_latency_ms'), 'dets', d.get('detections_last_step'))
print([(k['op'],k['us']) for k in d['kernels'] if not k['op'].startswith(('indice_conv','conv2d'))])
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/prof1 -- python $R/bench.py --steps 30 --warmup 10 --inflight 1 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs > $O/prof1.log 2>&1
cd $R
db=$(find $O/prof1 -name "*.db" | head -1); python tools/rocprof_summary.py $db --timeline k_vox_init > $O/step_timeline.txt 2>&1
rm -rf $O/prof1; cat $O/step_timeline.txt | cut -c1-120
