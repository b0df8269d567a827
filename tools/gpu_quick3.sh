#!/bin/bash
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_w}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for S in 1 0; do
  echo "== SEC_SELECT_THRESHOLD_SHORTCUT=$S"
  SEC_SELECT_THRESHOLD_SHORTCUT=$S timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-lines --no-other-configs 2>$O/bench.err > $O/bench_$S.json
  python - <<PY
import json
d=json.load(open("$O/bench_$S.json")); c=d['config']
print(d['value'], d['ms_per_step'], 'single', c.get('single_step_latency_ms'), 'dets', d.get('detections_last_step'), d.get('detections_match_cpu'))
print([(k['op'],k['us']) for k in d['kernels'] if k['op'] in ('voxelize','nms_sorted','predict_select','predict_decode')])
PY
done
