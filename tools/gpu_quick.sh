#!/bin/bash
# full -m gpu suite + one bench line with the per-op table (no CPU baseline, no extra lines)
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_u}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for T in ${2:-8 16}; do
  echo "== SEC_NMS_TILE_ROWS=$T"
  SEC_NMS_TILE_ROWS=$T timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-lines --no-other-configs 2>$O/bench$T.err > $O/bench$T.json
  python - <<PY
import json
d=json.load(open("$O/bench$T.json")); c=d['config']
print(d['value'], d['ms_per_step'], 'single', c.get('single_step_latency_ms'), 'dets', d.get('detections_last_step'))
print([(k['op'],k['us']) for k in d['kernels'] if k['op'] in ('voxelize','nms_sorted','predict_select','rulebook_subm','rulebook_conv')])
PY
done
