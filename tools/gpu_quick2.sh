#!/bin/bash
# rulebook / voxelize / NMS parity tests + bench lines over SEC_NMS_TILE_ROWS x SEC_NMS_WGS
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_v}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for CFG in "16 0" "16 64" "16 256" "8 0" "64 0"; do
  set -- $CFG
  echo "== SEC_NMS_TILE_ROWS=$1 SEC_NMS_WGS=$2"
  SEC_NMS_TILE_ROWS=$1 SEC_NMS_WGS=$2 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-lines --no-other-configs 2>$O/bench.err > $O/bench_$1_$2.json
  python - <<PY
import json
d=json.load(open("$O/bench_$1_$2.json")); c=d['config']
print(d['value'], d['ms_per_step'], 'single', c.get('single_step_latency_ms'), 'dets', d.get('detections_last_step'))
print([(k['op'],k['us']) for k in d['kernels'] if k['op'] in ('voxelize','nms_sorted','predict_select','rulebook_subm','rulebook_conv')])
PY
done
