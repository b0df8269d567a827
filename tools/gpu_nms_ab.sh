#!/bin/bash
# NMS mask tiling A/B (SEC_NMS_TILE_ROWS = 64 | 16 | 8) on the bench candidates + the NMS parity tests
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_t}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py -m gpu -q -x -k "nms or predict or e2e or bf16 or detector" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for T in 64 16 8; do
  echo "== SEC_NMS_TILE_ROWS=$T"
  SEC_NMS_TILE_ROWS=$T timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-lines --no-other-configs 2>$O/t$T.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(d['value'], d['ms_per_step'], 'single', c.get('single_step_latency_ms'), [ (k['op'],k['us']) for k in d['kernels'] if k['op'] in ('nms_sorted','predict_select')], 'dets', d.get('detections_last_step'))"
done
