export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_round2.py -m gpu -q -x -k "predict or tie or detector or e2e or in_flight or nms or nuscenes or pointpillars" 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('inflight3', d['value'], d['ms_per_step'], d['config']['single_step_latency_ms'], [ (k['op'],k['us']) for k in d['kernels'] if k['op'].startswith('predict') or k['op']=='nms_sorted'])"
