export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -x -k "gathered or dense_image or detector" 2>&1 | tail -3
for g in 1 1; do SEC_RPN_GATHER=$g timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('gather $g', d['value'], d['ms_per_step'], d['config']['single_step_latency_ms'], [ (k['op'],k['us']) for k in d['kernels'] if k['op'] in ('sparse_to_dense','sparse_site_map','conv2d_nhwc_gather') or k['op']=='conv2d_nhwc'][:3])"; done
