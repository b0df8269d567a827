export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_conv_rows.py tests/test_gpu_parity.py -m gpu -q -x -k "conv or detector" 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['single_step_latency_ms'])
print([(k['us'], k['kernel']) for k in d['kernels'] if k['op']=='indice_conv'])"
