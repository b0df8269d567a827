export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_properties.py tests/test_gpu_e2e.py -m gpu -q -x -k "sorted or detector or rulebook or e2e or laws" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --inflight 1 --no-kernel-table --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight1', d['value'], d['ms_per_step'])"; done
timeout 600 python bench.py --no-kernel-table --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight3', d['value'], d['ms_per_step'])"
