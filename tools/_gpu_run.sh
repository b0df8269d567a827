mkdir -p gpurun_out/r2z
export PYTHONUNBUFFERED=1
for n in 2 3 4 5; do timeout 300 python bench.py --inflight $n --no-kernel-table --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight', $n, d['value'], d['ms_per_step'], d['config'].get('single_step_latency_ms'))"; done
timeout 300 python bench.py --inflight 1 --branches 2 --no-kernel-table --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight 1 branches 2', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --inflight 3 --branches 2 --no-kernel-table --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight 3 branches 2', d['value'], d['ms_per_step'])"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "tie_ranking" 2>&1 | tail -1
