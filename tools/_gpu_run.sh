mkdir -p gpurun_out/r2z
export PYTHONUNBUFFERED=1
timeout 600 python bench.py --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight3', d['value'], d['ms_per_step'], d['config'].get('single_step_latency_ms'))"
timeout 300 python bench.py --inflight 1 --no-kernel-table --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight1', d['value'], d['ms_per_step'])"
SEC_RULEBOOK_NUMBERING=first_touch timeout 300 python bench.py --inflight 1 --no-kernel-table --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight1 first_touch', d['value'], d['ms_per_step'])"
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2z/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/r2z/pytest_all.log
tail -5 gpurun_out/r2z/pytest_all.log
