mkdir -p gpurun_out/r2z
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_train.py -m gpu -q -x -k "sorted or voxelize or detector or e2e or trainer or numbering" 2>&1 | tail -6
timeout 300 python tools/rulebook_microbench.py --numbering sorted 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight3', d['value'], d['ms_per_step'], d['config'].get('single_step_latency_ms'))"
timeout 300 python bench.py --inflight 1 --no-kernel-table --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight1', d['value'], d['ms_per_step'])"
