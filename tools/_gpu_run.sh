export PYTHONUNBUFFERED=1
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no vectorisers  inflight3', d['value'], d['ms_per_step'], d['config'].get('single_step_latency_ms'))"
SEC_HIP_LIB=$PWD/second.pytorch_amd/lib/libsecond_hip_slp.so timeout 600 python bench.py --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('with vectorisers inflight3', d['value'], d['ms_per_step'], d['config'].get('single_step_latency_ms'))"
done
