export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 600 python tools/inflight_stress.py 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_i3.json 2> gpurun_out/bench_i3.err; tail -c 600 gpurun_out/bench_i3.err
python -c "
import json; d=json.load(open('gpurun_out/bench_i3.json')); print('inflight3', d['value'], d['ms_per_step'], d['config']['single_step_latency_ms'], d['roofline'], d['cpu_baseline'])"
