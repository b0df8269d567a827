mkdir -p gpurun_out/r2e/prof1 gpurun_out/r2e/prof3 gpurun_out/r2e/pmc
rm -f gpurun_out/r2e/prof1/* gpurun_out/r2e/prof3/*
export PYTHONUNBUFFERED=1
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2e/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/r2e/pytest_all.log
tail -2 gpurun_out/r2e/pytest_all.log
timeout 900 python bench.py > gpurun_out/r2e/bench_default.json 2> gpurun_out/r2e/bench_default.err
timeout 300 python bench.py --inflight 1 --no-kernel-table --no-cpu-baseline > gpurun_out/r2e/bench_inflight1.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2e/prof1 -o t -- python $R/bench.py --steps 50 --warmup 10 --inflight 1 --no-kernel-table --no-cpu-baseline > $R/gpurun_out/r2e/prof1/bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2e/prof3 -o t -- python $R/bench.py --steps 50 --warmup 10 --no-kernel-table --no-cpu-baseline > $R/gpurun_out/r2e/prof3/bench.log 2>&1
cd $R
for d in prof1 prof3; do DB=$(find gpurun_out/r2e/$d -name "*.db" | head -1); python tools/rocprof_summary.py $DB --steps 60 > gpurun_out/r2e/$d/summary.txt; find gpurun_out/r2e/$d -name "*.db" -delete; done
for f in gpurun_out/r2e/bench_default.json gpurun_out/r2e/bench_inflight1.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_us'], d['config'].get('single_step_latency_ms'), d.get('cpu_baseline') and d['cpu_baseline'].get('value'), d.get('roofline_mfma') and d['roofline_mfma']['frac'])
PY
done
grep "k_conv_rows_buf<__hip_bfloat16, 64, 64" gpurun_out/r2e/prof1/summary.txt | cut -c1-170
