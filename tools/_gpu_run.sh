mkdir -p gpurun_out/r2g/prof1 gpurun_out/r2g/prof3
rm -f gpurun_out/r2g/prof1/* gpurun_out/r2g/prof3/*
export PYTHONUNBUFFERED=1
R=$PWD
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2g/pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/r2g/pytest_all.log
tail -2 gpurun_out/r2g/pytest_all.log
LANES=3 timeout 900 python tools/inflight_stress.py 2000 3 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python bench.py > gpurun_out/r2g/bench_default.json 2> gpurun_out/r2g/bench_default.err
timeout 300 python bench.py --inflight 1 --no-kernel-table --no-cpu-baseline > gpurun_out/r2g/bench_inflight1.json 2>/dev/null
for wl in nusc.pp nusc.fhd; do timeout 600 python bench.py --workload $wl --no-kernel-table > gpurun_out/r2g/bench_$wl.json 2>/dev/null; done
timeout 600 python bench.py --workload car.fhd.train --steps 30 --warmup 5 > gpurun_out/r2g/bench_train_car_fp32.json 2>/dev/null
timeout 600 python bench.py --workload car.fhd.train --dtype bf16 --steps 30 --warmup 5 > gpurun_out/r2g/bench_train_car_bf16.json 2>/dev/null
timeout 600 python bench.py --workload nusc.fhd.train --steps 20 --warmup 5 > gpurun_out/r2g/bench_train_nusc_fhd_fp16.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2g/prof1 -o t -- python $R/bench.py --steps 50 --warmup 10 --inflight 1 --no-kernel-table --no-cpu-baseline > $R/gpurun_out/r2g/prof1/bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2g/prof3 -o t -- python $R/bench.py --steps 50 --warmup 10 --no-kernel-table --no-cpu-baseline > $R/gpurun_out/r2g/prof3/bench.log 2>&1
cd $R
for d in prof1 prof3; do DB=$(find gpurun_out/r2g/$d -name "*.db" | head -1); python tools/rocprof_summary.py $DB --steps 60 > gpurun_out/r2g/$d/summary.txt; find gpurun_out/r2g/$d -name "*.db" -delete; done
for f in gpurun_out/r2g/bench_*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d.get('roofline') and d['roofline'].get('frac'), d['config'].get('single_step_latency_ms'))
PY
done
