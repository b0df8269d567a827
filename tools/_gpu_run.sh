mkdir -p gpurun_out/r2g
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2g/pytest_all.log 2>&1; echo "pytest_all rc=$?" >> gpurun_out/r2g/pytest_all.log
timeout 600 python bench.py > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err
timeout 300 python bench.py --inflight 1 --no-cpu-baseline --no-kernel-table > gpurun_out/r2g/bench_inflight1.json 2>> gpurun_out/r2g/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --inflight 1 --no-cpu-baseline --no-kernel-table > $GRAFT_REPO_ROOT/gpurun_out/r2g/rocprof_bench.log 2>&1
find /tmp/prof1 -name "*kernel_stats*" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r2g/ \;
cd $GRAFT_REPO_ROOT
tail -15 gpurun_out/r2g/pytest_all.log; python - <<'PY'
import json
for f in ("bench.json","bench_inflight1.json"):
    try:
        d=json.loads(open("gpurun_out/r2g/"+f).read().strip().splitlines()[-1])
        k=d.pop("kernels",None)
        print(f, json.dumps(d)[:3000])
        if k:
            for e in k: print("   ", e)
    except Exception as ex: print(f, "ERR", ex)
PY
tail -5 gpurun_out/r2g/bench.err; ls gpurun_out/r2g
