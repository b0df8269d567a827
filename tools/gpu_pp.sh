#!/bin/bash
# PointPillars (config 4) checks: conv parity tests, detector tests, the bench leg, its one-step timeline
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r04_pp}; mkdir -p $O
timeout 900 python -m pytest tests -q -x -m gpu -k "conv2d_nhwc or pointpillars or pillar or pfn" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python bench.py --workload nusc.pp --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_pp.json 2> $O/bench_pp.err; echo "pp rc=$?"; cut -c1-200 $O/bench_pp.json
python - <<PY
import json
d=json.load(open("$O/bench_pp.json"))
print("single", d["config"]["single_step_latency_ms"]); print([(k["op"],k["us"],k.get("detail","")[:28]) for k in d["kernels"] if "conv2d" in k["op"]])
PY
