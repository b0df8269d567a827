#!/bin/bash
# PointPillars (config 4): conv / pillar parity tests, bench leg with its kernel table, rocprofv3 kernel statistics / one-step timeline
TAG=${1:-pp_round}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -q -x -m gpu -k "conv2d_nhwc or pointpillars or pillar or pfn" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python bench.py --workload nusc.pp --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-extra-lines > $O/bench_pp.json 2> $O/bench_pp.err; echo "pp rc=$?"; cut -c1-300 $O/bench_pp.json
python - <<PY
import json
d=json.load(open("$O/bench_pp.json"))
print("single", d["config"]["single_step_latency_ms"])
for k in d["kernels"]:
    print(k["op"], k["us"], k.get("detail","")[:40], k.get("frac"))
PY
bash tools/gpu_pp_prof.sh $TAG
