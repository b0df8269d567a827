#!/bin/bash
# background-tile RPN: parity tests + A/B of the bench line (skip on / off) on one box
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r04_tiles}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rpn_tiles.py tests/test_gpu_round2.py -q -x -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for S in 1 0; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --background-skip $S --no-cpu-baseline --no-other-configs > $O/bench_skip$S.json 2> $O/bench_skip$S.err; echo "bench skip=$S rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_skip$S.json"))
print("skip=$S value",d["value"],"ms",d["ms_per_step"],"single",d["config"]["single_step_latency_ms"],"live",d["config"].get("rpn_background_tiles"))
print([ (k["op"],k.get("us")) for k in d["kernels"] if "conv" in k["op"] or "tile" in k["op"]])
PY
done
