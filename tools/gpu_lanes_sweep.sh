#!/bin/bash
# lanes x RPN tokens of the serving loop on the final build (bench line only), interleaved repetitions
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r06_lanes2}; mkdir -p $O
run() {
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs --inflight $1 --serialize-rpn $2 --rpn-tokens $3 > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d = json.load(open("$O/b.json")); print("inflight $1 serialize $2 tokens $3: %.0f frames/s  %.4f ms/step  spread %.1f %%" % (d["value"], d["ms_per_step"], d["timing"]["spread_pct"]))
except Exception as e:
    print("inflight $1 serialize $2 tokens $3: FAILED", e)
PY
}
for rep in 1 2 3; do
run 4 1 1
run 4 1 2
run 4 1 3
run 4 0 1
done
