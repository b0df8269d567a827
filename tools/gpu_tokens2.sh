#!/bin/bash
# the runner's own token rule (--rpn-tokens 0) against one / two tokens on the legs that serialise their RPN segments
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r06_tok2}; mkdir -p $O
for rep in 1 2; do for t in 0 1 2; do
for wl in "--workload car.fhd" "--scene dense" "--workload nusc.fhd"; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs --rpn-tokens $t $wl > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d = json.load(open("$O/b.json")); print("tokens $t $wl: %.0f frames/s  %.4f ms/step  (runner: %s)" % (d["value"], d["ms_per_step"], d["config"].get("rpn_tokens")))
except Exception as e:
    print("tokens $t $wl: FAILED", e, open("$O/b.err").read()[-600:])
PY
done; done; done
