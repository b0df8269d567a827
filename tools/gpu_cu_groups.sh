#!/bin/bash
# EXPERIMENT: lanes of the serving loop on CU-masked streams (SEC_LANE_CU_GROUPS=g: lane k runs on slice k % g of every XCD's CUs).
#   gpurun --timeout 900 -- 'bash tools/gpu_cu_groups.sh r06_cug'
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r06_cug}; mkdir -p $O
run() {   # groups inflight serialize tokens [workload]
  SEC_LANE_CU_GROUPS=$1 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs \
      --inflight $2 --serialize-rpn $3 --rpn-tokens $4 --workload ${5:-car.fhd} > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d = json.load(open("$O/b.json")); print("groups $1 inflight $2 serialize $3 tokens $4 ${5:-car.fhd}: %.0f frames/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("groups $1 inflight $2 serialize $3 tokens $4: FAILED", e); print(open("$O/b.err").read()[-600:])
PY
}
run 0 4 1 1
run 4 4 0 1
run 4 4 1 1
run 2 4 0 1
run 2 4 1 2
run 4 8 0 1
run 8 8 0 1
run 2 2 0 1
run 0 4 1 1
