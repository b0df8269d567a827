#!/bin/bash
# nusc.fhd (config 5) and nusc.pp (config 4) bench legs with the strided / patch conv kernel on and off (SEC_CONV2D_PATCH), same box
R=$PWD; O=$R/gpurun_out/${1:-nusc_ab}; mkdir -p $O
for w in nusc.fhd nusc.pp; do for v in 1 0 1 0; do
  SEC_CONV2D_PATCH=$v timeout 600 python bench.py --workload $w --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-extra-lines --no-kernel-table > $O/b_${w}_$v.json 2> $O/b.err
  python - <<PY
import json
d=json.load(open("$O/b_${w}_$v.json")); print("$w patch=$v", d["value"], d["ms_per_step"], d["config"].get("single_step_latency_ms"))
PY
done; done
