#!/bin/bash
TAG=${1:-r06_h}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_rows.py tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_chain.py tests/test_gpu_dropin_fused.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-lines --no-other-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("single_step_latency_ms"))
s=0
for k in d["kernels"]:
    if k["op"].startswith("indice_conv"): s+=k["us"]
    print(f'{k["op"][:26]:26s} {k["us"]:8.2f} {k.get("frac",0):.3f} {k.get("detail","")[:70]}')
print("sum indice_conv", s)
PY
