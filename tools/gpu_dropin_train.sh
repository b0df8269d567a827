#!/bin/bash
# The training drop-in on the GPU: its tests, then the bench leg alone.
TAG=${1:-r06_b}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dropin_train.py -m gpu -q -x -s > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -40 $O/pytest.log
timeout 600 python - > $O/dropin_train.json 2> $O/dropin_train.err <<'PY'
import json, sys
sys.argv = ["bench.py"]
import bench
import torch
print(json.dumps(bench.time_dropin_train(), indent=1))
PY
echo "leg rc=$?"; cat $O/dropin_train.json; tail -5 $O/dropin_train.err
