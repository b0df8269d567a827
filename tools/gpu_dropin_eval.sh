#!/bin/bash
TAG=${1:-r06_e}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dropin_fused.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -v "^E   \|amdgpu.ids" $O/pytest.log | tail -40
timeout 600 python - > $O/dropin_fused.json 2> $O/dropin_fused.err <<'PY'
import json, sys
sys.argv = ["bench.py"]
import bench, torch
from second_amd import synthetic as syn
dev = torch.device("cuda")
clouds, points, offsets = bench.build_inputs(0, dev)
det, cpu_state = bench.build_detector(dev, torch.bfloat16, syn.syn_kitti_cloud(0))
print(json.dumps(bench.time_dropin_fused(cpu_state, points, offsets), indent=1))
PY
echo "leg rc=$?"; cat $O/dropin_fused.json; tail -5 $O/dropin_fused.err
