#!/bin/bash
# RPN conv occupancy (3 vs 2 workgroups per CU) against throughput with three steps in flight; select-chunk index stepping
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_z}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "predict or select" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for PAD in 0 12288 0 12288; do
  for INF in 3 4; do
  echo "== SEC_CONV2D_LDS_PAD=$PAD inflight $INF"
  SEC_CONV2D_LDS_PAD=$PAD timeout 300 python bench.py --steps 300 --warmup 30 --inflight $INF --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs 2>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(d['value'], d['ms_per_step'], 'single', c.get('single_step_latency_ms'), 'rpn probe us', d['roofline_mfma']['launch_us'])"
  done
done
