#!/bin/bash
export PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/${1:-r03_j}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_rows.py tests/test_gpu_e2e.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "voxel or pillar" > $O/pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -2 $O/pytest2.log
for b in 0 1; do
  echo "== SEC_CONV_BAL=$b"; SEC_CONV_BAL=$b timeout 300 python tools/conv_microbench.py --all-layers --variants 1 --iters 100 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee $O/microbench_all_layers_bal$b.txt
done
for i in 1 2; do
  for b in 0 1; do
    SEC_CONV_BAL=$b timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-table --no-extra-lines 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('BAL=$b value', d['value'], 'lat', d['config']['single_step_latency_ms'], 'roof', d['roofline']['launch_us'], d['roofline']['frac'], d['roofline']['kernel'][:70])"
  done
done 2>&1 | tee $O/bench_ab.txt
