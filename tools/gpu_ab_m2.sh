#!/bin/bash
# A/B of the two-tiles-per-wave sparse conv (variant 41 / SEC_CONV_M2=1) against the shipped form, parity first.
export PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/${1:-r03_b}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv_rows.py tests/test_gpu_e2e.py -m gpu -q -x -s > $O/pytest_m2.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_m2.log
timeout 300 python tools/conv_microbench.py --layer subm2 --variants 22,41,22,41 --iters 200 2>&1 | tee $O/microbench_subm2.txt
for i in 1 2; do
  for m in 0 1; do
    SEC_CONV_M2=$m timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-table --no-extra-lines 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('M2=$m value', d['value'], 'lat', d['config']['single_step_latency_ms'], 'roof', d['roofline']['launch_us'], d['roofline']['frac'], d['roofline']['kernel'][:60])"
  done
done 2>&1 | tee $O/bench_ab.txt
SERIES_OUT=$O/power_series.json timeout 300 python tools/rpn_yardstick.py > $O/rpn_yardstick.txt 2>&1; echo "yardstick rc=$?"; cut -c1-400 $O/rpn_yardstick.txt
