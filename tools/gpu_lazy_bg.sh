#!/bin/bash
# lazy RPN background (sec_conv2d_nhwc_tiles_lazy): parity tests, then the bench line with the copies (0) and without (1), interleaved twice
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r04_bo}; mkdir -p $O
if [ "$2" != "skip-tests" ]; then
  timeout 200 python -m pytest tests/test_gpu_rpn_tiles.py -q -x > $O/pytest_tiles.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_tiles.log | cut -c1-220
fi
for rep in 1; do for lz in 0 1; do
  timeout 120 python bench.py --lazy-background $lz --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-kernel-table --no-extra-lines > $O/bench_lazy${lz}_$rep.json 2> $O/bench_lazy${lz}_$rep.err
  echo "lazy=$lz rep=$rep rc=$? $(python -c "import json;d=json.load(open('$O/bench_lazy${lz}_$rep.json'));print(d['value'], d['ms_per_step'], d['config']['single_step_latency_ms'], d['detections_last_step'])" 2>&1 | tail -1)"
done; done
