#!/bin/bash
# occupancy curve of the RPN on reachable tiles + training checks after the deferred BatchNorm counters
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r04_misc}; mkdir -p $O
timeout 600 python tools/rpn_tiles_density.py > $O/rpn_tiles_density.txt 2>&1; echo "density rc=$?"; cat $O/rpn_tiles_density.txt
timeout 900 python -m pytest tests/test_gpu_rpn_tiles.py tests/test_gpu_train_dense.py -q -x -m gpu > $O/pytest_train.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_train.log
timeout 600 python bench.py --workload car.fhd.train --dtype bf16 --gpus 1 --steps 20 --warmup 5 > $O/bench_train.json 2> $O/bench_train.err; echo "train rc=$?"; cut -c1-300 $O/bench_train.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-kernel-table --no-extra-lines > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
