#!/bin/bash
# training-side changes: parity tests, the captured bf16 step, its kernel statistics
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r04_tr}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_dense.py tests/test_gpu_train.py tests/test_gpu_parity.py -q -x -m gpu -k "bn or train or backward or trainer or step or wgrad" > $O/pytest_train.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_train.log
for W in car.fhd.train; do
timeout 600 python bench.py --workload $W --dtype bf16 --gpus 1 --steps 20 --warmup 5 > $O/bench_$W.json 2> $O/bench_$W.err; echo "$W rc=$?"; cut -c1-220 $O/bench_$W.json
done
timeout 600 python bench.py --workload nusc.fhd.train --gpus 1 --steps 20 --warmup 5 > $O/bench_nusc.json 2> $O/bench_nusc.err; echo "nusc rc=$?"; cut -c1-220 $O/bench_nusc.json
bash tools/gpu_train_prof4.sh $(basename $O) > $O/prof4.log 2>&1; head -4 $O/kernel_stats_train_graph.txt | cut -c1-150
