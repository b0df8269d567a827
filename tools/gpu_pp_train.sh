#!/bin/bash
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_ad}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_dense.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for M in mixed torch; do
  echo "== SEC_PP_TRAIN_RPN=$M"
  SEC_PP_TRAIN_RPN=$M timeout 300 python bench.py --workload nusc.pp.train --steps 20 --warmup 5 2>$O/pp_train_$M.err | cut -c1-420; tail -2 $O/pp_train_$M.err
done
