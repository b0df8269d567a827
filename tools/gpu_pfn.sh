#!/bin/bash
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_y}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k "pfn or pointpillars" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
for B in hip torch; do
  echo "== SEC_PFN_TRAIN_BACKEND=$B"
  SEC_PFN_TRAIN_BACKEND=$B timeout 300 python bench.py --workload nusc.pp.train --steps 15 --warmup 3 2>$O/pp_train_$B.err | cut -c1-700; tail -2 $O/pp_train_$B.err
done
