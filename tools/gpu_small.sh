#!/bin/bash
export PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/${1:-r03_o}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -q -x -k "pointpillars or predict or detector" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python bench.py --workload nusc.pp.train --steps 15 --warmup 3 2>$O/pp_train.err | cut -c1-500; tail -3 $O/pp_train.err
