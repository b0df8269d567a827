#!/bin/bash
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_ae}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_dense.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for WL in car.fhd.train nusc.fhd.train nusc.pp.train; do
  timeout 300 python bench.py --workload $WL --dtype bf16 --steps 40 --warmup 8 --no-cpu-baseline --no-extra-lines --no-other-configs 2>$O/$WL.err | cut -c1-330; tail -1 $O/$WL.err | cut -c1-200
done
