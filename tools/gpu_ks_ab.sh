#!/bin/bash
# A/B of the offset-split form of the mid-size 64 -> 64 sparse convs (k_conv_rows_ks, SEC_CONV_KS=0|1): parity tests + bench kernel table
TAG=${1:-r06_u}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_rows.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for X in 0 1; do
  SEC_CONV_KS=$X timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-lines --no-other-configs > $O/bench_ks$X.json 2> $O/bench_ks$X.err; echo "bench ks=$X rc=$?"
  cut -c1-200 $O/bench_ks$X.json
done
ls $O
