#!/bin/bash
export PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/${1:-r03_m}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_rows.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout 300 python tools/conv_microbench.py --layer subm2 --sorted-numbering --variants 22,41,46,22,41,46 --iters 200 2>&1 | grep -v amdgpu.ids | tee $O/microbench_subm2_sorted.txt
timeout 300 python tools/conv_microbench.py --layer subm3 --sorted-numbering --variants 22,46,22,46 --iters 200 2>&1 | grep -v amdgpu.ids | tee $O/microbench_subm3_sorted.txt
timeout 300 python tools/conv_microbench.py --layer subm2 --variants 22,46 --iters 200 2>&1 | grep -v amdgpu.ids | tee $O/microbench_subm2_first_touch.txt
