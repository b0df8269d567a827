#!/bin/bash
# Where does the 64 -> 64 sparse conv's time go?  Timing-only ablations (wrong results) of the shipped form (44: no gather touches
# memory, 45: all gathers hit one row) and of the two-tiles-per-wave form (42 / 43), next to the real kernels (22, 41), one process.
export PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/${1:-r03_l}; mkdir -p $O
SEC_HIP_LIB=$PWD/second.pytorch_amd/lib/libsecond_hip_abl.so timeout 300 python tools/conv_microbench.py --layer subm2 --variants 22,44,45,41,42,43,22,44,45,41,42,43 --iters 200 2>&1 | grep -v amdgpu.ids | tee $O/ablations_subm2.txt
