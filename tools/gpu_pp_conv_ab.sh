#!/bin/bash
# PointPillars RPN conv layers one by one (tools/pp_conv_microbench.py, graph-replayed launches): shipped library, generic kernel, experiment builds
R=$PWD; O=$R/gpurun_out/${1:-r06_pp_ab}; mkdir -p $O
L=$R/second.pytorch_amd/lib
: > $O/mb.txt
[ -z "$SKIP_BASE" ] && TAG=default python tools/pp_conv_microbench.py >> $O/mb.txt 2>&1
[ -z "$SKIP_BASE" ] && TAG=generic SEC_CONV2D_PATCH=0 python tools/pp_conv_microbench.py >> $O/mb.txt 2>&1
for t in ${TAGS:-s1rd8 abl1 abl2 abl3}; do [ -f $L/libsecond_hip_$t.so ] && SEC_HIP_LIB=$L/libsecond_hip_$t.so TAG=$t python tools/pp_conv_microbench.py >> $O/mb.txt 2>&1; done
grep -v amdgpu.ids $O/mb.txt
