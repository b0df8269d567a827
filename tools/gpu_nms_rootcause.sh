#!/bin/bash
# k_nms_mask determinism beside the RPN conv (tools/nms_stress.py) under build variants that classify the fault:
#   pk      vectorised build (packed fp32 on): the failing one (round 2: 25-30 % of the launches)
#   pksnop  + -mllvm --amdgpu-snop-padding=2 : every instruction preceded by s_nop -> an intra-wave VALU timing hazard disappears
#   pkwait  + -mllvm -amdgpu-waitcnt-forcezero: every wait is vmcnt(0) lgkmcnt(0) -> a memory / LDS ordering fault disappears
# build first (cross-compile):  for t in pk pksnop pkwait: SEC_BUILD_TAG=$t SEC_EXTRA_HIPCC_FLAGS="..." python second.pytorch_amd/build.py
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r04_nms}; mkdir -p $O
for t in "" _pk _pksnop _pkwait; do
  lib=$R/second.pytorch_amd/lib/libsecond_hip$t.so; [ -f $lib ] || continue
  echo "== build '$t'" >> $O/nms_rootcause.txt
  SEC_HIP_LIB=$lib timeout 250 python tools/nms_stress.py ${2:-500} 2>&1 | tail -4 >> $O/nms_rootcause.txt
done
cat $O/nms_rootcause.txt
