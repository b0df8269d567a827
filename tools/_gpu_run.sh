export PYTHONUNBUFFERED=1
echo "== vectorisers on, packed-fp32 feature off"; SEC_HIP_LIB=$PWD/second.pytorch_amd/lib/libsecond_hip_pf.so timeout 600 python tools/conv_microbench.py --all-layers --variants 1 2>&1 | grep "layer  [1-8]\|sum" | cut -c60-200
SEC_HIP_LIB=$PWD/second.pytorch_amd/lib/libsecond_hip_pf.so LOADKIND=conv LOAD=2 timeout 900 python tools/nms_stress.py 400 2>&1 | grep -v amdgpu.ids | tail -1
