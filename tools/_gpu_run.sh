mkdir -p gpurun_out/r2r
export PYTHONUNBUFFERED=1
SEC_DEBUG_OCCUPANCY=1 SEC_HIP_LIB=$PWD/second.pytorch_amd/lib/libsecond_hip_tl.so BATCHES=8,16 timeout 200 python tools/conv2d_timeline.py 2>&1 | grep -v amdgpu > gpurun_out/r2r/res.log
cat gpurun_out/r2r/res.log
