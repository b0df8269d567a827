mkdir -p gpurun_out/r2z
export PYTHONUNBUFFERED=1
SEC_HIP_LIB=$PWD/second.pytorch_amd/lib/libsecond_hip_exp.so timeout 600 python tools/conv_microbench.py --all-layers --variants 29,19,32,33 > gpurun_out/r2z/layers_deep.txt 2>&1
grep "layer  [6-9]\|layer 1[0-3]\|sum" gpurun_out/r2z/layers_deep.txt | cut -c1-260
