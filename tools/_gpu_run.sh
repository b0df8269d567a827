mkdir -p gpurun_out/r2b
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_conv_rows.py tests/test_gpu_round2.py -q > gpurun_out/r2b/pytest_new.log 2>&1; echo "pytest_new rc=$?" >> gpurun_out/r2b/pytest_new.log
timeout 300 python tools/conv_microbench.py --variants 8,1,9,13,16,17,18,19 --repeat 2 --iters 100 > gpurun_out/r2b/micro.log 2>&1
SEC_HIP_LIB=$PWD/second.pytorch_amd/lib/libsecond_hip_tl.so timeout 300 python tools/conv_microbench.py --variants 9,13,16,17,19 --timeline --iters 20 > gpurun_out/r2b/timeline.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "indice_conv or sparse_sequential or detector" > gpurun_out/r2b/pytest_conv.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/pytest_conv.log
tail -5 gpurun_out/r2b/pytest_new.log; cat gpurun_out/r2b/micro.log; cat gpurun_out/r2b/timeline.log; tail -3 gpurun_out/r2b/pytest_conv.log
