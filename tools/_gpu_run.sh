mkdir -p gpurun_out/r2u
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2u/pytest_all.log 2>&1; echo "pytest_all rc=$?" >> gpurun_out/r2u/pytest_all.log
SEC_HIP_LIB=$PWD/second.pytorch_amd/lib/libsecond_hip_exp.so timeout 900 python -m pytest tests/test_gpu_conv_rows.py -q -x > gpurun_out/r2u/pytest_exp.log 2>&1; echo "rc=$?" >> gpurun_out/r2u/pytest_exp.log
tail -4 gpurun_out/r2u/pytest_all.log; tail -4 gpurun_out/r2u/pytest_exp.log
