export PYTHONUNBUFFERED=1
echo "== regression test on the shipped build"; timeout 600 python -m pytest tests/test_gpu_round2.py -m gpu -q -k "deterministic_beside" 2>&1 | tail -2
echo "== same test on a build WITH the vectorisers (must fail to have teeth)"; SEC_HIP_LIB=$PWD/second.pytorch_amd/lib/libsecond_hip_slp.so timeout 600 python -m pytest tests/test_gpu_round2.py -m gpu -q -k "deterministic_beside" 2>&1 | tail -3
