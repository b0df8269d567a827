mkdir -p gpurun_out/r2v
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_properties.py tests/test_gpu_round2.py tests/test_gpu_e2e.py -m gpu -q -x > gpurun_out/r2v/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2v/pytest.log
timeout 600 python bench.py > gpurun_out/r2v/bench.json 2> gpurun_out/r2v/bench.err
timeout 300 python bench.py --inflight 1 --no-kernel-table > gpurun_out/r2v/bench1.json 2>> gpurun_out/r2v/bench.err
tail -4 gpurun_out/r2v/pytest.log; cat gpurun_out/r2v/bench1.json | cut -c1-300
