mkdir -p gpurun_out/r2l
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_conv_rows.py tests/test_gpu_parity.py -q -x -k "c4 or half_mfma or detector or first_layer" > gpurun_out/r2l/pytest_new.log 2>&1; echo "rc=$?" >> gpurun_out/r2l/pytest_new.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2l/smoke.log 2>&1
timeout 300 python tools/conv_microbench.py --all-layers --variants 29,1 --iters 100 > gpurun_out/r2l/layers.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2l/bench.json 2> gpurun_out/r2l/bench.err
tail -6 gpurun_out/r2l/pytest_new.log; tail -2 gpurun_out/r2l/smoke.log; head -3 gpurun_out/r2l/layers.log | cut -c1-250; tail -1 gpurun_out/r2l/layers.log; cut -c1-700 gpurun_out/r2l/bench.json
