mkdir -p gpurun_out/r2j
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_round2.py -q > gpurun_out/r2j/pytest_new.log 2>&1; echo "rc=$?" >> gpurun_out/r2j/pytest_new.log
timeout 600 python bench.py --workload car.fhd.train --steps 20 --warmup 5 > gpurun_out/r2j/train_fp32.json 2> gpurun_out/r2j/train.err
timeout 600 python bench.py --workload car.fhd.train --steps 20 --warmup 5 --dtype bf16 > gpurun_out/r2j/train_bf16.json 2>> gpurun_out/r2j/train.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o p -- python $GRAFT_REPO_ROOT/bench.py --workload car.fhd.train --steps 10 --warmup 3 --dtype bf16 > $GRAFT_REPO_ROOT/gpurun_out/r2j/rocprof_train.log 2>&1
cp /tmp/prof1/p_results.db $GRAFT_REPO_ROOT/gpurun_out/r2j/train_bf16.db
cd $GRAFT_REPO_ROOT
tail -8 gpurun_out/r2j/pytest_new.log; cat gpurun_out/r2j/train_fp32.json gpurun_out/r2j/train_bf16.json | cut -c1-330; tail -5 gpurun_out/r2j/train.err
