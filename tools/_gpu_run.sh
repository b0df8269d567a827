mkdir -p gpurun_out/r2x
export PYTHONUNBUFFERED=1
AMD_SERIALIZE_KERNEL=3 timeout 600 python tools/_dbg_train.py > gpurun_out/r2x/dbg.txt 2>&1
tail -12 gpurun_out/r2x/dbg.txt
timeout 900 python bench.py --workload nusc.fhd.train --steps 10 --warmup 3 > gpurun_out/r2x/bench_nusc_train.json 2> gpurun_out/r2x/bench_nusc_train.err
cat gpurun_out/r2x/bench_nusc_train.json; tail -5 gpurun_out/r2x/bench_nusc_train.err
