mkdir -p gpurun_out/r2k
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2k/pytest_all.log 2>&1; echo "pytest_all rc=$?" >> gpurun_out/r2k/pytest_all.log
timeout 600 python bench.py --workload nusc.pp --steps 50 --warmup 10 > gpurun_out/r2k/nusc_pp.json 2> gpurun_out/r2k/nusc.err
timeout 600 python bench.py --workload nusc.fhd --steps 50 --warmup 10 > gpurun_out/r2k/nusc_fhd.json 2>> gpurun_out/r2k/nusc.err
timeout 600 python bench.py --workload nusc.fhd --steps 50 --warmup 10 --inflight 1 > gpurun_out/r2k/nusc_fhd_1.json 2>> gpurun_out/r2k/nusc.err
tail -12 gpurun_out/r2k/pytest_all.log; for f in nusc_pp nusc_fhd nusc_fhd_1; do cut -c1-1200 gpurun_out/r2k/$f.json; done; tail -20 gpurun_out/r2k/nusc.err
