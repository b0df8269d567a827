mkdir -p gpurun_out/r2h
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o p -- python $R/bench.py --steps 50 --warmup 10 --inflight 1 --no-cpu-baseline --no-kernel-table > $R/gpurun_out/r2h/rocprof_bench.log 2>&1
cp /tmp/prof1/p_results.db $R/gpurun_out/r2h/bench_inflight1.db
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o p -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-table > $R/gpurun_out/r2h/rocprof_bench3.log 2>&1
cp /tmp/prof2/p_results.db $R/gpurun_out/r2h/bench_inflight3.db
for c in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/tools/conv_microbench.py --layer subm2 --iters 20 > $R/gpurun_out/r2h/pmc_$c.log 2>&1
mkdir -p $R/gpurun_out/r2h/pmc_$c; cp /tmp/pmc_$c/*.csv $R/gpurun_out/r2h/pmc_$c/ 2>/dev/null
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq -o p -- python $R/tools/conv_microbench.py --layer subm2 --iters 20 > $R/gpurun_out/r2h/pmc_sq.log 2>&1
mkdir -p $R/gpurun_out/r2h/pmc_sq; cp /tmp/pmc_sq/*.csv $R/gpurun_out/r2h/pmc_sq/ 2>/dev/null
cd $R; ls -la gpurun_out/r2h gpurun_out/r2h/pmc_sq; tail -3 gpurun_out/r2h/pmc_sq.log
