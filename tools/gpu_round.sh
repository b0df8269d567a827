#!/bin/bash
# One gpurun call = tests + the driver's bench line + the profiles the bench line's figures come from, all on the SAME build / box.
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r03_a [skip-tests]'
# Outputs land in gpurun_out/<tag>/; the summaries worth judging are copied to profiles/ by hand afterwards.
TAG=${1:-r03_x}
export PYTHONUNBUFFERED=1
R=$PWD
O=$R/gpurun_out/$TAG; mkdir -p $O
if [ "$2" != "skip-tests" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
  tail -3 $O/pytest.log
fi
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"     # the driver's own command line
cut -c1-400 $O/bench.json
timeout 600 python tools/bf16_error_trace.py > $O/bf16_error_trace.txt 2>&1; echo "trace rc=$?"; tail -32 $O/bf16_error_trace.txt
if [ "$3" == "yardstick" ]; then
  SERIES_OUT=$O/power_series.json timeout 300 python tools/rpn_yardstick.py > $O/rpn_yardstick.txt 2>&1; echo "yardstick rc=$?"; cat $O/rpn_yardstick.txt | cut -c1-330
fi
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -- python $R/bench.py --profile-run --steps 50 --warmup 10 --inflight 1 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs > $O/prof1.log 2>&1
cd $R
db=$(find $O/prof1 -name "*.db" | head -1); python tools/rocprof_summary.py $db --last-steps 40 --marker k_vox_init > $O/kernel_stats_bench_bs8_inflight1.txt 2>&1
python tools/rocprof_summary.py $db --timeline k_vox_init > $O/step_timeline_inflight1.txt 2>&1
rm -rf $O/prof1; head -12 $O/kernel_stats_bench_bs8_inflight1.txt | cut -c1-70,110-175
# PMC passes (counters only with --kernel-trace; FETCH_SIZE and WRITE_SIZE cannot share a pass)
cd /tmp
i=0
for SET in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  PMC_META=$O/pmc_meta.json timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc$i -- python $R/tools/pmc_workload.py > $O/pmc$i.log 2>&1
done
cd $R
python tools/pmc_report.py $O/pmc_meta.json $O/$TAG $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 > $O/pmc_report.log 2>&1; tail -60 $O/pmc_report.log
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5
ls -la $O
