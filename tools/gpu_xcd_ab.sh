#!/bin/bash
# A/B of the XCD-contiguous tile order of k_conv_rows_buf (SEC_CONV_ROWS_XCD=0|1): bench kernel table + traffic counters
TAG=${1:-r06_g}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv_rows.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for X in 0 1; do
  SEC_CONV_ROWS_XCD=$X timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-lines --no-other-configs > $O/bench_xcd$X.json 2> $O/bench_xcd$X.err; echo "bench xcd=$X rc=$?"
  cut -c1-200 $O/bench_xcd$X.json
done
cd /tmp; export TMPDIR=/tmp
for X in 0 1; do
  i=0
  for SET in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    SEC_CONV_ROWS_XCD=$X PMC_META=$O/pmc_meta_xcd$X.json timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc_${X}_$i -- python $R/tools/pmc_workload.py > $O/pmc_${X}_$i.log 2>&1
  done
  (cd $R; python tools/pmc_report.py $O/pmc_meta_xcd$X.json $O/${TAG}_xcd$X $O/pmc_${X}_1 $O/pmc_${X}_2 $O/pmc_${X}_3 $O/pmc_${X}_4 > $O/pmc_report_xcd$X.log 2>&1)
  rm -rf $O/pmc_${X}_*
done
ls $O
