#!/bin/bash
# Row-order experiment: the bench's kernel table and a PMC pass with the clouds in shuffled and in
# cell (z, y, x) order.  Outputs: gpurun_out/<tag>/bench_{shuffle,sorted}.json, <tag>_{shuffle,sorted}_pmc.txt / _traffic.json
TAG=${1:-r06_a}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
for ORD in shuffle sorted; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --point-order $ORD --no-cpu-baseline --no-extra-lines --no-other-configs > $O/bench_$ORD.json 2> $O/bench_$ORD.err; echo "bench $ORD rc=$?"
  cut -c1-300 $O/bench_$ORD.json
done
cd /tmp; export TMPDIR=/tmp
for ORD in shuffle sorted; do
  i=0
  for SET in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    POINT_ORDER=$ORD PMC_META=$O/pmc_meta_$ORD.json timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc_${ORD}_$i -- python $R/tools/pmc_workload.py > $O/pmc_${ORD}_$i.log 2>&1
  done
  (cd $R; python tools/pmc_report.py $O/pmc_meta_$ORD.json $O/${TAG}_$ORD $O/pmc_${ORD}_1 $O/pmc_${ORD}_2 $O/pmc_${ORD}_3 $O/pmc_${ORD}_4 > $O/pmc_report_$ORD.log 2>&1)
  rm -rf $O/pmc_${ORD}_*
done
ls -la $O
