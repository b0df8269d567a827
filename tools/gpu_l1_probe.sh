#!/bin/bash
# One PMC pass over the eager static-capacity forward of the bench (every kernel of a step, the RPN convs on their live-tile lists) +
# bench.py's re-issue loops (subm2 64->64 x 105, the dense RPN conv x 300), summarised per (kernel, grid, in-step / re-issued burst).
# Default counter set = the L1 (TCP) view: does the L1 serve the repeated gathers of the sparse conv / the identical weight streams of
# the three RPN-conv workgroups a CU holds, or does every lookup go to L2?
#   gpurun --timeout 400 -- 'bash tools/gpu_l1_probe.sh r04_bn'                      -> gpurun_out/<tag>/<tag>_tcp_l1.txt + the driver's bench line
#   gpurun --timeout 200 -- 'bash tools/gpu_l1_probe.sh r04_bo "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" sq nobench'
TAG=${1:-r04_l1}
SET=${2:-TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum}
NAME=${3:-tcp_l1}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 270 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/$NAME -- \
  python $R/bench.py --mode static --profile-run --steps 4 --warmup 2 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs > $O/$NAME.log 2>&1
echo "pmc rc=$?"; tail -3 $O/$NAME.log | cut -c1-300
cd $R
python tools/pmc_l1_summary.py $O/$NAME 2 > $O/${TAG}_$NAME.txt 2>&1; grep -E "^#|^kernel|sec::" -A 6 $O/${TAG}_$NAME.txt | grep -E "^#|^kernel|sec::|^    " | head -150 | cut -c1-60,97-210
find $O/$NAME -name "*.csv" -size +20M -delete
if [ "$4" != "nobench" ]; then
  timeout 240 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
fi
