#!/bin/bash
# Training kernels of the dense RPN: parity first, then the car.fhd training step with the hand-written path vs torch / MIOpen
# (+ a kernel-level profile of the hand-written path), then the packed-fp32 probe.
export PYTHONUNBUFFERED=1
R=$PWD
O=$PWD/gpurun_out/${1:-r03_e}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_dense.py tests/test_gpu_train.py tests/test_gpu_conv_rows.py -m gpu -q -x > $O/pytest_train.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_train.log
for be in hip miopen hip miopen; do
  SEC_RPN_TRAIN_BACKEND=$be timeout 300 python bench.py --workload car.fhd.train --dtype bf16 --steps 30 --warmup 5 2>/dev/null | tee -a $O/train_ab_$be.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$be', d['value'], d['unit'], d['ms_per_step'], d['loss_last_step'])"
done
cd /tmp; export TMPDIR=/tmp
SEC_RPN_TRAIN_BACKEND=hip timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_train -- python $R/bench.py --workload car.fhd.train --dtype bf16 --steps 10 --warmup 3 > $O/prof_train.log 2>&1
cd $R
db=$(find $O/prof_train -name "*.db" | head -1); python tools/rocprof_summary.py $db --steps 13 > $O/kernel_stats_train_bf16_bs4_hip.txt 2>&1
rm -rf $O/prof_train; head -45 $O/kernel_stats_train_bf16_bs4_hip.txt | cut -c1-100,110-175
for wl in nusc.fhd.train; do SEC_RPN_TRAIN_BACKEND=hip timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 2>/dev/null | cut -c1-300; done
