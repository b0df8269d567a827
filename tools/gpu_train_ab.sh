#!/bin/bash
# Training kernels of the dense RPN: parity first, then the car.fhd training step with the hand-written path vs torch / MIOpen,
# then BASELINE configs 4 / 5 at their stated sizes (urban synthetic clouds).
export PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/${1:-r03_c}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_dense.py tests/test_gpu_train.py -m gpu -q -x > $O/pytest_train.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_train.log
for be in hip miopen hip miopen; do
  SEC_RPN_TRAIN_BACKEND=$be timeout 300 python bench.py --workload car.fhd.train --dtype bf16 --steps 30 --warmup 5 2>/dev/null | tee -a $O/train_ab_$be.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$be', d['value'], d['unit'], d['ms_per_step'], d['loss_last_step'])"
done
for wl in nusc.pp nusc.fhd; do
  timeout 400 python bench.py --workload $wl --steps 50 --warmup 10 > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "$wl rc=$?"
  python -c "
import json
d=json.load(open('$O/bench_$wl.json')); k=d.pop('kernels',None) or []
print(d['value'], d['unit'], d['ms_per_step'], d['config']['points_per_frame'], d['config']['rows_per_frame'], d['config']['single_step_latency_ms'])
for e in k:
    print('   ', e.get('op'), e.get('us'), e.get('frac'), (e.get('detail') or e.get('error') or '')[:90])"
done
SEC_RPN_TRAIN_BACKEND=hip timeout 300 python bench.py --workload nusc.fhd.train --steps 20 --warmup 5 2>/dev/null | tee $O/train_nusc_fhd.json | cut -c1-600
