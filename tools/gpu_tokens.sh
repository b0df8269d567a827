#!/bin/bash
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r04_tokens}; mkdir -p $O
for L in 4 5 6; do
  for T in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --inflight $L --rpn-tokens $T --no-cpu-baseline --no-other-configs --no-kernel-table --no-extra-lines > $O/l${L}_t$T.json 2> $O/l${L}_t$T.err
  python -c "
import json; d=json.load(open('$O/l${L}_t$T.json')); print('inflight $L tokens $T value', d['value'], 'ms', d['ms_per_step'], 'spread', d['timing']['spread_pct'])"
  done
done
