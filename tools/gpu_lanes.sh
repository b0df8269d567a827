#!/bin/bash
# steps in flight sweep of the bench line (same box)
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r04_lanes}; mkdir -p $O
for L in ${2:-3 4 5 6}; do
  for S in ${3:-1}; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --inflight $L --serialize-rpn $S --no-cpu-baseline --no-other-configs --no-kernel-table --no-extra-lines > $O/lanes${L}_s$S.json 2> $O/lanes${L}_s$S.err
  python -c "
import json; d=json.load(open('$O/lanes${L}_s$S.json')); print('inflight $L serialize $S value', d['value'], 'ms', d['ms_per_step'], 'spread', d['timing']['spread_pct'])"
  done
done
