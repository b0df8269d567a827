#!/bin/bash
# RPN live-tile list threshold (SEC_RPN_LIST_MAX_LIVE, percent of tiles live up to which a conv follows its list) on the dense seeded scene
TAG=${1:-r06_x}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
for P in 75 82 88 94 100; do
  for rep in 1 2; do
  SEC_RPN_LIST_MAX_LIVE=$P timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --scene dense --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs 2> $O/b$P.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('dense threshold $P:', d['value'], d['ms_per_step'], d['timing']['spread_pct'], d['config']['rpn_background_tiles'].get('live_tiles_per_conv'))" | tee -a $O/threshold.txt
  done
done
