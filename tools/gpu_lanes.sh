#!/bin/bash
# headline against the number of steps in flight / RPN serialisation (several runs each: box noise is 1-2 %)
export PYTHONUNBUFFERED=1
for REP in 1 2 3; do
for ARGS in "--inflight 4" "--inflight 4 --serialize-rpn 0" "--inflight 3 --serialize-rpn 0" "--inflight 5 --serialize-rpn 0" "--inflight 6 --serialize-rpn 0"; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-lines --no-other-configs --no-kernel-table $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$ARGS', d['value'], d['ms_per_step'], d['timing']['spread_pct'])"
done
done
