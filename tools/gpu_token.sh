#!/bin/bash
# RPN segments serialised by a token between the lanes (three graphs per step) vs one graph per step; SEC_RPN_TOKENS = segments allowed at a time
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_as}; mkdir -p $O
for CFG in "1 4 1" "1 4 2" "1 5 2" "1 6 2" "1 6 3" "1 4 1" "0 4 1"; do
  set -- $CFG
  echo "== --serialize-rpn $1 --inflight $2 SEC_RPN_TOKENS=$3"
  SEC_RPN_TOKENS=$3 timeout 120 python bench.py --steps 300 --warmup 30 --serialize-rpn $1 --inflight $2 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs 2>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(d['value'], d['ms_per_step'], 'single', c.get('single_step_latency_ms'), 'dets', d.get('detections_last_step'))" || tail -5 $O/bench.err
done
