#!/bin/bash
# single-step latency with the rulebook chain on side streams (graph branches): SEC_OVERLAP_RULEBOOKS=0/1/2
export PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/${1:-r03_q}; mkdir -p $O
for M in 0 1 2; do
  echo "== SEC_OVERLAP_RULEBOOKS=$M"
  SEC_OVERLAP_RULEBOOKS=$M timeout 300 python bench.py --steps 200 --warmup 20 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs 2>$O/ov$M.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(d['value'], d['ms_per_step'], 'single', c.get('single_step_latency_ms'), 'one-at-a-time', c.get('frames_per_s_one_step_at_a_time'))"
  tail -2 $O/ov$M.err
done
