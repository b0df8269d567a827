#!/bin/bash
# RPN segments serialised (token / one CU-masked stream) vs one graph per step
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_as}; mkdir -p $O
for CFG in "1 4 ff" "1 4 fe" "1 4 ee" "1 4 7e" "1 3 fe" "1 5 fe" "1 4 -" "0 3 -"; do
  set -- $CFG
  M=$3; [ "$M" == "-" ] && M=""
  echo "== --serialize-rpn $1 --inflight $2 SEC_RPN_CU_MASK_BYTE=$M"
  SEC_RPN_CU_MASK_BYTE=$M timeout 120 python bench.py --steps 300 --warmup 30 --serialize-rpn $1 --inflight $2 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs 2>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(d['value'], d['ms_per_step'], 'single', c.get('single_step_latency_ms'), 'dets', d.get('detections_last_step'))" || tail -5 $O/bench.err
done
