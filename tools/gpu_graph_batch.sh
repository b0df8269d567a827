#!/bin/bash
# hipGraph replay submits its kernel nodes in batches (32 by default): a ~48 us bubble after the 32nd launch of a step.
# Single-step latency and the launch timeline over DEBUG_HIP_GRAPH_BATCH_SIZE.
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_s}; mkdir -p $O
for B in default 8 64 128 256; do
  if [ $B == default ]; then unset DEBUG_HIP_GRAPH_BATCH_SIZE; else export DEBUG_HIP_GRAPH_BATCH_SIZE=$B; fi
  echo "== DEBUG_HIP_GRAPH_BATCH_SIZE=$B"
  timeout 300 python bench.py --steps 200 --warmup 20 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs 2>$O/b$B.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(d['value'], d['ms_per_step'], 'single', c.get('single_step_latency_ms'), 'one-at-a-time', c.get('frames_per_s_one_step_at_a_time'))"
done
cd /tmp; export TMPDIR=/tmp
for B in 8 128; do
  export DEBUG_HIP_GRAPH_BATCH_SIZE=$B
  timeout 300 rocprofv3 --kernel-trace -d $O/prof$B -- python $R/bench.py --steps 30 --warmup 10 --inflight 1 --no-kernel-table --no-cpu-baseline --no-extra-lines --no-other-configs > $O/prof$B.log 2>&1
  db=$(find $O/prof$B -name "*.db" | head -1); python $R/tools/rocprof_summary.py $db --timeline k_vox_init > $O/step_timeline_b$B.txt 2>&1
  rm -rf $O/prof$B; echo "== timeline, batch size $B"; awk '$3 > 3.0 || /^#/' $O/step_timeline_b$B.txt | cut -c1-130
done
