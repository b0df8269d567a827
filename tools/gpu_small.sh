#!/bin/bash
export PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/${1:-r03_n}; mkdir -p $O
timeout 300 python tools/step_census.py 2>&1 | grep -v amdgpu.ids | tee $O/step_census.txt | cut -c1-220 | head -60
timeout 300 python bench.py --batch 16 --steps 100 --warmup 10 --no-cpu-baseline --no-extra-lines --no-kernel-table 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch16', d['value'], d['ms_per_step'], d['config']['single_step_latency_ms'], d['roofline'])" | tee $O/bench_batch16.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
