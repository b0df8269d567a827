#!/bin/bash
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_aj}; mkdir -p $O
for WL in nusc.fhd nusc.pp; do
timeout 300 python bench.py --workload $WL --steps 50 --warmup 10 --no-cpu-baseline --no-extra-lines --no-other-configs 2>$O/$WL.err > $O/$WL.json
python - <<PY
import json
d=json.load(open("$O/$WL.json")); c=d['config']
print("$WL", d['value'], d['ms_per_step'], 'single', c.get('single_step_latency_ms'))
for k in d['kernels']: print("  %-22s %8.2f us  frac %s  %s" % (k['op'], k['us'], k.get('frac'), (k.get('kernel') or k.get('detail') or '')[:90]))
PY
done
