#!/bin/bash
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_ak}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_round2.py -m gpu -q -x -k "predict or select or detector or e2e or bf16 or nusc or multiclass" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for WL in nusc.fhd nusc.pp car.fhd; do
timeout 300 python bench.py --workload $WL --steps 100 --warmup 10 --no-cpu-baseline --no-extra-lines --no-other-configs 2>$O/$WL.err > $O/$WL.json
python - <<PY
import json
d=json.load(open("$O/$WL.json")); c=d['config']
print("$WL", d['value'], d['ms_per_step'], 'single', c.get('single_step_latency_ms'), [(k['op'],k['us']) for k in d['kernels'] if k['op'].startswith(('predict','nms'))])
PY
done
