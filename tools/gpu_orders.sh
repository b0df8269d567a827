#!/bin/bash
TAG=${1:-r06_i}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
for ORD in shuffle sorted scan; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --point-order $ORD --no-cpu-baseline --no-extra-lines --no-other-configs > $O/bench_$ORD.json 2> $O/bench_$ORD.err; echo "bench $ORD rc=$?"
done
python - <<PY
import json
ds={o: json.loads(open("$O/bench_%s.json" % o).read().strip().splitlines()[-1]) for o in ("shuffle","sorted","scan")}
print({o: (d["value"], d["config"].get("single_step_latency_ms")) for o,d in ds.items()})
for ks in zip(*[d["kernels"] for d in ds.values()]):
    print(f'{ks[0]["op"][:24]:24s}', " ".join(f'{k["us"]:8.2f}' for k in ks), ks[0].get("detail","")[:50])
PY
