timeout 900 python -m pytest tests -q -x -m gpu -k "voxel or pointpillars or pillar or pfn or block_filter" 2>&1 | tail -3
bash tools/gpu_pp_round.sh r06_pp10 > /dev/null 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/r06_pp10/bench_pp.json")); print("pp", d["value"], d["ms_per_step"], d["config"]["single_step_latency_ms"])
PY
head -14 gpurun_out/r06_pp10/step_timeline_nusc_pp_inflight1.txt | cut -c1-110
