#!/bin/bash
# Last RPN conv with the 1x1 tail in its epilogue (sec_conv2d_nhwc_tiles_tail) vs the two launches (SEC_RPN_FUSED_TAIL=0): tests + interleaved A/B
#   gpurun --timeout 1200 -- 'bash tools/gpu_tail_ab.sh r06_tail'
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r06_tail}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rpn_tiles.py tests/test_capi_symbols.py -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
for rep in 1 2; do
for mode in 1 0; do
  SEC_RPN_FUSED_TAIL=$mode timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-lines --no-other-configs $2 > $O/b_$mode.json 2> $O/b_$mode.err
  python - <<PY
import json
try:
    d = json.load(open("$O/b_$mode.json"))
    print("fused tail $mode: %.0f frames/s  %.4f ms/step  single %.4f" % (d["value"], d["ms_per_step"], d["config"]["single_step_latency_ms"]))
    for k in d.get("kernels", []):
        if k["op"] in ("conv2d_nhwc_tiles_tail", "conv1x1_chain") or (k["op"] == "conv2d_nhwc_tiles" and k.get("live_tiles", 0) == max(x.get("live_tiles", 0) for x in d["kernels"])):
            print("   ", k["op"], k["us"], k.get("live_tiles"), k.get("frac"))
except Exception as e:
    print("fused tail $mode: FAILED", e); print(open("$O/b_$mode.err").read()[-800:])
PY
done
done
