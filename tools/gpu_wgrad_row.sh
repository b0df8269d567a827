#!/bin/bash
# A/B of the kernel-row form of the dense 3x3 weight gradient (SEC_WGRAD_ROW=0|1): parity tests, the wgrad probe, the captured training steps
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r06_ab}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_dense.py -q -x -m gpu > $O/pytest_dense.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_dense.log
for X in 0 1; do
  SEC_WGRAD_ROW=$X timeout 300 python - > $O/probe_$X.txt 2>&1 <<PY
import sys, torch
sys.path.insert(0, "second.pytorch_amd")
from second_amd import ops
for shape in ((4, 200, 176), (3, 248, 248), (8, 200, 176)):
    b, h, w = shape
    x = torch.randn(b, 128, h, w, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    dy = (torch.randn(b, 128, h, w, device="cuda") / 8).bfloat16().contiguous(memory_format=torch.channels_last)
    for _ in range(3): ops.conv2d_wgrad(x, dy)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): dw = ops.conv2d_wgrad(x, dy)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fl = 2.0 * b * h * w * 128 * 128 * 9
    print(f"SEC_WGRAD_ROW=$X {shape}: {us:.1f} us per wgrad + reduce, {fl / us / 1e6:.0f} TFLOP/s = {fl / us / 1e6 / 2500:.2f} of peak")
PY
  cat $O/probe_$X.txt | grep -v amdgpu.ids
done
for X in 0 1 0 1; do
  SEC_WGRAD_ROW=$X timeout 600 python bench.py --workload car.fhd.train --dtype bf16 --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_train_$X.json 2> $O/bench_train_$X.err; echo "car.fhd.train row=$X rc=$?"
  python -c "
import json
d=json.loads([l for l in open('$O/bench_train_$X.json').read().splitlines() if l.startswith('{')][-1]); print('   ', d['value'], d['unit'], d['ms_per_step'])"
done
