#!/usr/bin/env python3
"""Micro-benchmark of sec_conv2d_nhwc on the car.fhd RPN layer (3x3, 128->128, 8 x 200 x 176) vs MIOpen."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd"))
import torch
from second_amd import ops
torch.manual_seed(0)
x = torch.randn(8, 128, 200, 176, device="cuda")
density = float(os.environ.get("DENSITY", "1.0"))   # fraction of non-zero pixels (the real RPN input is ~3 % dense)
if density < 1.0:
    x = x * (torch.rand(8, 1, 200, 176, device="cuda") < density)
x = x.bfloat16().contiguous(memory_format=torch.channels_last)
w = (torch.randn(128, 128, 3, 3, device="cuda") / 34).bfloat16()
b = torch.randn(128, device="cuda")
pk = ops.conv2d_pack_weight(w)
WARM, N = int(os.environ.get("WARM", "300")), int(os.environ.get("ITERS", "100"))
def bench(fn, n=N):
    for _ in range(WARM): fn()          # ~0.1 s of warm-up: the first launches of a process run at idle clocks
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
ZSKIP = bool(int(os.environ.get("ZSKIP", "0")))   # flag the input as a scattered sparse tensor (all-zero tiles skipped)
t = bench(lambda: ops.conv2d_nhwc(x, pk, b, 128, 3, 1, 1, relu=True, sparse_input=ZSKIP))
flop = 2 * 8 * 200 * 176 * 128 * 128 * 9
print(f"density={density} zskip={int(ZSKIP)} hip: {t:.1f} us  {flop / t / 1e6:.0f} TFLOP/s")
if os.environ.get("WITH_MIOPEN"):
    wcl = w.contiguous(memory_format=torch.channels_last)
    t = bench(lambda: ops.bias_act_(torch.nn.functional.conv2d(x, wcl, None, 1, 1), b, True))
    print(f"miopen conv + fused bias/relu: {t:.1f} us  {flop / t / 1e6:.0f} TFLOP/s")
