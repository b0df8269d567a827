#!/usr/bin/env python3
"""Side measurement: the dense 3x3 / 1x1 weight gradient (sec_conv2d_wgrad_nhwc) at the car.fhd training shape, HIP events over
back-to-back launches.  Prints us per call and the fraction of the dense bf16 MFMA peak (2.5 PFLOP/s)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd")); sys.path.insert(0, ROOT)
import torch
from second_amd import ops

shapes = [(4, 200, 176, 3)] if os.environ.get("WGRAD_ONLY") else [(4, 200, 176, 3), (4, 200, 176, 1), (3, 256, 256, 3)]
for (b, h, w, k) in shapes:
    x = torch.randn(b, 128, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(b, 128, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(5):
        ops.conv2d_wgrad(x, dy, k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        ops.conv2d_wgrad(x, dy, k)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    fl = 2.0 * b * h * w * k * k * 128 * 128
    print(f"wgrad k{k} {b}x{h}x{w}: {us:.1f} us per call (kernel + reduce), {fl / us / 1e6 / 2.5e3:.3f} of 2.5 PF")
