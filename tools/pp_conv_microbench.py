#!/usr/bin/env python3
"""Every conv layer shape of the PointPillars RPN (nuscenes/all.pp.largea, batch 4) through sec_conv2d_nhwc: us per launch and
fraction of the bf16 MFMA peak.  Run once per library build / env setting (SEC_HIP_LIB, SEC_CONV2D_PATCH) to compare forms."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd"))
import torch
from second_amd import ops
torch.manual_seed(0)
B = int(os.environ.get("BATCH", "4"))
LAYERS = [(64, 64, 3, 2, 1, 400), (64, 64, 3, 1, 1, 200), (64, 128, 4, 4, 0, 200), (64, 128, 3, 2, 1, 200), (128, 128, 3, 1, 1, 100),
          (128, 128, 2, 2, 0, 100), (128, 256, 3, 2, 1, 100), (256, 256, 3, 1, 1, 50), (256, 128, 1, 1, 0, 50), (384, 256, 1, 1, 0, 50)]
GRAPH = int(os.environ.get("GRAPH", "1"))      # time launches replayed from a hipGraph (an eager launch through ctypes has a ~10 us floor)
if os.environ.get("SHAPES") == "nusc.fhd":      # config 5's RPN (all.fhd: 248 x 248 BEV map, blocks of 128 and 256 channels)
    LAYERS = [(128, 128, 3, 1, 1, 248), (128, 128, 2, 2, 0, 248), (128, 256, 3, 2, 1, 248), (256, 256, 3, 1, 1, 124), (64, 64, 3, 2, 1, 800)]
def bench(fn, warm=200, n=200):
    if GRAPH:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(3): fn()
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(20): fn()
        for _ in range(warm // 20): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n // 20): g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (n // 20 * 20)
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tag = os.environ.get("TAG", os.path.basename(os.environ.get("SEC_HIP_LIB", "default")))
tot = 0.0
for cin, cout, k, s, p, hw in LAYERS:
    x = torch.randn(B, cin, hw, hw, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5).bfloat16()
    b = torch.randn(cout, device="cuda")
    pk = ops.conv2d_pack_weight(w)
    t = bench(lambda: ops.conv2d_nhwc(x, pk, b, cout, k, s, p, relu=True))
    ho = (hw + 2 * p - k) // s + 1
    flop = 2 * B * ho * ho * cin * cout * k * k
    tot += t
    print(f"{tag}: {cin:3d}->{cout:3d} k{k} s{s} {hw:3d}x{hw:<3d} {t:6.2f} us  {flop / t / 1e6:6.0f} TFLOP/s  {flop / t / 1e6 / 2500:.3f} of peak", flush=True)
print(f"{tag}: sum {tot:.1f} us")
