#!/usr/bin/env python3
"""Per-workgroup timeline of k_conv2d_halo_reg (needs a build with SEC_EXTRA_HIPCC_FLAGS=-DSEC_CONV_TIMELINE, loaded through
SEC_HIP_LIB) + launch time over the batch size (how the time quantises into rounds of resident workgroups).
(The SEC_CONV2D_STAGGER experiment of round 2 -- delaying the resident slots of a CU -- is gone: no effect.)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd"))
import numpy as np
import torch
from second_amd import ops, runtime as rt
torch.manual_seed(0)
w = (torch.randn(128, 128, 3, 3, device="cuda") / 34).bfloat16()
b = torch.randn(128, device="cuda")
pk = ops.conv2d_pack_weight(w)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def bench(x, n=100):
    for _ in range(200):
        ops.conv2d_nhwc(x, pk, b, 128, 3, 1, 1, relu=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        ops.conv2d_nhwc(x, pk, b, 128, 3, 1, 1, relu=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for batch in [int(v) for v in os.environ.get("BATCHES", "1,2,3,4,6,8,16").split(",")]:
    x = torch.relu(torch.randn(batch, 128, 200, 176, device="cuda")).bfloat16().contiguous(memory_format=torch.channels_last)
    t = bench(x)
    tiles = batch * 25 * 11
    flop = 2.0 * batch * 200 * 176 * 128 * 128 * 9
    print(f"batch {batch:2d}: {tiles:5d} tiles = {tiles / 768:5.2f} rounds of 768  {t:7.2f} us  {flop / t / 1e6:6.0f} TFLOP/s", flush=True)
    if batch == 8 and hasattr(rt.lib(), "sec__debug_timeline2"):
        buf = torch.zeros((tiles + 8, 8), dtype=torch.int64, device="cuda")
        rt.lib().sec__debug_timeline2(ctypes.c_void_p(buf.data_ptr()))
        ops.conv2d_nhwc(x, pk, b, 128, 3, 1, 1, relu=True)
        torch.cuda.synchronize()
        rt.lib().sec__debug_timeline2(ctypes.c_void_p(0))
        raw = buf.cpu().numpy()
        raw = raw[raw[:, 3] > 0]
        res = (raw[:, 2] >> 56) & 0xff                      # workgroups already resident on the CU when this one started
        raw[:, 2] &= (1 << 56) - 1
        print("    resident workgroups on the CU at start (0,1,2,3+):", [int((res == v).sum()) for v in (0, 1, 2)], int((res >= 3).sum()))
        tt = raw.astype(np.float64)
        for name, col in (("halo load (prologue)", tt[:, 1] - tt[:, 0]), ("MFMA loop", tt[:, 2] - tt[:, 1]), ("epilogue stores", tt[:, 3] - tt[:, 2]),
                          ("workgroup life", tt[:, 3] - tt[:, 0])):
            q = np.percentile(col, [5, 50, 95])
            print(f"    {name:22s} p5={q[0]:8.0f} p50={q[1]:8.0f} p95={q[2]:8.0f} clocks")
        if raw[:, 7].any():                                 # finer stamps (16-clock units): DMA issue, the two epilogue barriers
            sub = raw[:, 7]
            for name, col in (("  prologue: halo DMA issue", (sub & 0xffff) * 16), ("  epilogue: loop end -> barrier 1 (wave skew)", (sub >> 16 & 0xffff) * 16),
                              ("  epilogue: -> barrier 2 (transposed tile in LDS)", (sub >> 32 & 0xffff) * 16)):
                q = np.percentile(col.astype(np.float64), [5, 50, 95])
                print(f"    {name:50s} p5={q[0]:8.0f} p50={q[1]:8.0f} p95={q[2]:8.0f} clocks")
        # real time (s_memrealtime, 100 MHz): shader clock during the kernel and how full the 3 resident slots per CU are
        w0, w1 = tt[:, 4], tt[:, 5]
        span = (w1.max() - w0.min()) * 10e-9
        ghz = (tt[:, 3] - tt[:, 0]) / ((w1 - w0) * 10.0)
        q = np.percentile(ghz, [5, 50, 95])
        ncu = len(np.unique(raw[:, 6]))
        print(f"    kernel span {span * 1e6:.1f} us over {ncu} CUs; shader clock per workgroup life p5={q[0]:.2f} p50={q[1]:.2f} p95={q[2]:.2f} GHz")
        print(f"    resident-slot utilisation = sum(lives) / (3 slots x {ncu} CUs x span) = {((w1 - w0).sum() * 10e-9) / (3 * ncu * span):.3f}")
        per_cu = {}
        for k, a, b_ in zip(raw[:, 6], w0, w1):
            per_cu.setdefault(int(k), []).append((a, b_))
        gaps = []
        for k, iv in per_cu.items():
            iv.sort()
            ends = sorted(b_ for _, b_ in iv)
            # the i-th start beyond the first three follows the (i-3)-th end on this CU
            for i in range(3, len(iv)):
                gaps.append((iv[i][0] - ends[i - 3]) * 10.0)
        if gaps:
            q = np.percentile(gaps, [5, 50, 95])
            print(f"    slot turnaround (end of a workgroup -> start of its successor on the CU) p5={q[0]:.0f} p50={q[1]:.0f} p95={q[2]:.0f} ns")
        first = np.array([min(a for a, _ in iv) for iv in per_cu.values()]); last = np.array([max(b_ for _, b_ in iv) for iv in per_cu.values()])
        print(f"    first start per CU spread {(first.max() - first.min()) * 10:.0f} ns; last end per CU: p5={np.percentile(last - w0.min(), 5) * 10e-3:.1f} p50={np.percentile(last - w0.min(), 50) * 10e-3:.1f} max={(last.max() - w0.min()) * 10e-3:.1f} us")
        # start times per XCD (blockIdx % 8), relative to the XCD's first workgroup: the rounds
        for xcd in (0, 3):
            sel = np.arange(len(buf))[:len(tt)] % 8 == xcd
            st = np.sort(tt[sel, 0] - tt[sel, 0].min())
            print(f"    XCD {xcd}: start offsets (clocks) deciles", np.percentile(st, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100]).astype(int).tolist())
