#!/usr/bin/env python3
"""Micro-benchmark of sec_indice_conv_fwd on the car.fhd subm2 layer (64->64, batch 8) for profiling
(rocprofv3 --kernel-trace / --pmc).  SEC_CONV_VARIANT selects the kernel variant.

    python tools/conv_microbench.py [--iters 50] [--cin 64 --cout 64] [--layer subm2|subm0|subm1|subm3]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from second_amd import ops, synthetic as syn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--layer", default="subm2")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--sorted", action="store_true", help="spatially sorted clouds (rows in (z,y,x) order per frame)")
    ap.add_argument("--timeline", action="store_true", help="per-wave clock64 timeline of the split-K kernel (needs a build with SEC_EXTRA_HIPCC_FLAGS=-DSEC_CONV_TIMELINE)")
    ap.add_argument("--backward", action="store_true", help="also time sec_indice_conv_bwd (dgrad + wgrad), bf16 and fp32")
    args = ap.parse_args()
    dev = torch.device("cuda")
    clouds = [syn.syn_kitti_cloud(s) for s in range(args.batch)]
    if args.sorted:
        clouds = [c[np.lexsort((c[:, 0], c[:, 1], c[:, 2]))] for c in clouds]
    pts, offs = syn.batch_clouds(clouds)
    vox = ops.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev), syn.CAR_FHD_RANGE, syn.CAR_FHD_VOXEL, 5, 40000)
    idx, shape = vox["coordinates"].contiguous(), [41, 1600, 1408]
    plan = {"subm0": (0, 16), "subm1": (1, 32), "subm2": (2, 64), "subm3": (3, 64)}
    ndown, c = plan[args.layer]
    downs = [(3, 2, 1), (3, 2, 1), (3, 2, (0, 1, 1))]
    for i in range(ndown):
        r = ops.rulebook_conv(idx, args.batch, shape, *downs[i])
        idx, shape = r["out_indices"].contiguous(), r["out_shape"]
    rb = ops.rulebook_subm(idx, args.batch, shape, 3)
    n = idx.shape[0]
    pairs = int((rb["nbr_out"] >= 0).sum())
    g = torch.Generator(device="cpu").manual_seed(0)
    feat = torch.randn(n, c, generator=g).to(dev).bfloat16()
    w = (torch.randn(3, 3, 3, c, c, generator=g) / 30).to(dev).bfloat16()
    packed = ops.pack_weight(w)
    scale = torch.ones(c, device=dev)
    shift = torch.zeros(c, device=dev)
    for _ in range(5):
        out = ops.indice_conv(feat, w, rb["nbr_out"], n, packed=packed, scale=scale, shift=shift, relu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        out = ops.indice_conv(feat, w, rb["nbr_out"], n, packed=packed, scale=scale, shift=shift, relu=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.iters
    if args.backward:
        for dt in (torch.bfloat16, torch.float32):
            f, ww, do = feat.to(dt), w.to(dt), torch.randn(n, c, device=dev).to(dt)
            for need in ((True, False), (False, True)):
                for _ in range(3):
                    ops.indice_conv_backward(f, ww, rb["nbr_out"], None, do, *need)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(10):
                    ops.indice_conv_backward(f, ww, rb["nbr_out"], None, do, *need)
                e1.record()
                torch.cuda.synchronize()
                print(f"backward {'dgrad' if need[0] else 'wgrad'} {dt}: {e0.elapsed_time(e1) * 100:.1f} us")
    if args.timeline:
        import ctypes
        from second_amd import runtime as rt
        nw = (n + 31) // 32 * 4
        buf = torch.zeros((nw, 6), dtype=torch.int64, device=dev)
        rt.lib().sec__debug_timeline(ctypes.c_void_p(buf.data_ptr()))
        ops.indice_conv(feat, w, rb["nbr_out"], n, packed=packed, scale=scale, shift=shift, relu=True)
        torch.cuda.synchronize()
        rt.lib().sec__debug_timeline(ctypes.c_void_p(0))
        t = buf.cpu().numpy().astype(np.float64)
        t = t[t[:, 4] > 0]
        z = t[:, 0].min()
        seg = np.stack([t[:, 0] - z, t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 4] - z], 1)
        names = ["start_offset", "setup(idx)", "k-loop", "lds+barrier", "finish", "end_offset"]
        for i, nm in enumerate(names):
            q = np.percentile(seg[:, i], [5, 50, 95])
            print(f"  {nm:14s} p5={q[0]:9.0f} p50={q[1]:9.0f} p95={q[2]:9.0f}  (clock64 ticks)")
        print("  waves", len(t), "span ticks", seg[:, 5].max())
    b_alg = 2 * (pairs * c + n * c) + 8 * pairs + 2 * 27 * c * c
    print(f"variant={os.environ.get('SEC_CONV_VARIANT', 'default')} layer={args.layer} rows={n} pairs={pairs} C={c} "
          f"us/launch={us:.2f} alg_GBs={b_alg / us / 1e3:.1f} frac_of_8TBs={b_alg / us / 1e3 / 8000:.3f}")


if __name__ == "__main__":
    main()
