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


def all_layers(args, dev):
    """Per-layer A/B over the whole car.fhd sparse stack (middle.py:146-189), bf16, fused scale/shift/ReLU epilogue."""
    from second_amd import runtime as rt
    clouds = [syn.syn_kitti_cloud(s) for s in range(args.batch)]
    pts, offs = syn.batch_clouds(clouds)
    vox = ops.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev), syn.CAR_FHD_RANGE, syn.CAR_FHD_VOXEL, 5, 40000)
    idx, shape = vox["coordinates"].contiguous(), [41, 1600, 1408]
    plan = [("subm", 4, 16), ("subm", 16, 16), ("down", 16, 32, 3, 2, 1), ("subm", 32, 32), ("subm", 32, 32), ("down", 32, 64, 3, 2, 1),
            ("subm", 64, 64), ("subm", 64, 64), ("subm", 64, 64), ("down", 64, 64, 3, 2, (0, 1, 1)), ("subm", 64, 64), ("subm", 64, 64),
            ("subm", 64, 64), ("down", 64, 64, (3, 1, 1), (2, 1, 1), 0)]
    variants = [int(x) for x in (args.variants or "29,22").split(",")]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.Generator(device="cpu").manual_seed(0)
    rb, total = None, {v: 0.0 for v in variants}
    for li, spec in enumerate(plan):
        kind, cin, cout = spec[:3]
        if kind == "subm":
            if rb is None or rb["kind"] != ("subm", idx.shape[0]):
                rb = ops.rulebook_subm(idx, args.batch, shape, 3)
                rb["kind"] = ("subm", idx.shape[0])
            nbr, n_out, ks = rb["nbr_out"], idx.shape[0], (3, 3, 3)
        else:
            r = ops.rulebook_conv(idx, args.batch, shape, spec[3], spec[4], spec[5])
            nbr, n_out = r["nbr_out"], r["num_out"]
            ks = (spec[3],) * 3 if isinstance(spec[3], int) else spec[3]
        n_in = idx.shape[0]
        pairs = int((nbr >= 0).sum())
        feat = torch.randn(n_in, cin, generator=g).to(dev).bfloat16()
        w = (torch.randn(*ks, cin, cout, generator=g) / 30).to(dev).bfloat16()
        packed = ops.pack_weight(w)
        scale, shift = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        run = lambda: ops.indice_conv(feat, w, nbr, n_out, packed=packed, scale=scale, shift=shift, relu=True)
        b_alg = 2 * (pairs * cin + n_out * cout) + 8 * pairs + 2 * int(np.prod(ks)) * cin * cout
        line = f"layer {li:2d} {kind} {cin:2d}->{cout:2d} k{''.join(map(str, ks))} rows_in={n_in:6d} rows_out={n_out:6d} pairs={pairs:7d} B_alg={b_alg / 1e6:6.1f}MB |"
        for v in variants:
            rt.lib().sec_indice_conv_set_variant(v)
            pl = rt.lib().sec_indice_conv_fwd_plan(cin, cout, int(np.prod(ks)), n_out, 2, 2, 1 if packed is not None else 0)
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            # the re-issues are replayed from a hipGraph: a Python -> ctypes launch costs ~10-15 us on the host, more than the
            # small layers' kernels take, so eager back-to-back launches would measure the host
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_, capture_error_mode="thread_local"):
                for _ in range(args.iters):
                    run()
            g_.replay()
            torch.cuda.synchronize()
            e0.record()
            g_.replay()
            e1.record()
            torch.cuda.synchronize()
            del g_
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            total[v] += us
            line += f" v{v}(plan {pl}): {us:6.2f} us {b_alg / us / 1e3 / 8000:.3f}"
        print(line, flush=True)
        if kind == "down":
            idx, shape, rb = r["out_indices"].contiguous(), r["out_shape"], None
    print("sum over the 14 layers:", {v: round(t, 1) for v, t in total.items()}, "us")
    rt.lib().sec_indice_conv_set_variant(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--layer", default="subm2")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--sorted", action="store_true", help="spatially sorted clouds (rows in (z,y,x) order per frame)")
    ap.add_argument("--sorted-numbering", action="store_true", help="strided rulebooks in spconv's GPU numbering (ascending cell order): what the device fast path runs")
    ap.add_argument("--timeline", action="store_true", help="per-wave clock64 timeline of the row-split kernels (with --variants; needs a build with SEC_EXTRA_HIPCC_FLAGS=-DSEC_CONV_TIMELINE)")
    ap.add_argument("--variants", default="", help="comma list of SEC_CONV_VARIANT numbers timed back to back in this process (each checked against split-K)")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--backward", action="store_true", help="also time sec_indice_conv_bwd (dgrad + wgrad), bf16 and fp32")
    ap.add_argument("--all-layers", action="store_true", help="every conv layer of SpMiddleFHD at batch 8, each --variants entry per layer")
    args = ap.parse_args()
    dev = torch.device("cuda")
    if args.sorted_numbering:
        ops.set_rulebook_numbering("sorted")
    if args.all_layers:
        return all_layers(args, dev)
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
    from second_amd import runtime as rt
    b_alg = 2 * (pairs * c + n * c) + 8 * pairs + 2 * 27 * c * c
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    run = lambda: ops.indice_conv(feat, w, rb["nbr_out"], n, packed=packed, scale=scale, shift=shift, relu=True)

    def time_it(iters):
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            out = run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters, out

    if args.variants:
        # several kernel families in ONE process (same box, same clocks): sec_indice_conv_set_variant switches between them
        rt.lib().sec_indice_conv_set_variant(8)
        ref = ops.indice_conv(feat, w, rb["nbr_out"], n, packed=packed, scale=scale, shift=shift, relu=True, out_dtype=torch.float32)
        code = rt.dtype_code(feat.dtype)
        for rep in range(args.repeat):
            for v in [int(x) for x in args.variants.split(",")]:
                rt.lib().sec_indice_conv_set_variant(v)
                plan = rt.lib().sec_indice_conv_fwd_plan(c, c, 27, n, code, code, 1)
                us, out = time_it(args.iters)
                err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
                print(f"variant={v:3d} plan={plan:2d} layer={args.layer} rows={n} pairs={pairs} us/launch={us:7.2f} "
                      f"alg_GBs={b_alg / us / 1e3:7.1f} frac_of_8TBs={b_alg / us / 1e3 / 8000:.3f} max_rel_err_vs_splitk={err:.2e}", flush=True)
                if args.timeline and plan >= 6 and rep == 0 and hasattr(rt.lib(), "sec__debug_timeline"):
                    import ctypes
                    nw = (n + 127) // 128 * 4
                    buf = torch.zeros((nw, 8), dtype=torch.int64, device=dev)
                    rt.lib().sec__debug_timeline(ctypes.c_void_p(buf.data_ptr()))
                    run()
                    torch.cuda.synchronize()
                    rt.lib().sec__debug_timeline(ctypes.c_void_p(0))
                    t = buf.cpu().numpy().astype(np.float64)
                    t = t[t[:, 3] > 0]
                    z = np.zeros(len(t))                      # s_memtime is per XCD: offsets relative to the XCD's first wave
                    for x in np.unique(t[:, 7]):
                        z[t[:, 7] == x] = t[t[:, 7] == x, 0].min()
                    cols = {"start_offset": t[:, 0] - z, "idx_loads": t[:, 1] - t[:, 0], "offset_loop": t[:, 2] - t[:, 1],
                            "epilogue": t[:, 3] - t[:, 2], "end_offset": t[:, 3] - z, "sum_wait+barrier": t[:, 4],
                            "sum_issue": t[:, 5], "sum_compute": t[:, 6]}
                    for nm, col in cols.items():
                        q = np.percentile(col, [5, 50, 95])
                        print(f"    {nm:18s} p5={q[0]:9.0f} p50={q[1]:9.0f} p95={q[2]:9.0f}  (shader clocks)")
        rt.lib().sec_indice_conv_set_variant(-1)
        return
    us, _ = time_it(args.iters)
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
    print(f"layer={args.layer} rows={n} pairs={pairs} C={c} "
          f"us/launch={us:.2f} alg_GBs={b_alg / us / 1e3:.1f} frac_of_8TBs={b_alg / us / 1e3 / 8000:.3f}")


if __name__ == "__main__":
    main()
