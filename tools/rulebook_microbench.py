#!/usr/bin/env python3
"""Strided-rulebook chain of SpMiddleFHD (four SparseConv3d layers, car.fhd batch 8) in both output numberings, each build
replayed from a hipGraph; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.

    python tools/rulebook_microbench.py [--iters 20] [--numbering first_touch,sorted]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd"))
import torch
from second_amd import ops, synthetic as syn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--numbering", default="first_touch,sorted")
    args = ap.parse_args()
    dev = torch.device("cuda")
    pts, offs = syn.batch_clouds([syn.syn_kitti_cloud(s) for s in range(args.batch)])
    vox = ops.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev), syn.CAR_FHD_RANGE, syn.CAR_FHD_VOXEL, 5, 40000)
    idx0, shape0 = vox["coordinates"].contiguous(), [41, 1600, 1408]
    downs = [(3, 2, 1), (3, 2, 1), (3, 2, (0, 1, 1)), ((3, 1, 1), (2, 1, 1), 0)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for mode in args.numbering.split(","):
        idx, shape, sites = idx0, shape0, None
        for li, (ks, st, pd) in enumerate(downs):
            n = idx.shape[0]
            nd = torch.tensor([n], dtype=torch.int32, device=dev)
            cap = n                                            # static-capacity form, as in the captured forward
            run = lambda: ops.rulebook_conv(idx, args.batch, shape, ks, st, pd, 1, n_dev=nd, out_cap=cap, want_nbr_in=False, numbering=mode,
                                            in_sites=sites if mode == "sorted" else None)
            r = run()
            sub = lambda: ops.rulebook_subm(r["out_indices"], args.batch, r["out_shape"], 3, 1, n_dev=r["num_out_dev"], site_table=r["site_table"])
            sub()
            torch.cuda.synchronize()
            res = []
            for fn in (run, sub):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    for _ in range(args.iters):
                        fn()
                g.replay()
                torch.cuda.synchronize()
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) * 1e3 / args.iters)
                del g
            m = int(r["num_out_dev"][0].item())
            print(f"{mode:12s} layer {li}: {n:6d} -> {m:6d} rows  strided build {res[0]:7.2f} us   SubM on its outputs {res[1]:7.2f} us", flush=True)
            idx, shape, sites = r["out_indices"][:m].contiguous(), r["out_shape"], r["site_table"]


if __name__ == "__main__":
    main()
