#!/usr/bin/env python3
"""sec_conv2d_nhwc_gather on 8 x 200 x 176 maps with ~2300 sites per frame placed three ways: spread like a KITTI-like cloud's
BEV (blobs), packed into the first tile rows of every frame (live tiles first in launch order), packed into the last rows, none."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd"))
import numpy as np
import torch
from second_amd import ops

rng = np.random.default_rng(0)
B, H, W, N = 8, 200, 176, 2300
w = (torch.randn(128, 128, 3, 3, device="cuda") / 34).bfloat16()
b = torch.randn(128, device="cuda")
pk = ops.conv2d_pack_weight(w[:, ops.gather_channel_perm(64, 2).cuda()].contiguous())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def sites(kind):
    out = []
    for f in range(B):
        if kind == "none":
            continue
        if kind == "blobs":                       # ~55 live tiles per frame: 14 blobs of ~160 sites
            cy, cx = rng.integers(8, H - 8, 14), rng.integers(8, W - 8, 14)
            y = np.clip(np.repeat(cy, N // 14 + 1)[:N] + rng.integers(-6, 7, N), 0, H - 1)
            x = np.clip(np.repeat(cx, N // 14 + 1)[:N] + rng.integers(-10, 11, N), 0, W - 1)
        else:
            rows = 40                             # 5 tile rows x 11 tiles = 55 live tiles
            y = rng.integers(0, rows, N) + (0 if kind == "first" else H - rows)
            x = rng.integers(0, W, N)
        z = rng.integers(0, 2, N)
        out.append(np.stack([np.full(N, f), z, y, x], 1))
    if not out:
        return np.zeros((0, 4), np.int32)
    idx = np.unique(np.concatenate(out), axis=0)
    return idx.astype(np.int32)


for kind in ("blobs", "first", "last", "none"):
    idx = sites(kind)
    n = len(idx)
    ind = torch.from_numpy(idx).cuda()
    feat = torch.randn(max(n, 1), 64, device="cuda").bfloat16()[:n].contiguous()
    smap = ops.sparse_site_map(ind, B, [2, H, W])
    tiles = int((torch.nn.functional.max_pool2d((smap > 0).any(1, keepdim=True).float(), (10, 18), (8, 16), (1, 1)) > 0).sum()) if n else 0
    fn = lambda: ops.conv2d_nhwc_gather(feat, smap, pk, b, 128, relu=True)
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(100):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{kind:6s}: {n:6d} sites, ~{tiles} live tiles of 2200: {e0.elapsed_time(e1) * 10:.1f} us", flush=True)
