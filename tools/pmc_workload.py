#!/usr/bin/env python3
"""The launches whose counters bench.py's roofline objects quote, in ONE process (one torch import per rocprofv3 pass):

    every sparse conv layer of SpMiddleFHD at car.fhd batch 8 (round 4: all ten distinct launches, the way the captured step issues
    them -- static capacity, device-side row count; subm2 64->64 at 56 298 rows / 594 482 pairs is bench.py's `roofline`)
    k_conv2d_halo_reg  RPN 3x3 128->128 on 8 x 200 x 176 (bench.py's `roofline_mfma`)

built exactly as the detector builds them (voxelise -> sorted-numbering strided rulebooks -> SubM rulebook by bitmap rank), ITERS
launches each after WARM warm-ups.  Run under `rocprofv3 --pmc <set> --kernel-trace --output-format csv`; tools/pmc_report.py
turns the pass directories into profiles/rNN_pmc_*.txt and rNN_traffic.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from second_amd import ops, synthetic as syn  # noqa: E402

WARM, ITERS = int(os.environ.get("WARM", "10")), int(os.environ.get("ITERS", "25"))
dev = torch.device("cuda")
ops.set_rulebook_numbering("sorted")                      # the numbering of the device fast path (SecondDetector.forward_points)
clouds = [syn.syn_kitti_cloud(s) for s in range(8)]
if os.environ.get("POINT_ORDER", "shuffle") == "sorted":   # cell-ordered rows (bench.py --point-order sorted): the gather-locality experiment
    import numpy as np
    clouds = [c[np.lexsort((c[:, 0], c[:, 1], c[:, 2]))] for c in clouds]
pts, offs = syn.batch_clouds(clouds)
vox = ops.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev), syn.CAR_FHD_RANGE, syn.CAR_FHD_VOXEL, 5, 40000)
idx, shape = vox["coordinates"].contiguous(), [41, 1600, 1408]
g = torch.Generator(device="cpu").manual_seed(0)
meta = {"launches": []}
# every sparse conv layer of SpMiddleFHD (middle.py:146-189) the way the captured step launches it: static capacity (live rows x 1.25,
# tables padded with -1 rows, device-side row count), packed weights, fused scale / shift / ReLU
LAYERS = [("subm0 4->16", 4, 16, "subm"), ("subm0 16->16", 16, 16, "subm"), ("conv1 16->32 s2", 16, 32, (3, 2, 1)),
          ("subm1 32->32", 32, 32, "subm"), ("subm1 32->32 (2nd)", 32, 32, "subm"), ("conv2 32->64 s2", 32, 64, (3, 2, 1)),
          ("subm2 64->64", 64, 64, "subm"), ("conv3 64->64 s2", 64, 64, (3, 2, (0, 1, 1))), ("subm3 64->64", 64, 64, "subm"),
          ("conv4 64->64 (3,1,1)", 64, 64, ((3, 1, 1), (2, 1, 1), 0))]
in_sites, site_table = None, vox.get("site_table")
sub_cache = None
for name, cin, cout, kind in LAYERS:
    if kind == "subm":
        if sub_cache is None or sub_cache[0] is not idx:
            sub_cache = (idx, ops.rulebook_subm(idx, 8, shape, 3, site_table=site_table))
        nbr, n_in, n_out, ks = sub_cache[1]["nbr_out"], idx.shape[0], idx.shape[0], (3, 3, 3)
    else:
        r = ops.rulebook_conv(idx, 8, shape, *kind, in_sites=in_sites, want_nbr_in=False)
        nbr, n_in, n_out = r["nbr_out"], idx.shape[0], r["num_out"]
        ks = (kind[0],) * 3 if isinstance(kind[0], int) else kind[0]
        idx, shape, in_sites, site_table = r["out_indices"].contiguous(), r["out_shape"], r.get("site_table"), r.get("site_table")
    cap = -(-int(n_out * 1.25) // 256) * 256
    table = torch.full((cap, nbr.shape[1]), -1, dtype=torch.int32, device=dev)
    table[:n_out] = nbr[:n_out]
    n_dev = torch.tensor([n_out, n_out], dtype=torch.int32, device=dev)
    pairs = int((nbr[:n_out] >= 0).sum())
    feat = torch.randn(n_in, cin, generator=g).to(dev).bfloat16()
    w = (torch.randn(*ks, cin, cout, generator=g) / 30).to(dev).bfloat16()
    packed = ops.pack_weight(w)
    scale, shift = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    for _ in range(WARM + ITERS):
        ops.indice_conv(feat, w, table, cap, packed=packed, scale=scale, shift=shift, relu=True, num_out_dev=n_dev)
    torch.cuda.synchronize()
    kvol = nbr.shape[1]
    plan = ops.indice_conv_plan(cin, cout, kvol, cap, torch.bfloat16)
    sig = {11: ops.last_kernel_name(), 12: "k_conv_c4_mfma", 4: f"k_conv_mfma_sk<__hip_bfloat16, __hip_bfloat16, {cin}, {cout}",
           5: f"k_conv_mfma_sks<__hip_bfloat16, __hip_bfloat16, {cin}, {cout}"}.get(plan, ops.last_kernel_name())
    meta["launches"].append({"kernel_signature": sig,
                             "count": WARM + ITERS, "rows": n_out, "pairs": pairs,
                             "alg_bytes": 2 * (pairs * cin + n_out * cout) + 8 * pairs + 2 * kvol * cin * cout,
                             "workload": f"{name}: {n_out} rows, {pairs} pairs, batch 8 synthetic KITTI clouds (seeds 0-7), sorted numbering, capacity {cap}"})
x = torch.relu(torch.randn(8, 128, 200, 176, generator=g)).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
w2 = (torch.randn(128, 128, 3, 3, generator=g) / 34).to(dev).bfloat16()
b2 = torch.randn(128, generator=g).to(dev)
pk = ops.conv2d_pack_weight(w2)
for _ in range(WARM * 5 + ITERS):
    ops.conv2d_nhwc(x, pk, b2, 128, 3, 1, 1, relu=True)
torch.cuda.synchronize()
meta["launches"].append({"kernel_signature": ops.last_kernel_name(), "count": WARM * 5 + ITERS, "flop": 2.0 * 8 * 200 * 176 * 128 * 128 * 9,
                         "workload": "RPN 3x3 128->128 on 8 x 200 x 176, post-ReLU random data"})
out = os.environ.get("PMC_META")
if out:
    with open(out, "w") as f:
        json.dump(meta, f, indent=1)
print(json.dumps(meta))
