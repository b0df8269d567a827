#!/usr/bin/env python3
"""The three launches whose counters bench.py's roofline objects quote, in ONE process (one torch import per rocprofv3 pass):

    k_conv_rows_buf  SubMConv3d 64->64 on the subm2 stage of car.fhd at batch 8 (56 298 rows / 594 482 pairs: bench.py's `roofline`)
    k_conv_rows_buf  the same on the subm3 stage (22 834 rows: the 4-wave form)
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
pts, offs = syn.batch_clouds(clouds)
vox = ops.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev), syn.CAR_FHD_RANGE, syn.CAR_FHD_VOXEL, 5, 40000)
idx, shape = vox["coordinates"].contiguous(), [41, 1600, 1408]
g = torch.Generator(device="cpu").manual_seed(0)
meta = {"launches": []}
in_sites = None
for stage, down in enumerate([(3, 2, 1), (3, 2, 1), (3, 2, (0, 1, 1))]):
    r = ops.rulebook_conv(idx, 8, shape, *down, in_sites=in_sites)
    idx, shape = r["out_indices"].contiguous(), r["out_shape"]
    in_sites = r.get("site_table")
    if stage == 0:
        continue
    rb = ops.rulebook_subm(idx, 8, shape, 3, site_table=r.get("site_table"))
    n = idx.shape[0]
    pairs = int((rb["nbr_out"] >= 0).sum())
    feat = torch.randn(n, 64, generator=g).to(dev).bfloat16()
    w = (torch.randn(3, 3, 3, 64, 64, generator=g) / 30).to(dev).bfloat16()
    packed = ops.pack_weight(w)
    scale, shift = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    for _ in range(WARM + ITERS):
        ops.indice_conv(feat, w, rb["nbr_out"], n, packed=packed, scale=scale, shift=shift, relu=True)
    torch.cuda.synchronize()
    meta["launches"].append({"kernel_signature": ops.last_kernel_name(), "rows": n, "pairs": pairs,
                             "alg_bytes": 2 * (pairs * 64 + n * 64) + 8 * pairs + 2 * 27 * 64 * 64,
                             "workload": f"SubMConv3d 64->64 subm{stage + 1}, batch 8 synthetic KITTI clouds (seeds 0-7), sorted numbering"})
x = torch.relu(torch.randn(8, 128, 200, 176, generator=g)).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
w2 = (torch.randn(128, 128, 3, 3, generator=g) / 34).to(dev).bfloat16()
b2 = torch.randn(128, generator=g).to(dev)
pk = ops.conv2d_pack_weight(w2)
for _ in range(WARM * 5 + ITERS):
    ops.conv2d_nhwc(x, pk, b2, 128, 3, 1, 1, relu=True)
torch.cuda.synchronize()
meta["launches"].append({"kernel_signature": ops.last_kernel_name(), "flop": 2.0 * 8 * 200 * 176 * 128 * 128 * 9,
                         "workload": "RPN 3x3 128->128 on 8 x 200 x 176, post-ReLU random data"})
out = os.environ.get("PMC_META")
if out:
    with open(out, "w") as f:
        json.dump(meta, f, indent=1)
print(json.dumps(meta))
