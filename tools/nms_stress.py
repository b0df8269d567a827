#!/usr/bin/env python3
"""NMS determinism under load: sec_nms_sorted_f32 on fixed inputs, replayed from a hipGraph on one stream while other streams keep
the chip busy with RPN convolutions; keep lists (defined part) and the suppression mask are compared with the first run."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd"))
import torch
from second_amd import ops, runtime as rt, synthetic as syn
from second_amd.models import SecondDetector, CAR_FHD

torch.manual_seed(0)
det = SecondDetector(CAR_FHD).cuda().prepare_inference(torch.bfloat16)
pts, offs = syn.batch_clouds([syn.syn_kitti_cloud(s, num_points=7000, num_voxels=6000) for s in range(3)])
pts, offs = torch.from_numpy(pts).cuda(), torch.from_numpy(offs).cuda()
cap = {}
ops.set_op_hook(lambda name, fn, a, kw, res: cap.setdefault(name, (a, kw, res)))
with torch.no_grad():
    det.forward_points(pts, offs)
ops.set_op_hook(None)
(dets, counts, thr, kind, sem), kw, _ = cap["nms_sorted"]
dets, counts = dets.clone(), counts.clone()
print("dets", tuple(dets.shape), "counts", counts.tolist(), thr, kind, sem, kw)
b, max_n, stride = dets.shape
l = rt.lib()
ws = torch.empty(l.sec_nms_workspace_bytes(b, max_n), dtype=torch.uint8, device="cuda")
keep = torch.empty((b, max_n), dtype=torch.int32, device="cuda")
num_keep = torch.empty((b,), dtype=torch.int32, device="cuda")
s_nms = torch.cuda.Stream()


def nms():
    rc = l.sec_nms_sorted_f32(rt.ptr(dets), rt.ptr(counts), b, max_n, stride, float(thr), {"rotate": 0, "axis_aligned": 1}[kind],
                              {"numba": 0, "cpu": 1}[sem], 1.0, int(kw.get("post_max", 0)), rt.ptr(keep), rt.ptr(num_keep), rt.ptr(ws),
                              ws.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0


with torch.cuda.stream(s_nms):
    nms()
torch.cuda.synchronize()
ref_keep, ref_nk, ref_ws = keep.clone(), num_keep.clone(), ws.clone()
words = (max_n + 63) // 64
x = torch.relu(torch.randn(8, 128, 200, 176, device="cuda")).bfloat16().contiguous(memory_format=torch.channels_last)
w = (torch.randn(128, 128, 3, 3, device="cuda") / 34).bfloat16()
pk = ops.conv2d_pack_weight(w)
bias = torch.randn(128, device="cuda")
xz = torch.zeros_like(x)
_pts, _offs = syn.batch_clouds([syn.syn_kitti_cloud(s) for s in range(8)])
_vox = ops.voxelize(torch.from_numpy(_pts).cuda(), torch.from_numpy(_offs).cuda(), syn.CAR_FHD_RANGE, syn.CAR_FHD_VOXEL, 5, 40000)
sp_nbr = ops.rulebook_subm(_vox["coordinates"].contiguous(), 8, [41, 1600, 1408], 3)["nbr_out"]
sp_feat = torch.randn(sp_nbr.shape[0], 64, device="cuda").bfloat16()
sp_w = (torch.randn(3, 3, 3, 64, 64, device="cuda") / 30).bfloat16()
sp_pk = ops.pack_weight(sp_w)
sp_scale, sp_shift = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
ma = torch.randn(4096, 4096, device="cuda").bfloat16(); mb = torch.randn(4096, 4096, device="cuda").bfloat16()
xf = torch.randn(64 * 1024 * 1024, device="cuda")
load = [torch.cuda.Stream() for _ in range(int(os.environ.get("LOAD", "2")))]
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2000):
    for s in load:
        with torch.cuda.stream(s):
            for _ in range(3):
                if os.environ.get("LOADKIND", "conv") == "conv":
                    ops.conv2d_nhwc(x, pk, bias, 128, 3, 1, 1, relu=True)
                elif os.environ["LOADKIND"] == "convskip":      # all-zero input + zero-tile skip: halo DMA + epilogue, no MFMA loop
                    ops.conv2d_nhwc(xz, pk, bias, 128, 3, 1, 1, relu=True, sparse_input=True)
                elif os.environ["LOADKIND"] == "sparse":        # the row-split sparse conv (same MFMA instruction, 2 waves per SIMD)
                    ops.indice_conv(sp_feat, sp_w, sp_nbr, sp_feat.shape[0], packed=sp_pk, scale=sp_scale, shift=sp_shift, relu=True)
                elif os.environ["LOADKIND"] == "mm":
                    torch.mm(ma, mb)
                else:
                    xf.mul_(1.0001)
    with torch.cuda.stream(s_nms):
        nms()
    torch.cuda.synchronize()
    nk = num_keep.tolist()
    same = torch.equal(num_keep, ref_nk) and all(torch.equal(keep[i, :nk[i]], ref_keep[i, :nk[i]]) for i in range(b))
    if not same:
        bad += 1
        m_now = ws.view(torch.int64).reshape(b, max_n, words)
        m_ref = ref_ws.view(torch.int64).reshape(b, max_n, words)
        diffs = []
        for i in range(b):
            n = int(counts[i])
            nw = (n + 63) // 64
            for r in range(n):
                for c in range(r // 64, nw):
                    if m_now[i, r, c] != m_ref[i, r, c]:
                        diffs.append((i, r, c, hex(int(m_now[i, r, c]) & (2 ** 64 - 1)), hex(int(m_ref[i, r, c]) & (2 ** 64 - 1))))
        if bad <= 6:
            print(f"run {it}: keep differs; num_keep {nk} ref {ref_nk.tolist()}; defined mask words that differ: {diffs[:6]} ({len(diffs)} total)", flush=True)
print("runs with a different keep list:", bad)
if hasattr(l, "sec__debug_nms_counters"):
    h = (ctypes.c_int * 4)()
    l.sec__debug_nms_counters(h, 0)
    print("tile != reloaded box data at the end: column waves", h[0], "row waves", h[1], "| right after the load:", h[2], "| two derivations differ:", h[3])

if hasattr(l, "sec__debug_nms_vals"):
    v = (ctypes.c_float * 256)()
    l.sec__debug_nms_vals(v)
    import numpy as np
    a = np.array(list(v), np.float32).reshape(8, 32)
    np.set_printoptions(precision=7, suppress=False, linewidth=220)
    for r in a[:4]:
        print("c  ", r[0:8]); print("c2 ", r[8:16]); print("lds", r[16:24]); print("d  ", r[24:30], "idx", int(r[30]), "w*1000+lane", int(r[31]))
