#!/usr/bin/env python3
"""Headline benchmark: frames/s of the car.fhd VoxelNet forward (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Started under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE set) it runs as one rank of the
job; started plainly (``python bench.py --gpus 8``) it launches its own N ranks through torch.distributed.run on 127.0.0.1 and
relays rank 0's JSON line -- the single-command form of the reference's multi-GPU entry (second/pytorch/train.py:203-206).

One *step* = one pass of the whole hot path over one batch of 8 synthetic KITTI clouds that are already
resident in HBM: points_to_voxel (+SimpleVoxel mean) -> 14 sparse conv layers (rulebooks + fused
indice_conv) -> dense -> RPNV2 (bf16, hand-written MFMA convs) -> decode / top-k / rotated NMS, detections left on the device.
Per-frame data parallel: every rank runs its own batch, no data-path collective ("weak" scaling).
Default launch mode: hipGraph replays with four steps in flight (--inflight 4: four lanes with their own activation buffers on
four streams; every step is still a full pass over its batch).  A step is three graphs -- sparse front end / RPN / predict -- and the
lanes pass a token from RPN segment to RPN segment (--serialize-rpn 1), so that one lane at a time is in its MFMA-bound segment while
the others' latency-bound segments run beside it; --serialize-rpn 0 replays one graph per step on uncoordinated lanes.
`config.single_step_latency_ms` is one step alone as ONE graph, start to finish; `--inflight 1` runs strictly one step at a time.

Prints ONE JSON line on rank 0 with, besides the contract fields,
  roofline      -- the SubMConv3d 64->64 gather-GEMM kernel (the kernel BASELINE.json's metric names): algorithmic bytes
                   per launch / mean launch duration (HIP events around 100 re-issues of the very launch the timed graph
                   runs, right after the timed region), against the 8 TB/s HBM peak; `traffic` = HBM bytes per launch from
                   the committed PMC passes (`traffic_source` names the file), null when no pass matches this launch;
  roofline_mfma -- the RPN 3x3 conv (largest share of the step, MFMA bound), timed the same way;
  kernels       -- the per-launch table of the whole step (voxelise, 8 rulebook builds, 14 sparse convs, dense scatter, RPN
                   convs, predict): algorithmic bytes or FLOPs (SURVEY 8d formulas), launch time, fraction of the bounding peak;
  cpu_baseline  -- the same forward on the host cores through the CPU oracle (kind "port"), rank 0, N=1 only, with a
                   detection-level comparison against the device results of the same frames (`detections_match_cpu`: at least
                   MATCH_MIN_FOUND of the CPU detections present on the device within 0.25 m / 0.05 score AND per-frame counts
                   within MATCH_COUNT_SLACK; the network is a seeded random one with trained-like heads, see build_detector);
  config.*      -- besides the workload: `single_step_latency_ms` (one batch-8 step alone), `e2e_from_pinned_host` (the same
                   timed loop with every step's clouds copied from pinned host memory and its detections copied back; SURVEY
                   8d "end-to-end"; never `value`), `batch1` (one frame per step: the reference's only published figure is
                   batch-1 latency, README.md:27).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver (multi-process runs)
ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "second.pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured-achievable)
BATCH = 8
# detections_match_cpu (bf16 device path vs fp32 CPU forward of the same frames): fraction of the CPU detections that must be
# present on the device (same frame, BEV centre within 0.25 m, score within 0.05) and the allowed per-frame count difference.
# bf16 features move a logit by ~1 % through 21 layers; with nms_iou_threshold 0.01 (car.fhd.config:94: any overlap suppresses)
# two overlapping candidates of near-equal score can swap -- tests/test_gpu_e2e.py attributes every miss to its cause.
MATCH_MIN_FOUND = 0.9      # (0.85 until round 5: observed 90 of 96 on every box; the fp32 device path of the same line must match 96 of 96)
MATCH_COUNT_SLACK = 2


class ConvCapture:
    """Captures the arguments of the first selected sec_indice_conv_fwd launch of a forward pass so that the
    very same launch (same tensors, same rulebook) can be re-issued back-to-back between two HIP events."""

    def __init__(self, select):
        self.select, self.call, self.enabled = select, None, False

    def begin(self, meta):
        if self.enabled and self.call is None and self.select(meta):
            self.call = meta
        return None

    def end(self, token):
        pass


def time_kernel(call, reps=100):
    """Mean duration of one launch: `reps` back-to-back launches on the current stream between two HIP events
    (the kernel is ~20 us, far above the ~5 us host launch cost, so the stream never drains).  Also returns the kernel
    instantiation the dispatcher took (sec_last_kernel_name), read right after the launches."""
    from second_amd import ops
    a = call["args"]
    for _ in range(5):
        ops.indice_conv(*a["pos"], **a["kw"])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    for _ in range(reps):
        ops.indice_conv(*a["pos"], **a["kw"])
    e1.record(torch.cuda.current_stream())
    name = ops.last_kernel_name()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps, name


def time_rpn_conv(det, batch, reps=100):
    """The RPN's 3x3 128->128 conv (k_conv2d_halo_reg, the largest single share of the step) re-issued `reps` times between
    two HIP events on the launch stream, on an input of the live shape: achieved TFLOP/s against the dense bf16 MFMA peak."""
    from second_amd import ops
    rpn = det.rpn
    if not getattr(rpn, "use_hip", False):
        return None
    _, h, w = det.feature_map_size
    # post-ReLU-like activations (half zeros), as between the RPN layers
    x = torch.relu(torch.randn(batch, 128, h, w, device="cuda")).to(rpn.ws[1].dtype).contiguous(memory_format=torch.channels_last)
    wgt, pk, b = rpn.ws[1], rpn.packed[1], rpn.bs[1]
    if tuple(wgt.shape) != (128, 128, 3, 3):
        return None
    fn = lambda: ops.conv2d_nhwc(x, pk, b, 128, 3, 1, 1, relu=True)
    for _ in range(200):          # ~20 ms: the first launches after an idle gap run at low clocks
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    for _ in range(reps):
        fn()
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    flop = 2.0 * batch * h * w * 128 * 128 * 9
    return {"bound": "mfma", "achieved": round(flop / t / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
            "frac": round(flop / t / 2.5e15, 4), "kernel": f"{ops.last_kernel_name()} (RPN 3x3 128->128, 5 launches per step + the gathered first layer)",
            "launch_us": round(t * 1e6, 2), "launches_timed": reps, "flop_per_launch": flop}


PLAN_NAMES = {0: "k_conv_generic", 1: "k_conv_tiled", 2: "k_conv_c4", 3: "k_conv_mfma", 4: "k_conv_mfma_sk", 5: "k_conv_mfma_sks",
              6: "k_conv_rows", 7: "k_conv_rows", 8: "k_conv_rows", 9: "k_conv_rows", 10: "k_conv_rows_reg", 11: "k_conv_rows_buf", 12: "k_conv_c4_mfma", 13: "k_conv_rows_m2", 14: "k_conv_rows_lds", 15: "k_conv_rows_ks"}


def kernel_table(det, points, offsets, reps=30):
    """Every launch group of ONE static forward, re-issued `reps` times between two HIP events on the launch stream, with the
    algorithmic bytes / FLOPs of SURVEY 8(d):  voxelise 4F*Npts + Nvox*(4F*T+16); SubM rulebook 16N + 8P + 4K; strided rulebook
    16Nin + 16Nout + 8P + 4K; indice_conv s(P*Cin + Nout*Cout) + 8P + sK*Cin*Cout; dense s*N*C + 16N + s*B*C*D*H*W;
    conv2d 2*B*Ho*Wo*Cin*Cout*k^2 FLOP (MFMA bound).  A rulebook / voxelise entry is a chain of several kernels: the time is
    the chain's, the fraction is algorithmic bytes over it."""
    from second_amd import ops
    calls = []
    ops.set_op_hook(lambda name, fn, a, kw, res: calls.append((name, fn, a, kw, res)))
    with torch.no_grad():
        det.forward_points(points, offsets, static=True)
    torch.cuda.synchronize()
    ops.set_op_hook(None)
    prev_numbering = ops.set_rulebook_numbering(det.rulebook_numbering)   # the re-issues below run outside forward_points: same numbering as the pass
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def site_note(kw, key):
        t = kw.get(key)
        if t is None:
            return ""
        tag = t[0] if isinstance(t[0], str) else "hash"
        return {"vox": " (site lookup: the voxeliser's hash table)", "sorted": " (site lookup: bitmap ranks of the strided build)" if key == "site_table"
                else " (output bitmap derived from the input sites' bitmap)", "hash": " (reuses the strided build's hash table)"}[tag]

    def live(t, cap):
        return cap if t is None else int(t.reshape(-1)[0].item())

    def elt(dt):
        return 4 if dt == torch.float32 else 2
    def time_reissue(fn, a, kw):
        """`reps` re-issues captured in a hipGraph (no host launch cost between them: several entries are 5-15 us chains, below
        the ~10 us a Python -> ctypes launch costs), replayed between two events; eager re-issue if capture is refused."""
        for _ in range(2):
            fn(*a, **kw)
        torch.cuda.synchronize()
        try:
            g = torch.cuda.CUDAGraph()
            with ops.rt.capture_guard(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                for _ in range(reps):
                    fn(*a, **kw)
            g.replay()
            torch.cuda.synchronize()
            e0.record(torch.cuda.current_stream())
            g.replay()
            e1.record(torch.cuda.current_stream())
            torch.cuda.synchronize()
            del g
            return e0.elapsed_time(e1) * 1e3 / reps, "graph"
        except Exception:   # noqa: BLE001 -- a host sync inside the op: time it eagerly
            torch.cuda.synchronize()
            e0.record(torch.cuda.current_stream())
            for _ in range(reps):
                fn(*a, **kw)
            e1.record(torch.cuda.current_stream())
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps, "eager"

    rows = []
    for name, fn, a, kw, res in calls:
        us, how = time_reissue(fn, a, kw)
        ent = {"op": name, "us": round(us, 2)}
        if how != "graph":
            ent["timed"] = how
        if name == "voxelize":
            pts = a[0]
            nv = int(res["voxel_offsets"][-1].item())
            f = pts.shape[1]
            if res.get("voxels") is not None:
                t = res["voxels"].shape[1]
                ent.update(bytes=4 * f * pts.shape[0] + nv * (4 * f * t + 16), detail=f"{pts.shape[0]} points -> {nv} voxels")
            else:            # fill=False: point lists only, the [rows, T, F] tensor is never written (pfn_forward_slots reads the lists)
                ent.update(bytes=4 * f * pts.shape[0] + nv * 20 + 4 * pts.shape[0],
                           detail=f"{pts.shape[0]} points -> {nv} voxels (point lists only, no voxel tensor)")
        elif name == "rulebook_subm":
            n = live(kw.get("n_dev"), a[0].shape[0])
            p_ = int((res["nbr_out"][:n] >= 0).sum().item())
            k = res["nbr_out"].shape[1]
            ent.update(bytes=16 * n + 8 * p_ + 4 * k, detail=f"subm {n} rows {p_} pairs" + site_note(kw, "site_table"))
        elif name == "rulebook_chain":
            # all rulebooks of the stack from one fused build: the sum of the per-layer formulas above over its levels
            lv = res["levels"]
            tot, parts = 0, []
            for li, L in enumerate(lv):
                n = live(L["num_dev"], L["cap"])
                if L["subm_nbr"] is not None:
                    p_ = int((L["subm_nbr"][:n] >= 0).sum().item())
                    tot += 16 * n + 8 * p_ + 4 * 27
                    parts.append(f"subm{li} {n} rows {p_} pairs")
                if li >= 1:
                    n_in = live(lv[li - 1]["num_dev"], lv[li - 1]["cap"])
                    p_ = int((L["nbr_out"][:n] >= 0).sum().item())
                    tot += 16 * n_in + 16 * n + 8 * p_ + 4 * L["nbr_out"].shape[1]
                    parts.append(f"conv{li} {n_in} -> {n} rows {p_} pairs")
            ent.update(bytes=tot, detail="fused chain (sorted numbering): " + "; ".join(parts) + ("; + BEV site map" if res["site_map"] is not None else ""))
        elif name == "rulebook_conv":
            n = live(kw.get("n_dev"), a[0].shape[0])
            m = live(res["num_out_dev"], res["num_out"])
            p_ = int((res["nbr_out"][:m] >= 0).sum().item())
            k = res["nbr_out"].shape[1]
            ent.update(bytes=16 * n + 16 * m + 8 * p_ + 4 * k, detail=f"strided {n} -> {m} rows {p_} pairs, {det.rulebook_numbering} numbering" + site_note(kw, "in_sites"))
        elif name == "indice_conv":
            feat, w, nbr, cap = a[:4]
            m = live(kw.get("num_out_dev"), cap)
            p_ = int((nbr[:m] >= 0).sum().item())
            cin, cout = w.shape[-2], w.shape[-1]
            k = w.numel() // (cin * cout)
            s_ = elt(feat.dtype)
            plan = ops.indice_conv_plan(cin, cout, k, cap, feat.dtype, kw.get("out_dtype") or feat.dtype, kw.get("packed") is not None)
            ent.update(bytes=s_ * (p_ * cin + m * cout) + 8 * p_ + s_ * k * cin * cout, flop=2.0 * p_ * cin * cout,
                       kernel=(ops.last_kernel_name() if plan in (11, 13, 14, 15) else "") or PLAN_NAMES.get(plan, str(plan)),
                       detail=f"{cin}->{cout} k{k} {m} rows {p_} pairs")
        elif name == "pfn_forward":
            if isinstance(a[1], dict):           # pfn_forward_slots(points, vox, ...): pillars read through the voxeliser's point lists
                vd = a[1]
                n = live(kw.get("num_dev"), vd["coordinates"].shape[0])
                t_, f_ = int(vd["site_table"][4]), a[0].shape[1]
            else:
                vox_, npv = a[0], a[1]
                n = live(kw.get("num_dev"), vox_.shape[0])
                t_, f_ = vox_.shape[1], vox_.shape[2]
            # every point slot of a live pillar read once (4F bytes), 64 output channels written once
            ent.update(bytes=4 * f_ * t_ * n + 4 * n + 16 * n + elt(res.dtype) * res.shape[1] * n, flop=2.0 * n * t_ * (f_ + 5) * res.shape[1],
                       detail=f"PillarFeatureNet {n} pillars x {t_} point slots x {f_} -> {res.shape[1]} channels")
        elif name == "pillar_scatter":
            n = live(kw.get("num_dev"), a[0].shape[0])
            c = a[0].shape[1]
            ent.update(bytes=elt(a[0].dtype) * n * c + 16 * n + elt(a[0].dtype) * res.numel(), detail=f"{n} pillars x {c} -> {tuple(res.shape)} (zero fill included)")
        elif name == "voxel_block_filter":
            vin = a[0]
            nv = live(vin["voxel_offsets"][-1:], vin["voxels"].shape[0])
            t_, f_ = vin["voxels"].shape[1], vin["voxels"].shape[2]
            kept = live(res["voxel_offsets"][-1:], res["voxels"].shape[0])
            ent.update(bytes=nv * (4 * f_ * t_ + 20) + kept * (4 * f_ * t_ + 20), detail=f"block filter {nv} -> {kept} voxels")
        elif name == "sparse_to_dense":
            n = live(kw.get("num_dev"), a[0].shape[0])
            c = a[0].shape[1]
            ent.update(bytes=elt(a[0].dtype) * n * c + 16 * n + elt(a[0].dtype) * res.numel(), detail=f"{n} rows x {c} -> {tuple(res.shape)}")
        elif name == "conv2d_nhwc":
            x, cout, ks = a[0], a[3], a[4]
            b_, cin, _, _ = x.shape
            ent.update(flop=2.0 * b_ * res.shape[2] * res.shape[3] * cin * cout * ks * ks,
                       bytes=elt(x.dtype) * (x.numel() + res.numel()), detail=f"{cin}->{cout} k{ks} {tuple(x.shape[2:])}")
        elif name == "pillar_site_map":
            n = live(kw.get("num_dev"), a[0].shape[0])
            ent.update(bytes=16 * n + 4 * n + 4 * res.numel(), detail=f"{n} pillars -> map {tuple(res.shape)} (zero fill included)")
        elif name == "conv2d_nhwc_rows":
            feat, smap, cout, ks = a[0], a[1], a[4], a[5]
            n = int((smap > 0).sum().item())
            cin = feat.shape[1]
            # FLOPs of the dense-equivalent conv (empty tiles are skipped, not counted separately); bytes: map + pillar rows + output
            ent.update(flop=2.0 * res.shape[0] * res.shape[2] * res.shape[3] * cin * cout * ks * ks,
                       bytes=4 * smap.numel() + elt(feat.dtype) * n * cin + elt(res.dtype) * res.numel(),
                       detail=f"{cin}->{cout} k{ks} {tuple(smap.shape[1:])} gathered from {n} pillar rows (no canvas)")
        elif name == "sparse_site_map":
            n = live(kw.get("num_dev"), a[0].shape[0])
            ent.update(bytes=16 * n + 4 * res.numel(), detail=f"{n} sites -> map {tuple(res.shape)}")
        elif name == "conv2d_nhwc_gather":
            feat, smap, cout = a[0], a[1], a[4]
            b_, _, h_, w_ = smap.shape
            n = int((smap > 0).sum().item())
            # the kernel skips every 8 x 16 output tile whose 10 x 18 halo holds no site (it writes act(bias) there): the MFMA
            # fraction is quoted on the LIVE tiles' FLOPs only (128 pixels x 128 x cout x 9 MACs each); bytes: map + rows + output
            occ = (smap > 0).any(dim=1, keepdim=True).float()
            th, tw = -(-h_ // 8), -(-w_ // 16)
            occ = torch.nn.functional.pad(occ, (1, 16 * tw - w_ + 1, 1, 8 * th - h_ + 1))
            live = int(torch.nn.functional.max_pool2d(occ, (10, 18), (8, 16)).sum().item())
            ent.update(flop=2.0 * live * 128 * 128 * cout * 9, bytes=4 * smap.numel() + elt(feat.dtype) * n * 64 + elt(feat.dtype) * res.numel(),
                       live_tiles=live, tiles=b_ * th * tw, dense_equivalent_flop=2.0 * b_ * h_ * w_ * 128 * cout * 9,
                       detail=f"128->{cout} k3 ({h_}, {w_}) gathered from {n} sparse rows (no dense image); {live} of {b_ * th * tw} "
                              f"tiles hold sites, the others are written from the bias vector")
        elif name == "conv2d_nhwc_tiles":
            x, cout, tl = a[0], a[3], a[4]
            b_, cin, h_, w_ = x.shape
            lv, tot = int(a[5].sum().item()), tl.numel()
            copied = a[6] is not None        # lazy consumers: no background tile is written
            # MFMA fraction on the LIVE tiles' FLOPs (128 pixels x 128 x cout x 9 MACs each); background tiles are one 4 KB store each
            ent.update(flop=2.0 * lv * 128 * cin * cout * 9,
                       bytes=elt(x.dtype) * (x.numel() * lv // tot + (res.numel() if copied else res.numel() * lv // tot)), live_tiles=lv, tiles=tot,
                       dense_equivalent_flop=2.0 * b_ * h_ * w_ * cin * cout * 9,
                       detail=f"{cin}->{cout} k3 ({h_}, {w_}); {lv} of {tot} tiles are within reach of a site and are convolved, "
                              + ("the others are copied from the empty frame's output" if copied else
                                 "the others are not written (the next conv reads them from the empty frame's map)"))
        elif name == "conv2d_nhwc_tiles_tail":
            x, tl = a[0], a[3]
            b_, cin, h_, w_ = x.shape
            lv, tot = int(a[4].sum().item()), tl.numel()
            # the last 3x3 conv with the 1x1 tail in its epilogue: FLOPs of both on the live tiles; the conv's output never reaches memory
            ent.update(flop=2.0 * lv * 128 * (cin * 128 * 9 + 128 * (128 + res.shape[1])),
                       bytes=elt(x.dtype) * (x.numel() * lv // tot + res.numel() * lv // tot), live_tiles=lv, tiles=tot,
                       dense_equivalent_flop=2.0 * b_ * h_ * w_ * (cin * 128 * 9 + 128 * (128 + res.shape[1])),
                       detail=f"{cin}->128 k3 ({h_}, {w_}) + fused 1x1 tail 128->128->{res.shape[1]} in the epilogue on {lv} of {tot} tiles; the conv's "
                              "output stays in LDS, the other tiles of the head map are not written (select / decode read them from the empty frame's head map)")
        elif name == "rpn_tile_live":
            smap = a[0]
            ent.update(bytes=4 * smap.numel() + 2 * res[0].numel(), detail=f"site map {tuple(smap.shape)} -> live-tile maps of {a[1]} conv layers")
        elif name == "conv1x1_chain":
            x = a[0]
            if kw.get("live_counts") is not None:      # only the live tiles (128 pixels each) are computed, the others copied
                lv, tot = int(kw["live_counts"].sum().item()), kw["tile_order"].numel()
                lazy_out = kw.get("background") is None      # lazy heads: the other tiles are not written (predict reads them from the empty frame's map)
                ent.update(flop=2.0 * lv * 128 * 128 * (128 + res.shape[1]),
                           bytes=elt(x.dtype) * (x.numel() * lv // tot + (res.numel() * lv // tot if lazy_out else res.numel())),
                           live_tiles=lv, tiles=tot,
                           detail=f"128->128->{res.shape[1]} 1x1 on {lv} of {tot} tiles, the others "
                                  + ("not written (select / decode read them from the empty frame's head map)" if lazy_out
                                     else "copied from the empty frame's output"))
            else:
                ent.update(flop=2.0 * x.shape[0] * x.shape[2] * x.shape[3] * 128 * (128 + res.shape[1]), bytes=elt(x.dtype) * (x.numel() + res.numel()),
                           detail=f"128->128->{res.shape[1]} 1x1")
        if "flop" in ent and name.startswith("conv"):
            ent["bound"], ent["frac"] = "mfma", round(ent["flop"] / (us * 1e-6) / 2.5e15, 4)
        elif "bytes" in ent:
            ent["bound"], ent["frac"] = "hbm", round(ent["bytes"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            if ent["frac"] > 1.0:   # possible only through cache reuse -- every pair's row counts in the algorithmic bytes
                ent["note"] = "above the HBM peak on ALGORITHMIC bytes: the gathered rows of a dense rulebook are re-read from L2 / Infinity Cache"
        rows.append(ent)
    ops.set_rulebook_numbering(prev_numbering)
    return rows


# Other BASELINE configs (parity-test cases; timed only on request with --workload, never the default line):
#   nusc.pp  = nuscenes/all.pp.largea (PointPillars), nusc.fhd = nuscenes/all.fhd (block-filtered voxels, 10 classes)
WORKLOADS = {
    "car.fhd": dict(cfg="CAR_FHD", batch=8, metric="frames/sec VoxelNet fwd (car.fhd, ~16k active voxels)",
                    desc="car.fhd.config VoxelNet forward (voxelise+VFE+SpMiddleFHD+RPNV2+rotated NMS), inference, "
                         "batch=8 synthetic KITTI clouds/GPU (17000 pts, 16000 voxels each), random-init weights, "
                         "inputs resident in HBM"),
    # BASELINE configs 4 / 5 at their stated input sizes: ~300k points per 10-sweep cloud (SURVEY 8d SYN-NUSC); the live pillar /
    # voxel counts of the run are reported in config.rows_per_frame
    "nusc.pp": dict(cfg="ALL_PP_LARGEA", batch=4, points=300000, dtype="bf16", metric="frames/sec VoxelNet fwd (nuscenes/all.pp.largea)",
                    desc="nuscenes/all.pp.largea VoxelNet forward (voxelise+PillarFeatureNet+scatter+RPNV2 3 blocks+"
                         "axis-aligned NMS), inference, batch=4 synthetic 10-sweep NuScenes clouds/GPU (~300k pts each), "
                         "random-init weights, inputs resident in HBM"),
    "nusc.fhd": dict(cfg="ALL_FHD_NUSC", batch=4, points=300000, dtype="fp16", metric="frames/sec VoxelNet fwd (nuscenes/all.fhd)",
                     desc="nuscenes/all.fhd VoxelNet forward (block-filtered voxelise+SpMiddleFHD on 1984x1984x40+RPNV2+"
                          "axis-aligned NMS), inference, fp16 (BASELINE config 5), batch=4 synthetic 10-sweep NuScenes clouds/GPU "
                          "(~300k pts each), random-init weights, inputs resident in HBM"),
}
WORKLOADS["car.fhd.train"] = dict(cfg="CAR_FHD", batch=4, metric="samples/sec VoxelNet training step (car.fhd, batch 4/GPU)",
                                  desc="car.fhd.config training step (voxelise + target assignment + SpMiddleFHD/RPNV2 forward, focal / "
                                       "smooth-L1 / direction loss, backward, one flat-bucket gradient all-reduce, clip, AdamW), "
                                       "batch=4 synthetic KITTI clouds/GPU (17000 pts, 16000 voxels, 12 ground-truth boxes each), "
                                       "random-init weights, inputs resident in HBM")
WORKLOADS["nusc.fhd.train"] = dict(cfg="ALL_FHD_NUSC", batch=3, points=300000, dtype="fp16",
                                   metric="samples/sec VoxelNet training step (nuscenes/all.fhd, batch 3/GPU)",
                                   desc="nuscenes/all.fhd.config training step (BASELINE config 5: ten classes, per-class target "
                                        "assignment, fp16 features over fp32 master weights, DDP), batch=3 synthetic 10-sweep clouds/GPU "
                                        "(~300k pts each, 24 ground-truth boxes over the ten classes), random-init weights, inputs "
                                        "resident in HBM")
WORKLOADS["nusc.pp.train"] = dict(cfg="ALL_PP_LARGEA", batch=3, points=300000, dtype="bf16",
                                  metric="samples/sec VoxelNet training step (nuscenes/all.pp.largea, batch 3/GPU)",
                                  desc="nuscenes/all.pp.largea.config training step (PointPillars: PillarFeatureNet in its torch formulation, "
                                       "differentiable pillar scatter, three-block RPN, assign_all targets with per-anchor thresholds), "
                                       "batch=3 synthetic 10-sweep clouds/GPU (~300k pts each, 24 ground-truth boxes), random-init weights, "
                                       "inputs resident in HBM")
WL = WORKLOADS["car.fhd"]


def train_bench(args, rank, local_rank, world, device):
    """BASELINE config 3: DDP training, one process per GPU, ONE all-reduce of the flat gradient bucket per step (RCCL over
    xGMI); weak scaling (batch 4 per GPU).  Default fp32 throughout (the reference's training precision); --dtype bf16 / fp16 =
    16-bit features over fp32 master weights (BASELINE config 5 names fp16)."""
    import torch.distributed as dist
    from second_amd import synthetic as syn
    from second_amd.models import SecondDetector, CAR_FHD
    from second_amd.training import DeviceTrainer
    from second_amd import models as M
    bs = WL["batch"]
    cfg = getattr(M, WL["cfg"])
    gcls = None
    if WL["cfg"] == "CAR_FHD":
        clouds = [syn.syn_kitti_cloud(rank * bs + s) for s in range(bs)]
        boxes = [syn.syn_kitti_boxes(rank * bs + s, 12) for s in range(bs)]
    else:
        # ten-class ground truth: class-sized boxes (the class's anchor size, scaled a little) at random places and headings
        r = cfg["point_cloud_range"]
        clouds = [syn.syn_nusc_cloud(rank * bs + s, num_points=WL["points"], point_cloud_range=tuple(r), scene="urban") for s in range(bs)]
        boxes, classes = [], []
        for s in range(bs):
            g = np.random.default_rng(1000 + rank * bs + s)
            k = 24
            cls = g.integers(1, len(cfg["anchor_groups"]) + 1, k)      # classes that own an anchor generator
            size = np.array([cfg["anchor_sizes"][cfg["anchor_groups"][c - 1][0]] for c in cls], np.float32) * g.uniform(0.9, 1.1, (k, 3))
            z = np.array([cfg["anchor_ranges"][cfg["anchor_groups"][c - 1][0]][2] for c in cls], np.float32)
            xy = g.uniform(r[0] + 5, r[3] - 5, (k, 2))
            boxes.append(np.concatenate([xy, z[:, None], size, g.uniform(-np.pi, np.pi, (k, 1))], 1).astype(np.float32))
            classes.append(cls.astype(np.int32))
        gcls = torch.from_numpy(np.concatenate(classes)).to(device)
    pts, offs = syn.batch_clouds(clouds)
    gt = np.concatenate(boxes).astype(np.float32)
    goffs = np.cumsum([0] + [len(b) for b in boxes]).astype(np.int32)
    pts, offs, gt, goffs = (torch.from_numpy(a).to(device) for a in (pts, offs, gt, goffs))
    torch.manual_seed(0)
    det = SecondDetector(cfg).to(device)
    amp = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": None, "fp32_exact": None}[args.dtype]
    tr = DeviceTrainer(det, amp_dtype=amp)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    step_fn, launch = (lambda: tr.step(pts, offs, gt, goffs, gcls)), "eager"
    if args.mode == "graph" and amp is not None and tr.loss_scale is None and not det.pillars:
        try:        # the whole step as one hipGraph (DeviceTrainer.capture_step); anything that cannot be captured keeps the eager step
            replay = tr.capture_step(pts, offs, gt, goffs, gcls)
            step_fn, launch = (lambda: replay()), "one hipGraph per step (static-capacity rows)"
        except Exception as e:  # noqa: BLE001
            print(f"[bench] whole-step capture failed ({e!r}); eager steps", file=sys.stderr)
            tr.static = False
    elapsed, timing, out6 = measure(step_fn, barrier, args.steps, args.warmup, world, device)
    if tr.static:
        tr.check_overflow()
    # the gradient all-reduce alone (the only collective of the step): 20 back-to-back reductions of the live bucket
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        tr.bucket.allreduce(average=True)
    e1.record()
    torch.cuda.synchronize()
    ar_us = e0.elapsed_time(e1) * 1e3 / 20     # world 1: no collective runs, this is the bucket's pack / unpack cost only
    if rank == 0:
        losses = tr.loss_dict()
        res = {"metric": WL["metric"], "value": round(bs * args.steps * world / elapsed, 2), "unit": "samples/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "timing": timing,
               "dtype": "fp32 (IEEE fp32 products forward and backward: spconv.functional.TRAIN_FP32_MODE = exact)" if amp is None else f"{args.dtype} features (sparse stack + RPN autocast) over fp32 master weights", "data": "synthetic",
               "config": {"workload": WL["desc"], "samples_per_step_per_gpu": bs, "parallelism": f"ddp{world}",
                          "points_per_frame": int(pts.shape[0]) // bs, "launch_mode": launch,
                          "gradient_bucket_bytes": tr.bucket.numel * 4,
                          "allreduce_us": round(ar_us, 1) if world > 1 else None, "bucket_pack_unpack_us": round(ar_us, 1) if world == 1 else None,
                          "skipped_steps_loss_scale": tr.skipped_steps if (tr.loss_scale is not None or tr.loss_scale_dev is not None) else None,
                          "optimizer": "AdamW (adam + fixed weight decay 0.01, car.fhd.config:180-188)",
                          "target_assignment": "per anchor range" if tr.class_ranges else "single class"},
               "roofline": None, "cpu_baseline": None, "loss_last_step": {k: round(v, 5) for k, v in losses.items()}}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()



def _scene_clouds(scene, seeds):
    """the dense scene's generator takes ~3 s per frame: its clouds are cached under /tmp for the child runs of one bench"""
    from second_amd import synthetic as syn
    if scene == "open":
        return [syn.syn_kitti_cloud(s) for s in seeds]
    out = []
    for s in seeds:
        path = f"/tmp/second_amd_scene_{scene}_{s}.npy"
        try:
            c = np.load(path)
        except (OSError, ValueError):
            c = syn.syn_kitti_cloud(s, scene=scene)
            try:
                np.save(path, c)
            except OSError:
                pass
        out.append(c)
    return out


def build_inputs(rank, device, order="shuffle", scene="open"):
    from second_amd import synthetic as syn
    if WL["cfg"] != "CAR_FHD":
        rng = (-50, -50, -5, 50, 50, 3) if WL["cfg"] == "ALL_PP_LARGEA" else (-49.6, -49.6, -5, 49.6, 49.6, 3)
        clouds = [syn.syn_nusc_cloud(rank * BATCH + s, num_points=WL["points"], point_cloud_range=rng, scene="urban") for s in range(WL["batch"])]
        pts, offs = syn.batch_clouds(clouds)
        return clouds, torch.from_numpy(pts).to(device), torch.from_numpy(offs).to(device)
    clouds = _scene_clouds(scene, [rank * BATCH + s for s in range(WL["batch"])])
    if order == "sorted":   # experiment: points in spatial (z, y, x) order instead of the shuffled order of SURVEY 8d
        clouds = [c[np.lexsort((c[:, 0], c[:, 1], c[:, 2]))] for c in clouds]
    elif order == "scan":   # firing order: azimuth step of 0.16 degrees (the generator's), then elevation (beam) -- real .bin files are un-shuffled
        def scan(c):
            az = np.round(np.degrees(np.arctan2(c[:, 1], c[:, 0])) / 0.16).astype(np.int64)
            el = np.arctan2(c[:, 2], np.hypot(c[:, 0], c[:, 1]))
            return c[np.lexsort((-el, az))]
        clouds = [scan(c) for c in clouds]
    pts, offs = syn.batch_clouds(clouds)
    return clouds, torch.from_numpy(pts).to(device), torch.from_numpy(offs).to(device)


def build_detector(device, dtype, calib_cloud=None, exact=False):
    """Seeded random weights of the configured architecture.  car.fhd: made to BEHAVE like a trained detector where the
    selection stages can tell the difference (second_amd.synthetic.randomise_like_trained / sharpen_heads: empty regions of the
    map carry zero activations and score below nms_score_threshold, candidate scores are distinct) -- with default-initialised
    heads ~35 000 anchors per frame tie at the top score and top-k / NMS are decided by tie order, which no two arithmetic
    paths break alike.  The heads are calibrated on ``calib_cloud`` through the fp32 DEVICE forward; the CPU baseline loads
    the resulting state dict.  Shapes, layer count and the timed launches are those of the default-initialised network."""
    from second_amd import models, synthetic as syn
    from second_amd.models import SecondDetector
    torch.manual_seed(0)
    det = SecondDetector(getattr(models, WL["cfg"]))
    if WL["cfg"] == "CAR_FHD" and calib_cloud is not None:
        syn.randomise_like_trained(det, seed=1)
        det = det.eval().to(device)
        pts, offs = syn.batch_clouds([calib_cloud])
        with torch.no_grad():
            vox = det.voxel_generator.generate_device(torch.from_numpy(pts).to(device), torch.from_numpy(offs).to(device), mean_features=4)
            preds = det.network_forward(vox["mean"], vox["coordinates"], 1)
            syn.sharpen_heads(det, preds["cls_preds"].float(), preds["box_preds"].float())
    else:
        g = torch.Generator().manual_seed(1)
        for m in det.modules():  # BN in eval mode with non-trivial statistics so that folding is exercised
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.running_mean.copy_(torch.empty_like(m.running_mean).uniform_(-0.1, 0.1, generator=g))
                m.running_var.copy_(torch.empty_like(m.running_var).uniform_(0.5, 1.5, generator=g))
        det = det.eval().to(device)
    cpu_state = {k: v.detach().cpu().clone() for k, v in det.state_dict().items()}
    # fp32: BatchNorms folded; products as three bf16 MFMA passes on split operands ("bf16x3") unless exact (IEEE fp32 products)
    det.prepare_inference(dtype, exact=exact)
    return det, cpu_state


# ------------------------------------------------------------------------------------------ timing harness
SELF_WARM_MIN_S = 1.0      # keep replaying after the --warmup steps until the clocks have ramped ...
SELF_WARM_MAX_S = 4.0      # ... two consecutive 50-step windows agree within 2 %, or this much time has passed
TIMED_MIN_S = 0.5          # repeat windows of exactly --steps steps until this much time is covered (at least 5 windows)


def measure(step, barrier, steps, warmup, world=1, device=None):
    """The timed region of every workload.  `warmup` untimed steps as the contract asks, then the harness warms ITSELF -- 50-step
    windows until >= SELF_WARM_MIN_S have passed and two consecutive windows agree within 2 % (a fresh box ramps its clocks over
    the first second; `--warmup 5` alone is 3 ms of work) -- and then times WINDOWS of exactly `steps` steps, each bracketed by
    barrier + torch.cuda.synchronize() on both sides, until they cover >= TIMED_MIN_S (5 ... 200 windows).  Per window the time
    is the MAX over ranks; the reported figure is the MEDIAN window, the spread is reported beside it.  Every rank runs the same
    number of windows (the continue / stop decisions are all-reduced)."""
    import torch.distributed as dist

    def agree(x, op=None):       # the same decision on every rank
        if world <= 1:
            return x
        t = torch.tensor([float(x)], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=op or dist.ReduceOp.MAX)
        return float(t.item())

    def window(k):
        barrier()
        t0 = time.perf_counter()
        r = None
        for _ in range(k):
            r = step()
        barrier()
        return agree(time.perf_counter() - t0), r

    r = None
    for _ in range(warmup):
        r = step()
    t_start, prev, warm_windows = time.perf_counter(), None, 0
    while SELF_WARM_MAX_S > 0:
        t, r = window(50)
        warm_windows += 1
        spent = agree(time.perf_counter() - t_start)
        if (spent >= SELF_WARM_MIN_S and prev is not None and abs(t - prev) <= 0.02 * prev) or spent >= SELF_WARM_MAX_S:
            break
        prev = t
    times = []
    t, r = window(steps)
    times.append(t)
    n = int(min(200, max(5, -(-TIMED_MIN_S // max(t, 1e-6))))) if TIMED_MIN_S > 0 else 1
    for _ in range(n - 1):
        t, r = window(steps)
        times.append(t)
    while TIMED_MIN_S > 0 and float(np.sum(times)) < TIMED_MIN_S and len(times) < 400:    # the first window was a slow one: keep going
        t, r = window(steps)
        times.append(t)
    med = float(np.median(times))
    info = {"windows": len(times), "steps_per_window": steps, "timed_s": round(float(np.sum(times)), 4),
            "ms_per_step_median": round(med / steps * 1e3, 4), "ms_per_step_min": round(min(times) / steps * 1e3, 4),
            "ms_per_step_max": round(max(times) / steps * 1e3, 4), "spread_pct": round((max(times) - min(times)) / med * 100, 2),
            "self_warm_windows_of_50": warm_windows,
            "what": "value = work of one window / MEDIAN window time; every window is exactly --steps steps between barrier + synchronize"}
    return med, info, r


# ------------------------------------------------------------------------------------------ CPU baseline
def _match_detections(res, gpu_out):
    """device vs host (fp32 CPU forward): a CPU detection counts as found when the device has a box of the same frame within
    0.25 m (BEV centre) and 0.05 in score"""
    gb, gs, gv = gpu_out["boxes"].float().cpu().numpy(), gpu_out["scores"].float().cpu().numpy(), gpu_out["valid"].cpu().numpy()
    found = total = 0
    gpu_counts, cpu_counts = [], [r["num_detections"] for r in res]
    for f, r in enumerate(res):
        m = gv[f]
        gpu_counts.append(int(m.sum()))
        for bx, sc in zip(r["boxes"], r["scores"]):
            total += 1
            if m.any():
                d = np.hypot(gb[f][m][:, 0] - bx[0], gb[f][m][:, 1] - bx[1])
                found += bool(((d < 0.25) & (np.abs(gs[f][m] - sc) < 0.05)).any())
    check = {"frames": len(res), "detections_cpu": cpu_counts, "detections_gpu": gpu_counts,
             "cpu_detections_found_on_gpu": found, "of": total,
             "rule": f"found / of >= {MATCH_MIN_FOUND} and |count_gpu - count_cpu| <= {MATCH_COUNT_SLACK} in every frame "
                     f"(found = same frame, BEV centre within 0.25 m, score within 0.05)"}
    counts_ok = all(abs(a - b) <= MATCH_COUNT_SLACK for a, b in zip(gpu_counts, cpu_counts))
    return check, bool(total > 0 and found >= MATCH_MIN_FOUND * total and counts_ok)


def cpu_baseline(cpu_state, clouds, gpu_out=None, budget_s=20.0, gpu_out_fp32=None):
    """The same forward on the host through the oracle ("port"): single-threaded oracle for voxelise /
    rulebook / indice_conv / NMS (like the reference's worker-side C++), torch CPU (all cores) for the RPN
    (oracle/cpu_forward.py).  ``gpu_out``: the device results of the same frames -- every CPU detection is looked up in them."""
    from oracle.cpu_forward import forward_frame
    from second_amd.models import SecondDetector, CAR_FHD
    cores = min(os.cpu_count() or 1, 32)   # more threads than that only oversubscribe the 200x176 convs
    torch.set_num_threads(cores)
    det = SecondDetector(CAR_FHD)
    det.load_state_dict(cpu_state)
    det.eval()
    t0 = time.perf_counter()
    res = []
    while len(res) < len(clouds) and (len(res) < 1 or time.perf_counter() - t0 < budget_s):
        res.append(forward_frame(det, clouds[len(res)]))
    n = len(res)
    timed = n
    while time.perf_counter() - t0 < min(10.0, budget_s):      # a sample of >= 10 s: the same frames again, timing only
        forward_frame(det, clouds[timed % len(clouds)])
        timed += 1
    dt = time.perf_counter() - t0
    out = {"value": round(timed / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"{timed} forward(s) over {n} synthetic KITTI frame(s) (17k pts, 16k voxels), fp32; oracle (1 thread) for voxelise/"
                     f"rulebook/indice_conv/NMS + torch CPU ({cores} threads) for the RPN; {dt:.1f} s",
           "detections": [r["num_detections"] for r in res]}
    if gpu_out is not None:
        out["check"], out["detections_match_cpu"] = _match_detections(res, gpu_out)
    if gpu_out_fp32 is not None:    # the SAME network on the device in fp32 storage: {"fp32_exact": ..., "bf16x3": ...} (or one output)
        if isinstance(gpu_out_fp32, dict) and "boxes" not in gpu_out_fp32:
            for label, o in gpu_out_fp32.items():
                chk, ok = _match_detections(res, o)
                out["check_" + label + "_device"] = dict(chk, match=ok)
            gpu_out_fp32 = None
    if gpu_out_fp32 is not None:
        chk, ok = _match_detections(res, gpu_out_fp32)
        chk.pop("rule")
        out["check_fp32_device"] = dict(chk, match=ok)
    return out


# ------------------------------------------------------------------------------------------ launcher
def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` started plainly: become the launcher of N ranks (one process per GPU) of this very command
    line under torch.distributed.run on 127.0.0.1 and relay their output; rank 0 prints the JSON line."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    return subprocess.call(cmd, env=env)


def pin_cpu_affinity(local_rank, local_world):
    """One process per GPU: give rank r the r-th contiguous share of the cores this process may run on, so that the ranks' launch
    threads (a hipGraph replay per step each) do not migrate across each other.  Returns the core list (None: not pinned)."""
    try:
        cores = sorted(os.sched_getaffinity(0))
        if local_world <= 1 or len(cores) < 2 * local_world:
            return None
        per = len(cores) // local_world
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        return mine
    except (AttributeError, OSError):
        return None


def dry_run(args, rank, local_rank, world, affinity=None):
    """Launcher plumbing without a GPU (tests/test_bench_launcher.py): every rank reports who it is over a gloo group, the
    timing reduction (MAX over ranks) is exercised, rank 0 prints one JSON line."""
    import torch.distributed as dist
    info = {"rank": rank, "local_rank": local_rank, "world": world, "pid": os.getpid(), "cpu_affinity": affinity}
    ranks, tmax = [info], float(rank + 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ranks = [None] * world
        dist.all_gather_object(ranks, info)
        t = torch.tensor([tmax], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax = float(t.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "requested_gpus": args.gpus, "ranks": ranks, "max_over_ranks": tmax,
                          "workload": args.workload}), flush=True)


def other_configs(budget_s=270.0):
    """BASELINE.json configs 3 / 4 / 5 at their stated sizes, measured by THIS command (each in a child process: same file, --workload,
    a short run) so that the driver's record carries them; never part of `value`.  Best effort: a failure is reported, not raised."""
    import subprocess
    runs = [("car.fhd.train", "car.fhd.train", ["--dtype", "bf16"], "config 3 (per-GPU step, whole step one hipGraph; DDP adds one 7.3 MB gradient all-reduce)"),
            ("nusc.pp", "nusc.pp", [], "config 4"), ("nusc.fhd", "nusc.fhd", [], "config 5 network, inference, fp16"),
            ("nusc.fhd.train", "nusc.fhd.train", [], "config 5 (per-GPU step, fp16 features + dynamic loss scaling on the device, whole step one hipGraph)"),
            ("nusc.pp.train", "nusc.pp.train", [], "config 4's network trained on the device step (PFN batch statistics + argmax backward on "
                                                   "sec_pfn_train_fwd / _bwd)"),
            ("car.fhd.bf16x3", "car.fhd", ["--dtype", "fp32"], "config 2's network with fp32 storage: sparse convs, RPN and heads on the bf16 MFMA pipe with "
                                                                "split operands (x = hi + lo, three products, fp32 accumulation; 16 significant bits per operand)"),
            ("car.fhd.fp32_exact", "car.fhd", ["--dtype", "fp32_exact", "--inflight", "2"],
             "config 2's network in true fp32, the reference's default precision (train.py:232-235): sparse convs on "
             "v_mfma_f32_32x32x2_f32 / VALU, RPN on torch's fp32 convolutions (MIOpen)")]
    out, t0 = {}, time.time()
    for key, wl, extra, what in runs:
        if time.time() - t0 > budget_s:
            out[key] = {"skipped": "time budget"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", "30", "--warmup", "5", "--no-kernel-table",
               "--no-cpu-baseline", "--no-extra-lines", "--no-other-configs", *extra]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1])
            cfg = d.get("config", {})
            out[key] = {"what": what, "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"],
                       "per_step_per_gpu": cfg.get("frames_per_step_per_gpu") or cfg.get("samples_per_step_per_gpu"),
                       "points_per_frame": cfg.get("points_per_frame"), "rows_per_frame": cfg.get("rows_per_frame"),
                       "steps_in_flight": cfg.get("steps_in_flight"), "loss_last_step": d.get("loss_last_step"), "steps": d["steps"]}
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": repr(e)[:300]}
    return out


# ------------------------------------------------------------------------------------------ extra lines
def time_e2e(det, points, offsets, inflight, steps, warmup, serialize_rpn=False, rpn_tokens=None):
    """SURVEY 8(d) "end-to-end": the same loop as the timed region, but every step's clouds start in PINNED HOST memory
    (copied into the step's own input buffers on its stream) and its detections end in pinned host memory.  Per-lane input
    buffers, so lane k's copy overlaps the other lanes' compute."""
    from second_amd.models import InFlightRunner
    runner = InFlightRunner(det, points, offsets, inflight=inflight, private_inputs=True, serialize_rpn=serialize_rpn and not det.pillars, rpn_tokens=rpn_tokens)
    hp, ho = points.cpu().pin_memory(), offsets.cpu().pin_memory()
    for _ in range(max(3, warmup)):
        runner.step(hp, ho, fetch=True)
    dt, timing, _ = measure(lambda: runner.step(hp, ho, fetch=True), torch.cuda.synchronize, steps, 0)
    runner.synchronize()
    lat = []
    for _ in range(20):
        t1 = time.perf_counter()
        runner.step(hp, ho, fetch=True)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    frames = offsets.numel() - 1
    return {"frames_per_s": round(frames * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "steps_in_flight": inflight,
            "single_step_latency_ms": round(float(np.median(lat)) * 1e3, 4),
            "h2d_bytes_per_step": int(hp.numel() * 4 + ho.numel() * 4),
            "d2h_bytes_per_step": int(sum(v.numel() * v.element_size() for v in runner.host_outputs[0].values())),
            "timing": {k: timing[k] for k in ("windows", "steps_per_window", "spread_pct")},
            "what": "pinned host clouds -> HBM -> detections -> pinned host, copies inside the timed loop"}


def time_batch1(det, points, offsets, steps=100):
    """One frame per step (the reference's only published figure is batch-1 latency, README.md:27: 0.04 s per KITTI frame
    on a 1080 Ti including pre-processing).  Latency = host wall time of one graph replay of the whole path on frame 0,
    synchronised; `e2e` adds the pinned-host copies either side; `frames_per_s_inflight3` = three single-frame steps in flight."""
    from second_amd.models import InFlightRunner
    n0 = int(offsets[1].item())
    p1, o1 = points[:n0].clone(), offsets[:2].clone()
    det.calibrate(p1, o1)
    runner = InFlightRunner(det, p1, o1, inflight=3, private_inputs=True)
    hp, ho = p1.cpu().pin_memory(), o1.cpu().pin_memory()

    def lat(fn, n):
        for _ in range(5):
            fn()
            torch.cuda.synchronize()
        v = []
        for _ in range(n):
            t1 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            v.append(time.perf_counter() - t1)
        return round(float(np.median(v)) * 1e3, 4)
    res = {"latency_ms": lat(lambda: runner.replays[0](), steps), "e2e_latency_ms": lat(lambda: runner.step(hp, ho, fetch=True), steps)}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps * 3):
        runner.step()
    torch.cuda.synchronize()
    res["frames_per_s_inflight3"] = round(steps * 3 / (time.perf_counter() - t0), 1)
    runner.synchronize()
    res["points"], res["what"] = n0, "frame 0 alone, hipGraph replay + host sync per step (median)"
    return res


def time_dropin_modules(cpu_state, points, offsets, iters=15):
    """The drop-in path's own speed (never `value`): ``SecondDetector.forward(example)`` -- the reference's VoxelNet.forward contract
    (voxelnet.py:339-375: example dict of voxels / num_points / coordinates / anchors in, list of per-frame dicts out) through the
    MODULE graph the unmodified reference builds over this package's `spconv`: SimpleVoxel module, spconv.SparseSequential of 14 x
    (conv, BatchNorm1d, ReLU) with the drop-in default first-touch rulebooks (hash builds, one host sync per strided layer, like
    spconv's numActOut), `.dense()`, the torch RPN (MIOpen convolutions), per-frame result dicts -- in fp32, the reference's default
    precision, on the same 8 frames.  `peephole`: the inference fusion conv + BN + ReLU inside SparseSequential (what the reference
    gets in eval mode); `unfused`: three separate modules per layer."""
    from second_amd.models import SecondDetector, CAR_FHD
    from second_amd import ops
    det = SecondDetector(CAR_FHD)
    det.load_state_dict(cpu_state)
    det = det.eval().cuda()
    batch = offsets.numel() - 1
    prev = ops.set_rulebook_numbering("first_touch")
    try:
        with torch.no_grad():
            vox = det.voxel_generator.generate_device(points, offsets)
            example = {"voxels": vox["voxels"], "num_points": vox["num_points_per_voxel"], "coordinates": vox["coordinates"],
                       "anchors": det.anchors.unsqueeze(0).expand(batch, -1, -1).contiguous()}
            out = {}
            for tag, fuse in (("peephole", True), ("unfused", False)):
                det.middle_feature_extractor.middle_conv.fuse_inference = fuse
                for _ in range(3):
                    res = det(example)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    res = det(example)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / iters
                out[tag] = {"frames_per_s": round(batch / dt, 1), "ms_per_batch": round(dt * 1e3, 3)}
            out["detections_last_batch"] = int(sum(r["box3d_lidar"].shape[0] for r in res))
    finally:
        ops.set_rulebook_numbering(prev)
    out["what"] = ("SecondDetector.forward(example), fp32, eager dynamic shapes, first-touch rulebooks, .dense(), torch RPN; voxelisation "
                   "not included (the example dict is the reference's input to VoxelNet.forward)")
    return out


def time_dropin_fused(cpu_state, points, offsets, iters=60):
    """The accelerated drop-in path (never `value`): a network OBJECT shaped like the reference's ``build_network`` result
    (tests/reference_standin.py -- the GPU box has no reference checkout; tests/test_dropin_reference.py ties it to the real one)
    handed to ``second_amd.compat.accelerate_model`` and called as the reference's evaluate() calls it: ``net(example)``
    (voxelnet.py:339-375, train.py:524) with the example dict of voxels / num_points / coordinates / anchors in, the list of
    per-frame dicts out.  Every call = copies into the static buffers, ONE hipGraph replay, one device -> host copy, one host
    synchronisation (the reference's return value has data-dependent shapes), result views: wall time per call, synchronous, one
    call at a time.  An fp32 network (the reference's default) is served either with split-operand products (`fp32_net_bf16x3`: the
    default, 16 significant bits per operand) or with IEEE fp32 products (`fp32_net_exact`: ``accelerate_model(net, fp32_exact=True)``);
    fp16 = after ``net.half()`` (train.py:468-472) with float16 examples; bf16 = ``accelerate_model(net, dtype=torch.bfloat16)``;
    `bf16_forced_deferred` = ``accelerate_model(net, dtype=torch.bfloat16, deferred=True)`` driven like evaluate() (results collected in a
    list, read after the loop; the time includes reading every one of them): calls return at once and alternate between two lanes.  Voxelisation is not included (it is the data loader's job in
    the reference: the example dict is VoxelNet.forward's input)."""
    tests_dir = os.path.join(ROOT, "tests")
    if tests_dir not in sys.path:
        sys.path.insert(0, tests_dir)
    from reference_standin import build_voxelnet
    from second_amd import compat
    from second_amd.models import CAR_FHD
    batch = offsets.numel() - 1
    out = {}
    base = None
    for tag, half, forced, exact, deferred in (("fp32_net_bf16x3", False, None, False, False), ("fp32_net_exact", False, None, True, False),
                                               ("fp16_net_half", True, None, False, False), ("bf16_forced", False, torch.bfloat16, False, False),
                                               ("bf16_forced_deferred", False, torch.bfloat16, False, True)):
        if os.environ.get("SEC_BENCH_DROPIN_VARIANTS") and tag not in os.environ["SEC_BENCH_DROPIN_VARIANTS"].split(","):
            continue                                       # (one variant alone: diagnostics, and the child process below)
        if deferred and not os.environ.get("SEC_BENCH_DROPIN_VARIANTS"):
            # the asynchronous engine in a FRESH process -- what evaluate() is: one engine per process.  (Until its lanes got
            # high-priority streams -- models.lane_stream: a hardware queue each, whatever streams the process created before -- the same
            # loop measured 11-13 k instead of 17 k frames/s in a process that had run another engine first.)
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dropin-leg", tag], capture_output=True, text=True, timeout=240,
                                   env=dict(os.environ, SEC_BENCH_DROPIN_VARIANTS=tag))
                out[tag] = dict(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])[tag], process="fresh child process")
            except Exception as e:  # noqa: BLE001
                out[tag] = {"error": repr(e)[:300]}
            continue
        net = build_voxelnet(CAR_FHD)
        net.load_state_dict(cpu_state)
        net = net.eval().cuda()
        with torch.no_grad():
            vox = net.voxel_generator.generate_device(points, offsets)
        fdt = torch.float16 if half else torch.float32
        example = {"voxels": vox["voxels"].to(fdt), "num_points": vox["num_points_per_voxel"], "coordinates": vox["coordinates"],
                   "anchors": net.anchors.unsqueeze(0).expand(batch, -1, -1).contiguous().to(fdt)}
        if half:
            net.half()
            for m in net.modules():
                if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                    m.float()
        compat.accelerate_model(net, dtype=forced, fp32_exact=exact, deferred=deferred)
        n_it = iters if not exact else max(10, iters // 4)
        with torch.no_grad():
            # warm-up by TIME, like the main harness: the network was just built on the host (GPU idle, clocks down); deferred mode
            # also needs every lane captured, its pinned slots and allocator pools in place
            tw, nw = time.perf_counter(), 0
            while nw < (30 if deferred else 5) or time.perf_counter() - tw < (1.0 if not exact else 0.3):
                res = net(example)
                len(res[0])                               # (deferred results resolve when read)
                nw += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if deferred:        # the reference's evaluate() loop: `detections += net(example)`, first read after the loop (train.py:519-539)
                n_it *= 4
                passes = []
                for rep in range(3):                      # first pass: pinned landing slots / allocator pools of the deep pipeline are made
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    collected = []
                    for _ in range(n_it):
                        collected += net(example)
                    t_issue = (time.perf_counter() - t0) / n_it
                    for d in collected:
                        d["scores"]
                    torch.cuda.synchronize()
                    passes.append(round(batch * n_it / (time.perf_counter() - t0), 1))
                    res = collected[-batch:]
                    del collected
                torch.cuda.synchronize()
                t0 = time.perf_counter() - batch * n_it / passes[-1]      # `frames_per_s` below = the last (steady) pass
            else:
                for _ in range(n_it):
                    res = net(example)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_it
        eng = net._second_amd_engine
        dets = int(sum(r["box3d_lidar"].shape[0] for r in res))
        base = dets if base is None else base
        out[tag] = {"frames_per_s": round(batch / dt, 1), "ms_per_call": round(dt * 1e3, 3), "arithmetic": eng._det.arithmetic(),
                    "detections_last_call": dets, "graph_captures": eng.stats["captures"], "calls_served_by_the_original_forward": eng.stats["original_calls"]}
        if deferred:
            out[tag].update(frames_per_s_by_pass=passes, calls_per_pass=n_it, issue_ms_per_call_last_pass=round(t_issue * 1e3, 3),
                            lanes=eng.lanes, calls_redone_synchronously=eng.stats["deferred_redone"])
        del net, eng
        import gc
        gc.collect()               # net <-> its shadowed forward <-> the engine form a cycle: free the engine's buffers and graphs NOW
        torch.cuda.empty_cache()
    out["what"] = ("compat.accelerate_model(net); net(example) -- VoxelNet.forward's contract, synchronous, one call at a time, batch "
                   f"{batch}; rows per call {int(vox['voxels'].shape[0])}; voxelisation not included")
    return out


def time_dropin_train(iters_eager=6, iters_fused=40, batch=4):
    """The reference's TRAINING loop on the drop-in surface (never `value`; BASELINE config 3's per-GPU batch of 4): a network OBJECT
    shaped like ``build_network``'s result (tests/reference_standin.py; default init, train mode) driven exactly as
    second/pytorch/train.py:306-325 drives it --

        ret = net(example); loss = ret["loss"].mean(); loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0); optimizer.step(); optimizer.zero_grad()

    -- with torch.optim.AdamW on the network's own parameters (adam + fixed weight decay 0.01, car.fhd.config:180-188), samples/s.
    `eager`: the un-accelerated module graph, fp32 (three modules per sparse layer, first-touch rulebooks with a host round trip per
    strided layer, .dense(), torch / MIOpen RPN, ~60 torch launches of loss glue, autograd through all of it).  `fused`:
    ``compat.accelerate_model(net, train_dtype=torch.bfloat16)`` -- the same loop, the same optimizer object type and parameters, forward
    and backward each ONE hipGraph replay (second_amd/dropin_train.py; bf16 features over the fp32 parameters).  The example dict
    (voxels, coordinates, labels, reg_targets ...) is resident in HBM: voxelisation and target assignment are the data loader's job
    in the reference."""
    tests_dir = os.path.join(ROOT, "tests")
    if tests_dir not in sys.path:
        sys.path.insert(0, tests_dir)
    from reference_standin import build_voxelnet, train_example_of
    from second_amd import compat, synthetic as syn
    from second_amd.models import CAR_FHD
    clouds = [syn.syn_kitti_cloud(s) for s in range(batch)]
    boxes = [syn.syn_kitti_boxes(s, 12) for s in range(batch)]
    out = {}
    for tag, iters in (("eager", iters_eager), ("fused", iters_fused)):
        torch.manual_seed(0)
        net = build_voxelnet(CAR_FHD).cuda().train()
        ex = train_example_of(net, clouds, boxes, torch.device("cuda"))
        if tag == "fused":
            compat.accelerate_model(net, train_dtype=torch.bfloat16)
        opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=0.01, betas=(0.9, 0.99))

        def step():
            ret = net(ex)
            loss = ret["loss"].mean()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)
            opt.step()
            opt.zero_grad()
            return loss
        for _ in range(3):
            first = step()
        first = float(first.detach())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            last = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        out[tag] = {"samples_per_s": round(batch / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": iters,
                    "loss_after_3_steps": round(first, 4), "loss_last_step": round(float(last.detach()), 4)}
        eng = getattr(net, "_second_amd_engine", None)
        if eng is not None:
            out[tag].update(graph_captures=eng.stats.get("train_captures"), calls_served_by_the_original_forward=eng.stats["original_calls"],
                            fallback_reason=eng.stats.get("train_fallback_reason"))
        del net, opt, ex
        torch.cuda.empty_cache()
    out["what"] = (f"train.py:306-325 loop, batch {batch}, 17 000 points -> 16 000 voxels per frame, torch.optim.AdamW + clip_grad_norm_(10) on the "
                   "network's own parameters; eager = the fp32 module graph, fused = accelerate_model(train_dtype=bfloat16): two hipGraph replays per step")
    return out


def time_scene_density(args, main_value, budget_s=200.0):
    """How much of the headline depends on the BEV sparsity of the scene: the default path on a DENSE seeded scene
    (synthetic.syn_kitti_cloud(scene="dense"): 14-20 % of the 200 x 176 BEV cells occupied instead of 4-7 %, same 16 000 voxels /
    17 000 points per frame -- and, because its voxels are scattered instead of clustered, 2-3x the active rows in the strided
    levels of the sparse middle) and both scenes with the background-tile skip switched off (every RPN tile convolved).  Each is
    THIS command in a fresh child process (same harness, lanes and flags; `--scene` / `--background-skip`), so that no figure
    depends on what the parent process ran before; never `value`."""
    import subprocess
    res = {"sparse_scene": {"frames_per_s": main_value, "what": "the timed region of this line (SURVEY 8d clouds)"}}
    t0 = time.time()
    for key, extra in (("dense_scene", ["--scene", "dense"]), ("dense_scene_skip_off", ["--scene", "dense", "--background-skip", "0"]),
                       ("sparse_scene_skip_off", ["--background-skip", "0"]), ("sparse_scene_scan_order", ["--point-order", "scan"])):
        if time.time() - t0 > budget_s:
            res[key] = {"skipped": "time budget"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--inflight", str(args.inflight),
               "--serialize-rpn", str(args.serialize_rpn), "--rpn-tokens", str(args.rpn_tokens), "--dtype", args.dtype, "--no-kernel-table", "--no-cpu-baseline",
               "--no-extra-lines", "--no-other-configs", *extra]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            bt = d["config"].get("rpn_background_tiles") or {}
            res[key] = {"frames_per_s": d["value"], "ms_per_step": d["ms_per_step"], "spread_pct": d["timing"]["spread_pct"],
                        "single_step_latency_ms": d["config"].get("single_step_latency_ms"), "rows_per_frame": d["config"].get("rows_per_frame"),
                        "bev_cells_occupied": bt.get("bev_cells_occupied"), "live_tiles_per_conv": bt.get("live_tiles_per_conv_last_step")}
        except Exception as e:  # noqa: BLE001
            res[key] = {"error": repr(e)[:300]}
    res["skip_off"] = {"sparse_scene": res["sparse_scene_skip_off"].get("frames_per_s"), "dense_scene": res["dense_scene_skip_off"].get("frames_per_s")}
    res["what"] = ("frames/s of the default path on the bench's clouds (sparse_scene = `value`) and on a dense seeded scene, of "
                   "--background-skip 0 on both, and of the bench's clouds in lidar firing order instead of shuffled (--point-order scan: real "
                   ".bin files are un-shuffled; the shuffle of SURVEY 8d is the worst case for every gather); fresh child processes of this command")
    return res


# ------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # >= 200 timed iterations after 20 warm-ups
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32", "fp32_exact"],
                    help="fp32 = fp32 storage with split-operand bf16 MFMA products (reported as dtype \"bf16x3\"); fp32_exact = IEEE fp32 "
                         "products (sparse convs on the fp32 MFMA, RPN on torch's fp32 convolutions): the reference's arithmetic")
    ap.add_argument("--mode", default="graph", choices=["graph", "static", "eager"],
                    help="graph: static-capacity forward captured in a hipGraph (default); static: same, eager "
                         "launches; eager: the dynamic-shape drop-in path (host syncs per strided layer)")
    ap.add_argument("--point-order", default="shuffle", choices=["shuffle", "sorted", "scan"],
                    help="shuffle = SURVEY 8d (worst case for every gather); sorted = cell (z, y, x) order; scan = firing order of a spinning "
                         "lidar (azimuth step, then beam), the order of a KITTI velodyne .bin file")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-table", action="store_true", help="skip the per-launch roofline table (`kernels` key)")
    ap.add_argument("--no-extra-lines", action="store_true", help="skip config.e2e_from_pinned_host and config.batch1")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short child runs of BASELINE configs 3 / 4 / 5 (`other_configs`)")
    ap.add_argument("--stages", action="store_true", help="also print per-stage timings to stderr")
    ap.add_argument("--inflight", type=int, default=4,
                    help="graph mode: number of steps (graph replays, each a full pass over the batch with its own activation "
                         "buffers) kept in flight on separate HIP streams; the latency-bound sparse stages of one step then "
                         "overlap the MFMA-bound RPN of another.  1 = strictly one step at a time")
    ap.add_argument("--serialize-rpn", type=int, default=1,
                    help="1 (default, car.fhd-type networks): every step is three graphs (sparse front / RPN / predict) and the lanes' RPN "
                         "segments pass a token: one MFMA-bound segment at a time; 0: one graph per step, lanes uncoordinated")
    ap.add_argument("--branches", type=int, default=1,
                    help="graph mode: capture the batch as this many independent frame-group chains on separate streams of "
                         "ONE hipGraph (the single-step-latency variant of --inflight); 1 = a single chain")
    ap.add_argument("--workload", default="car.fhd", choices=sorted(WORKLOADS),
                    help="car.fhd (default, the BASELINE metric) or another BASELINE config for a side measurement")
    ap.add_argument("--batch", type=int, default=0, help="frames (samples) per step per GPU; 0 = the workload's BASELINE batch")
    ap.add_argument("--default-heads", action="store_true",
                    help="car.fhd: keep the default-initialised heads (tie-dominated top-k; the round-1/2 bench network)")
    ap.add_argument("--profile-run", action="store_true",
                    help="for runs under rocprofv3: no self-warming beyond --warmup and ONE timed window (keeps the trace small); the "
                         "printed value is then not a benchmark figure")
    ap.add_argument("--rpn-tokens", type=int, default=0,
                    help="--serialize-rpn 1: RPN segments allowed to run at a time; 0 = InFlightRunner's own rule (two when at most 60 %% of the last "
                         "conv's tiles are live on the calibration scene, one on denser scenes)")
    ap.add_argument("--background-skip", type=int, default=1,
                    help="RPN convs after the first compute only the tiles a site (or the zero padding) can reach and fill the others "
                         "with the layer's background vector (bit-identical outputs; 0 = convolve every tile)")
    ap.add_argument("--lazy-background", type=int, default=1,
                    help="with --background-skip 1: the RPN convs write their live tiles only and read background tiles of their input from the "
                         "empty frame's maps (sec_conv2d_nhwc_tiles_lazy; bit-identical); 0 = every layer copies its background tiles")
    ap.add_argument("--scene", default="open", choices=["open", "dense"],
                    help="car.fhd: open = the SURVEY 8d clouds (default, the BASELINE workload); dense = synthetic.syn_kitti_cloud(scene='dense'), "
                         "14-20 %% of the BEV cells occupied at the same points / voxels per frame (a robustness side measurement)")
    ap.add_argument("--dry-run", action="store_true", help="launcher plumbing only: ranks report themselves (gloo), no GPU work")
    ap.add_argument("--dropin-leg", default="", help="(internal) run ONE variant of config.dropin_fused in this process and print it")
    args = ap.parse_args()
    global WL, SELF_WARM_MIN_S, SELF_WARM_MAX_S, TIMED_MIN_S
    if args.profile_run:
        SELF_WARM_MIN_S = SELF_WARM_MAX_S = TIMED_MIN_S = 0.0
    WL = dict(WORKLOADS[args.workload])
    if args.batch > 0:
        WL["desc"] = WL["desc"].replace(f"batch={WL['batch']} ", f"batch={args.batch} ")
        WL["batch"] = args.batch
    if args.workload != "car.fhd":
        args.no_cpu_baseline = True
        if "dtype" in WL and "--dtype" not in " ".join(sys.argv):
            args.dtype = WL["dtype"]

    if args.dropin_leg:            # child of time_dropin_fused: one drop-in variant alone in this process
        from second_amd import synthetic as syn
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        _, pts_, offs_ = build_inputs(0, dev)
        _, state_ = build_detector(dev, torch.bfloat16, syn.syn_kitti_cloud(0))
        os.environ["SEC_BENCH_DROPIN_VARIANTS"] = args.dropin_leg
        print(json.dumps(time_dropin_fused(state_, pts_, offs_)), flush=True)
        return
    # one process per GPU: under torch.distributed.run this is one rank; started plainly with --gpus N > 1 it launches them
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    affinity = pin_cpu_affinity(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world))) if world > 1 else None
    if args.dry_run:
        return dry_run(args, rank, local_rank, world, affinity)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (use gpurun)"
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s); running {world}", file=sys.stderr)
    assert local_rank < torch.cuda.device_count(), f"LOCAL_RANK {local_rank} but {torch.cuda.device_count()} visible GPU(s)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "fp32_exact": torch.float32}[args.dtype]
    exact = args.dtype == "fp32_exact"

    if args.workload.endswith(".train"):
        if args.workload == "car.fhd.train" and args.dtype == "bf16" and "--dtype" not in " ".join(sys.argv):
            args.dtype = "fp32"                     # training default: the reference's precision (config 5 names fp16)
        return train_bench(args, rank, local_rank, world, device)
    from second_amd import ops
    clouds, points, offsets = build_inputs(rank, device, args.point_order, args.scene)
    # the heads are calibrated on seed-0's cloud on EVERY rank (same network everywhere), not on the rank's own first frame
    from second_amd import synthetic as syn
    det, cpu_state = build_detector(device, dtype, None if args.default_heads else syn.syn_kitti_cloud(0), exact=exact)
    if hasattr(det.rpn, "skip_background"):
        det.rpn.skip_background = bool(args.background_skip)
        det.rpn.lazy_background = bool(args.lazy_background)

    # first subm2 layer (64->64 SubM on the 11x400x352 grid): the largest 64->64 3x3x3 launch of the forward
    timer = ConvCapture(lambda m: m["cin"] == 64 and m["cout"] == 64 and m["kvol"] == 27 and m["n_in"] == m["n_out"]
                        and m["n_out"] > 20000 * WL["batch"] // 8)
    ops.set_conv_profiler(timer)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        if args.mode != "eager":
            det.calibrate(points, offsets)
        if args.mode == "graph":
            from second_amd.models import InFlightRunner
            serialize = (bool(args.serialize_rpn) and not det.pillars and getattr(det, "_infer_dtype", None) is not None
                         and args.branches <= 1 and args.inflight > 1)
            runner = InFlightRunner(det, points, offsets, inflight=args.inflight, branches=args.branches, serialize_rpn=serialize,
                                    rpn_tokens=args.rpn_tokens or None)
            graph_parts = runner.parts      # branches > 1: the roofline probe below times one branch's launch
            replays = runner.replays
            outs = runner.outputs[-1]
            out = {"valid": torch.cat([o["valid"] for o in outs])} if args.branches > 1 else outs
            step = runner.step
        elif args.mode == "static":
            step = lambda: det.forward_points(points, offsets, static=True)
        else:
            step = lambda: det.forward_points(points, offsets)
        elapsed, timing, r = measure(step, barrier, args.steps, args.warmup, world, device)
        latency_ms = None
        if args.mode == "graph":   # one step alone, start to finish, as ONE graph (what --inflight 1 would run back to back)
            one = det.make_graphed(points, offsets)[0] if isinstance(replays[0], tuple) else replays[0]
            for _ in range(3):
                one()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                one()
                torch.cuda.synchronize()
            latency_ms = round((time.perf_counter() - t1) / 20 * 1e3, 4)
        if args.mode != "graph":
            out = r
        if args.mode == "graph":
            runner.synchronize()         # capacity-overflow counters of every lane
        elif args.mode != "eager":
            det.check_overflow()
        # per-kernel timing of the SubMConv3d kernel: capture the launch arguments during one eager forward
        # right after the timed region, then re-issue that launch 100x back-to-back between two HIP events on
        # the launch stream (events cannot be timed inside a captured graph; a single event pair around one
        # ~20 us launch would add ~10 us of launch gap).  profiles/ holds the rocprofv3 cross-check.
        # capture exactly the launch the timed region runs: with a branched graph that is the static forward of one branch
        timer.enabled = True
        if args.mode == "graph" and args.branches > 1:
            det.forward_points(*graph_parts[0], static=True)
            frames_in_launch = graph_parts[0][1].numel() - 1
        else:
            det.forward_points(points, offsets, static=args.mode != "eager")
            frames_in_launch = WL["batch"]
        torch.cuda.synchronize()
        timer.enabled = False
        t_kernel, kernel_sig = time_kernel(timer.call) if timer.call is not None else (None, "")
        # auxiliary: the same layer as ONE full-batch launch (what a single-chain graph, --branches 1, runs)
        aux = None
        if args.mode == "graph" and args.branches > 1 and timer.call is not None:
            det.calibrate(points, offsets)
            timer8 = ConvCapture(timer.select)
            ops.set_conv_profiler(timer8)
            timer8.enabled = True
            det.forward_points(points, offsets, static=True)
            torch.cuda.synchronize()
            timer8.enabled = False
            if timer8.call is not None:
                m8 = timer8.call
                rows8 = int(m8["num_out_dev"][0].item()) if m8.get("num_out_dev") is not None else m8["n_out"]
                pairs8 = int((m8["nbr_out"][:rows8] >= 0).sum().item())
                s8 = 2 if m8["dtype"] != torch.float32 else 4
                b8 = s8 * (pairs8 * 64 + rows8 * 64) + 8 * pairs8 + s8 * 27 * 64 * 64
                t8, _ = time_kernel(m8)
                aux = {"frames": WL["batch"], "rows": rows8, "pairs": pairs8, "launch_us": round(t8 * 1e6, 2),
                       "frac": round(b8 / t8 / 1e9 / HBM_PEAK_GBS, 4)}
        ops.set_conv_profiler(None)
        roof_mfma = time_rpn_conv(det, WL["batch"]) if args.dtype == "bf16" else None
        dtype_note = {"bf16x3": "fp32 storage; every product = three bf16 MFMA passes on (hi, lo) operand pairs: 16 significant bits per operand "
                                "and per stored activation plane pair, fp32 accumulation -- narrower than fp32 (see --dtype fp32_exact)",
                      "fp32": "IEEE fp32 products and accumulation (v_mfma_f32_32x32x2_f32 / VALU sparse convs, torch fp32 RPN)"}.get(det.arithmetic())
        ktable = None
        if rank == 0 and args.mode != "eager" and not args.no_kernel_table:
            try:
                ktable = kernel_table(det, points, offsets)
            except Exception as e:  # noqa: BLE001 -- the table is diagnostics: never lose the line over it
                ktable = [{"error": repr(e)}]
        e2e = batch1 = dropin = fused = scenes = dtrain = None
        if rank == 0 and args.workload == "car.fhd" and not args.no_extra_lines and not args.default_heads:
            try:
                dropin = time_dropin_modules(cpu_state, points, offsets)
            except Exception as e:  # noqa: BLE001 -- a side measurement: never lose the line over it
                dropin = {"error": repr(e)[:300]}
            try:
                fused = time_dropin_fused(cpu_state, points, offsets)
            except Exception as e:  # noqa: BLE001
                fused = {"error": repr(e)[:300]}
            try:
                with torch.enable_grad():          # (this block of main() runs under no_grad)
                    dtrain = time_dropin_train()
            except Exception as e:  # noqa: BLE001
                dtrain = {"error": repr(e)[:300]}
        if rank == 0 and args.mode == "graph" and args.branches == 1 and args.workload == "car.fhd" and not args.no_extra_lines:
            e2e = time_e2e(det, points, offsets, max(1, args.inflight), min(args.steps, 200), args.warmup, serialize_rpn=serialize, rpn_tokens=args.rpn_tokens or None)
            batch1 = time_batch1(det, points, offsets)      # last: it re-calibrates the static capacities for one frame
        if (rank == 0 and world == 1 and args.mode == "graph" and args.branches == 1 and args.workload == "car.fhd" and not args.no_extra_lines
                and getattr(det.rpn, "background_convs", 0) and det.rpn.skip_background):
            try:
                scenes = time_scene_density(args, round(WL["batch"] * args.steps * world / elapsed, 1))
            except Exception as e:  # noqa: BLE001
                scenes = {"error": repr(e)[:300]}

    # roofline of the SubMConv3d 64->64 kernel: algorithmic bytes / mean measured launch time
    roof = None
    if timer.call is not None:
        meta = timer.call
        s = 2 if meta["dtype"] != torch.float32 else 4
        plan = ops.indice_conv_plan(meta["cin"], meta["cout"], meta["kvol"], meta["n_out"], meta["dtype"], packed=meta["mfma"])
        kname = kernel_sig or PLAN_NAMES.get(plan, str(plan))
        rows = int(meta["num_out_dev"][0].item()) if meta.get("num_out_dev") is not None else meta["n_out"]
        pairs = int((meta["nbr_out"][:rows] >= 0).sum().item())
        meta = dict(meta, n_out=rows)
        b_alg = s * (pairs * meta["cin"] + meta["n_out"] * meta["cout"]) + 8 * pairs + s * meta["kvol"] * meta["cin"] * meta["cout"]
        t_mean = t_kernel
        ach = b_alg / t_mean / 1e9
        # HBM-side bytes per launch: NOT measured in this run -- read from the committed PMC passes of THIS kernel instantiation
        # (the full template signature is the key) on this (seeded, deterministic) launch; the file says how they were
        # collected; null if no pass matches signature + rows + pairs
        traffic, traffic_source, pmc = None, None, None
        def _recency(fname):         # profiles are tagged rNN_<a..z, aa..>: the newest pass of this kernel wins
            part = fname.split("_")
            return (part[0], len(part[1]) if len(part) > 1 else 0, part[1] if len(part) > 1 else "")
        for fname in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), key=_recency, reverse=True):
            try:
                with open(os.path.join(ROOT, "profiles", fname)) as f:
                    tj = json.load(f)
                for ent in tj["entries"]:
                    if ent["rows"] == meta["n_out"] and ent["pairs"] == pairs and ent.get("kernel_signature") == kname:
                        traffic, traffic_source = ent["traffic_bytes_per_launch"], "profiles/" + fname
                        pmc = {k: ent[k] for k in ("mfma_pipe_busy", "effective_clock_ghz", "launch_us_under_profiler") if k in ent} or None
            except (OSError, KeyError, ValueError):
                pass
            if traffic is not None:
                break
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source, "pmc": pmc,
                "kernel": f"{kname} (SubMConv3d subm2, {frames_in_launch} frames per launch)",
                "launch_us": round(t_mean * 1e6, 2), "launches_timed": 100, "alg_bytes_per_launch": b_alg,
                "rows": meta["n_out"], "pairs": pairs, "frac_of_6.29TBs_measured_peak": round(ach / 6290.0, 4),
                "same_layer_as_one_full_batch_launch": aux}

    if args.stages and rank == 0:
        stage_times(det, points, offsets)

    bg_tiles = None
    if rank == 0 and getattr(det.rpn, "background_convs", 0):
        if det.rpn.skip_background:
            lt = [k for k in (ktable or []) if k["op"] in ("conv2d_nhwc_tiles", "conv2d_nhwc_gather", "conv2d_nhwc_tiles_tail")]
            bg_tiles = {"enabled": True, "lazy": bool(getattr(det.rpn, "lazy_background", False)), "live_tiles_per_conv": [k.get("live_tiles") for k in lt] or None, "tiles": lt[0].get("tiles") if lt else None,
                        "what": "conv j of the RPN (j = 0..5) convolves only the 8 x 16 tiles a site of the sparse middle can reach within j + 1 "
                                "steps; the other tiles equal the network's output for an EMPTY frame at that position exactly (any weights) "
                                "and are copied from it -- or, with lazy = true (--lazy-background 1, default), never written: the next conv reads the halo "
                                "pixels that fall into such a tile from the empty frame's map, the 1x1 tail runs in the epilogue of the last conv's live "
                                "tiles (bit-identical either way).  Data dependent: --background-skip 0 convolves every tile"}
        else:
            bg_tiles = {"enabled": False}
        try:
            with torch.no_grad():        # what share of the RPN input's BEV cells holds a site, per frame (the quantity the tile skip depends on)
                v_ = det.voxel_generator.generate_device(points, offsets, mean_features=4, mean_dtype=det._infer_dtype)
                sp_ = det.middle_feature_extractor(v_["mean"], v_["coordinates"], WL["batch"], channels_last=True, bev_sparse=True)
                if hasattr(sp_, "site_map"):
                    bg_tiles["bev_cells_occupied"] = [round(float(x), 4) for x in (sp_.site_map() > 0).any(dim=1).float().mean(dim=(1, 2)).cpu()]
            if det.rpn.skip_background and det.rpn.last_live_counts is not None and not bg_tiles.get("live_tiles_per_conv"):
                bg_tiles["live_tiles_per_conv_last_step"] = [int(x) for x in det.rpn.last_live_counts.sum(1).cpu()]
        except Exception as e:  # noqa: BLE001
            bg_tiles["bev_cells_occupied"] = repr(e)[:200]
        if scenes is not None:
            bg_tiles["frames_per_s"] = scenes
    rows_per_frame = None
    if rank == 0:
        with torch.no_grad():   # live voxel / pillar count of this input (what BASELINE quotes the configs on)
            v_ = det.voxel_generator.generate_device(points, offsets)
        rows_per_frame = int(v_["voxel_num"]) // WL["batch"]
    if rank == 0:
        frames = WL["batch"] * args.steps * world
        one_at_a_time = round(WL["batch"] / (latency_ms * 1e-3), 1) if latency_ms else None
        res = {
            "metric": WL["metric"], "value": round(frames / elapsed, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": det.arithmetic(), "data": "synthetic", "timing": timing,
            "config": {"workload": WL["desc"], "dtype_note": dtype_note,
                       "frames_per_step_per_gpu": WL["batch"], "parallelism": f"frame-dp{world}", "launch_mode": args.mode,
                       "graph_branches": args.branches if args.mode == "graph" else None,
                       "steps_in_flight": max(1, args.inflight) if args.mode == "graph" else 1,
                       "rpn_segments_serialized": bool(serialize) if args.mode == "graph" else None,
                       "rpn_tokens": runner.rpn_tokens if (args.mode == "graph" and serialize) else None,
                       "single_step_latency_ms": latency_ms, "frames_per_s_one_step_at_a_time": one_at_a_time,
                       "rulebook_numbering": det.rulebook_numbering, "points_per_frame": int(points.shape[0]) // WL["batch"],
                       "rows_per_frame": rows_per_frame,
                       "weights": "seeded random, default heads (tie-dominated top-k)" if args.default_heads or WL["cfg"] != "CAR_FHD"
                                  else "seeded random with trained-like heads (synthetic.randomise_like_trained / sharpen_heads)",
                       "rpn_background_tiles": bg_tiles,
                       "e2e_from_pinned_host": e2e, "batch1": batch1, "dropin_module_path": dropin, "dropin_fused": fused, "dropin_train": dtrain},
            "roofline": roof,
            "roofline_mfma": roof_mfma,     # second-largest consumer by kind: the dense RPN conv, MFMA bound
            "kernels": ktable,              # per-launch table of one step (SURVEY 8d formulas)
        }
        if scenes is not None:
            # the headline's dependence on the scene, at the TOP level of the record: the same command on the dense seeded scene and
            # on the bench's clouds in lidar firing order (never `value`; details under config.rpn_background_tiles.frames_per_s)
            res["value_dense_scene"] = (scenes.get("dense_scene") or {}).get("frames_per_s")
            res["value_scan_order"] = (scenes.get("sparse_scene_scan_order") or {}).get("frames_per_s")
        if world == 1 and not args.no_cpu_baseline:
            out32 = None
            if args.workload == "car.fhd" and args.dtype in ("bf16", "fp16") and "boxes" in out:
                # the SAME network on the device in both fp32 arithmetics: exact (IEEE fp32 products; module graph, torch RPN) and
                # the split-operand pipeline ("bf16x3": what --dtype fp32 times)
                from second_amd.models import SecondDetector, CAR_FHD
                out32 = {}
                det32 = SecondDetector(CAR_FHD)
                det32.load_state_dict(cpu_state)
                det32 = det32.eval().to(device)
                with torch.no_grad(), ops.fp32_mode("exact"):
                    out32["fp32_exact"] = det32.forward_points(points, offsets)
                det32.prepare_inference(torch.float32)
                with torch.no_grad():
                    out32[ops.FP32_SPLIT_LABEL] = det32.forward_points(points, offsets)
                torch.cuda.synchronize()
                del det32
            res["cpu_baseline"] = cpu_baseline(cpu_state, clouds, out if "boxes" in out else None, gpu_out_fp32=out32)
            res["detections_match_cpu"] = res["cpu_baseline"].pop("detections_match_cpu", None)
        det_count = int(out["valid"].sum().item())
        res["detections_last_step"] = det_count
        if world == 1 and args.workload == "car.fhd" and not args.no_other_configs and not args.no_cpu_baseline:
            torch.cuda.synchronize()
            res["other_configs"] = other_configs()     # child processes on the same GPU, after every timed region of this one
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


def stage_times(det, points, offsets, iters=20):
    """The reference's own stage boundaries (voxelnet.py:325-336,371-374), sync-bracketed like its timers."""
    import torch
    bs = offsets.numel() - 1

    def timeit(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3, r
    with torch.no_grad():
        t_vox, vox = timeit(lambda: det.voxel_generator.generate_device(points, offsets, mean_features=4))
        dt = det._infer_dtype
        feats = vox["mean"].to(dt) if dt is not None else vox["mean"]
        t_mid, spatial = timeit(lambda: det.middle_feature_extractor(feats, vox["coordinates"], bs, channels_last=dt is not None))
        t_rpn, preds = timeit(lambda: det.rpn(spatial))
        t_pred, _ = timeit(lambda: det.predict_device(preds, bs))
    print(json.dumps({"stage_ms_per_batch8": {"voxelize+vfe": round(t_vox, 3), "middle": round(t_mid, 3),
                                              "rpn": round(t_rpn, 3), "predict": round(t_pred, 3)}}), file=sys.stderr)


if __name__ == "__main__":
    main()
