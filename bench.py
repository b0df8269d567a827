#!/usr/bin/env python3
"""Headline benchmark: frames/s of the car.fhd VoxelNet forward (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One *step* = one pass of the whole hot path over one batch of 8 synthetic KITTI clouds that are already
resident in HBM: points_to_voxel (+SimpleVoxel mean) -> 14 sparse conv layers (rulebooks + fused
indice_conv) -> dense -> RPNV2 (bf16, hand-written MFMA convs) -> decode / top-k / rotated NMS, detections left on the device.
Per-frame data parallel: every rank runs its own batch, no data-path collective ("weak" scaling).
Default launch mode: ONE hipGraph replay per step; the batch is captured as two independent 4-frame chains on two
streams of that graph (--branches 2), which overlaps the path's latency-bound kernels.

Prints ONE JSON line on rank 0 with, besides the contract fields,
  roofline      -- the SubMConv3d 64->64 gather-GEMM kernel (the kernel BASELINE.json's metric names): algorithmic bytes
                   per launch / mean launch duration (HIP events around 100 re-issues of the very launch the timed graph
                   runs, right after the timed region), against the 8 TB/s HBM peak; `traffic` from committed PMC passes;
  roofline_mfma -- the RPN 3x3 conv (largest share of the step, MFMA bound), timed the same way;
  cpu_baseline  -- the same forward on the host cores through the CPU oracle (kind "port"), rank 0, N=1 only.
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
    (the kernel is ~40 us, far above the ~5 us host launch cost, so the stream never drains)."""
    from second_amd import ops
    a = call["args"]
    for _ in range(5):
        ops.indice_conv(*a["pos"], **a["kw"])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    for _ in range(reps):
        ops.indice_conv(*a["pos"], **a["kw"])
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


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
            "frac": round(flop / t / 2.5e15, 4), "kernel": "k_conv2d_halo_reg<bf16,128> (RPN 3x3 128->128, 6 launches per step)",
            "launch_us": round(t * 1e6, 2), "launches_timed": reps, "flop_per_launch": flop}


# Other BASELINE configs (parity-test cases; timed only on request with --workload, never the default line):
#   nusc.pp  = nuscenes/all.pp.largea (PointPillars), nusc.fhd = nuscenes/all.fhd (block-filtered voxels, 10 classes)
WORKLOADS = {
    "car.fhd": dict(cfg="CAR_FHD", batch=8, metric="frames/sec VoxelNet fwd (car.fhd, ~16k active voxels)",
                    desc="car.fhd.config VoxelNet forward (voxelise+VFE+SpMiddleFHD+RPNV2+rotated NMS), inference, "
                         "batch=8 synthetic KITTI clouds/GPU (17000 pts, 16000 voxels each), random-init weights, "
                         "inputs resident in HBM"),
    "nusc.pp": dict(cfg="ALL_PP_LARGEA", batch=4, metric="frames/sec VoxelNet fwd (nuscenes/all.pp.largea)",
                    desc="nuscenes/all.pp.largea VoxelNet forward (voxelise+PillarFeatureNet+scatter+RPNV2 3 blocks+"
                         "axis-aligned NMS), inference, batch=4 synthetic 10-sweep NuScenes clouds/GPU (<= 120k pts), "
                         "random-init weights, inputs resident in HBM"),
    "nusc.fhd": dict(cfg="ALL_FHD_NUSC", batch=4, metric="frames/sec VoxelNet fwd (nuscenes/all.fhd)",
                     desc="nuscenes/all.fhd VoxelNet forward (block-filtered voxelise+SpMiddleFHD on 1984x1984x40+RPNV2+"
                          "axis-aligned NMS), inference, batch=4 synthetic 10-sweep NuScenes clouds/GPU (<= 120k pts), "
                          "random-init weights, inputs resident in HBM"),
}
WL = WORKLOADS["car.fhd"]


def build_inputs(rank, device, order="shuffle"):
    from second_amd import synthetic as syn
    if WL["cfg"] != "CAR_FHD":
        rng = (-50, -50, -5, 50, 50, 3) if WL["cfg"] == "ALL_PP_LARGEA" else (-49.6, -49.6, -5, 49.6, 49.6, 3)
        clouds = [syn.syn_nusc_cloud(rank * BATCH + s, num_points=120000, point_cloud_range=rng) for s in range(WL["batch"])]
        pts, offs = syn.batch_clouds(clouds)
        return clouds, torch.from_numpy(pts).to(device), torch.from_numpy(offs).to(device)
    clouds = [syn.syn_kitti_cloud(rank * BATCH + s) for s in range(BATCH)]
    if order == "sorted":   # experiment: points in spatial (z, y, x) order instead of the shuffled order of SURVEY 8d
        clouds = [c[np.lexsort((c[:, 0], c[:, 1], c[:, 2]))] for c in clouds]
    pts, offs = syn.batch_clouds(clouds)
    return clouds, torch.from_numpy(pts).to(device), torch.from_numpy(offs).to(device)


def build_detector(device, dtype):
    from second_amd import models
    from second_amd.models import SecondDetector
    torch.manual_seed(0)
    det = SecondDetector(getattr(models, WL["cfg"]))
    g = torch.Generator().manual_seed(1)
    for m in det.modules():  # BN in eval mode with non-trivial statistics so that folding is exercised
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.copy_(torch.empty_like(m.running_mean).uniform_(-0.1, 0.1, generator=g))
            m.running_var.copy_(torch.empty_like(m.running_var).uniform_(0.5, 1.5, generator=g))
    det.eval()
    cpu_state = {k: v.clone() for k, v in det.state_dict().items()}
    det = det.to(device)
    if dtype != torch.float32:
        det.prepare_inference(dtype)
    return det, cpu_state


# ------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(cpu_state, clouds, budget_s=20.0):
    """The same forward on the host through the oracle ("port"): single-threaded oracle for voxelise /
    rulebook / indice_conv / NMS (like the reference's worker-side C++), torch CPU (all cores) for the RPN."""
    from oracle import oracle as orc
    from second_amd.models import SecondDetector, CAR_FHD, decode_boxes
    cores = min(os.cpu_count() or 1, 32)   # more threads than that only oversubscribe the 200x176 convs
    torch.set_num_threads(cores)
    det = SecondDetector(CAR_FHD)
    det.load_state_dict(cpu_state)
    det.eval()
    cfg = CAR_FHD
    seq = list(det.middle_feature_extractor.middle_conv.children())
    anchors = det.anchors

    def one(cloud):
        v = orc.points_to_voxel(cloud, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_points_per_voxel"], cfg["max_voxels"])
        feat = orc.simple_voxel_mean(v["voxels"], v["num_points_per_voxel"], 4)
        idx = np.concatenate([np.zeros((v["voxel_num"], 1), np.int32), v["coordinates"]], 1)
        shape = det.middle_feature_extractor.sparse_shape
        cache = {}
        i = 0
        while i < len(seq):
            conv, bn = seq[i], seq[i + 1]
            if conv.subm:
                if conv.indice_key not in cache:
                    cache[conv.indice_key] = orc.rulebook_subm(idx, 1, shape, conv.kernel_size)
                out_idx, pairs, num = cache[conv.indice_key]
                n_out = len(idx)
            else:
                out_idx, pairs, num, oshape = orc.rulebook_conv(idx, 1, shape, conv.kernel_size, conv.stride, conv.padding)
                n_out = len(out_idx)
            y = orc.indice_conv(feat, conv.weight.detach().numpy(), pairs, num, n_out, acc64=False)
            scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().numpy()
            shift = (bn.bias - bn.running_mean * torch.from_numpy(scale)).detach().numpy()
            feat = np.maximum(y * scale + shift, 0).astype(np.float32)
            if not conv.subm:
                idx, shape = out_idx, [int(s) for s in oshape]
            i += 3
        dense = orc.sparse_to_dense(feat, idx, 1, shape)
        x = torch.from_numpy(dense).view(1, -1, shape[1], shape[2])
        with torch.no_grad():
            preds = det.rpn(x)
            cls = torch.sigmoid(preds["cls_preds"].reshape(-1))
            keep = cls >= cfg["nms_score_threshold"]
            sc, ix = torch.topk(cls[keep], min(cfg["nms_pre_max_size"], int(keep.sum())))
            sel = torch.nonzero(keep).squeeze(1)[ix]
            boxes = decode_boxes(preds["box_preds"].reshape(-1, 7)[sel], anchors[sel])
            dets = torch.cat([boxes[:, [0, 1, 3, 4, 6]], sc[:, None]], 1).numpy()
        k = orc.rotate_nms_sorted(dets, cfg["nms_iou_threshold"], "cpu")[:cfg["nms_post_max_size"]]
        return len(k)

    t0 = time.perf_counter()
    n = 0
    while n < len(clouds) and (n < 1 or time.perf_counter() - t0 < budget_s):
        one(clouds[n])
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} synthetic KITTI frame(s) (17k pts, 16k voxels), fp32; oracle (1 thread) for voxelise/"
                      f"rulebook/indice_conv/NMS + torch CPU ({cores} threads) for the RPN; {dt:.1f} s"}


# ------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--mode", default="graph", choices=["graph", "static", "eager"],
                    help="graph: static-capacity forward captured in a hipGraph (default); static: same, eager "
                         "launches; eager: the dynamic-shape drop-in path (host syncs per strided layer)")
    ap.add_argument("--point-order", default="shuffle", choices=["shuffle", "sorted"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stages", action="store_true", help="also print per-stage timings to stderr")
    ap.add_argument("--inflight", type=int, default=3,
                    help="graph mode: number of steps (graph replays, each a full pass over the batch with its own activation "
                         "buffers) kept in flight on separate HIP streams; the latency-bound sparse stages of one step then "
                         "overlap the MFMA-bound RPN of another.  1 = strictly one step at a time")
    ap.add_argument("--branches", type=int, default=1,
                    help="graph mode: capture the batch as this many independent frame-group chains on separate streams of "
                         "ONE hipGraph (the single-step-latency variant of --inflight); 1 = a single chain")
    ap.add_argument("--workload", default="car.fhd", choices=sorted(WORKLOADS),
                    help="car.fhd (default, the BASELINE metric) or another BASELINE config for a side measurement")
    args = ap.parse_args()
    global WL
    WL = WORKLOADS[args.workload]
    if args.workload != "car.fhd":
        args.no_cpu_baseline = True

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (use gpurun)"
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU is started by torch.distributed.run "
              f"(python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py "
              f"--gpus {args.gpus}); running {world} rank(s)", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]

    from second_amd import ops
    clouds, points, offsets = build_inputs(rank, device, args.point_order)
    det, cpu_state = build_detector(device, dtype)

    # first subm2 layer (64->64 SubM on the 11x400x352 grid): the largest 64->64 3x3x3 launch of the forward
    timer = ConvCapture(lambda m: m["cin"] == 64 and m["cout"] == 64 and m["kvol"] == 27 and m["n_in"] == m["n_out"]
                        and m["n_out"] > 20000)
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
            runner = InFlightRunner(det, points, offsets, inflight=args.inflight, branches=args.branches)
            graph_parts = runner.parts      # branches > 1: the roofline probe below times one branch's launch
            replays = runner.replays
            outs = runner.outputs[-1]
            out = {"valid": torch.cat([o["valid"] for o in outs])} if args.branches > 1 else outs
            step = runner.step
        elif args.mode == "static":
            step = lambda: det.forward_points(points, offsets, static=True)
        else:
            step = lambda: det.forward_points(points, offsets)
        for _ in range(args.warmup):
            r = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r = step()
        barrier()
        elapsed = time.perf_counter() - t0
        latency_ms = None
        if args.mode == "graph":   # one step alone, start to finish (what --inflight 1 would run back to back)
            t1 = time.perf_counter()
            for _ in range(20):
                replays[0]()
                torch.cuda.synchronize()
            latency_ms = round((time.perf_counter() - t1) / 20 * 1e3, 4)
        if args.mode != "graph":
            out = r
        if args.mode != "eager":
            det.check_overflow()
        # per-kernel timing of the SubMConv3d kernel: capture the launch arguments during one eager forward
        # right after the timed region, then re-issue that launch 100x back-to-back between two HIP events on
        # the launch stream (events cannot be timed inside a captured graph; a single event pair around one
        # ~40 us launch would add ~10 us of launch gap).  profiles/ holds the rocprofv3 cross-check.
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
        t_kernel = time_kernel(timer.call) if timer.call is not None else None
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
                t8 = time_kernel(m8)
                aux = {"frames": WL["batch"], "rows": rows8, "pairs": pairs8, "launch_us": round(t8 * 1e6, 2),
                       "frac": round(b8 / t8 / 1e9 / HBM_PEAK_GBS, 4)}
        ops.set_conv_profiler(None)
        roof_mfma = time_rpn_conv(det, WL["batch"]) if args.dtype == "bf16" else None

    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # roofline of the SubMConv3d 64->64 kernel: algorithmic bytes / mean measured launch time
    roof = None
    if timer.call is not None:
        meta = timer.call
        s = 2 if meta["dtype"] != torch.float32 else 4
        rows_kernel = meta["n_out"] >= 32768      # launch capacity decides the kernel (sec_indice_conv_fwd dispatch)
        rows = int(meta["num_out_dev"][0].item()) if meta.get("num_out_dev") is not None else meta["n_out"]
        pairs = int((meta["nbr_out"][:rows] >= 0).sum().item())
        meta = dict(meta, n_out=rows)
        b_alg = s * (pairs * meta["cin"] + meta["n_out"] * meta["cout"]) + 8 * pairs + s * meta["kvol"] * meta["cin"] * meta["cout"]
        t_mean = t_kernel
        ach = b_alg / t_mean / 1e9
        # HBM-side bytes per launch from the committed PMC passes of this kernel on this (seeded, deterministic)
        # workload -- profiles/r01_k_traffic.json says how they were collected; null if the shapes do not match
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_k_traffic.json")) as f:
                tj = json.load(f)
            for ent in tj["entries"]:
                if ent["rows"] == meta["n_out"] and ent["pairs"] == pairs and meta["mfma"] and rows_kernel:
                    traffic = ent["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": (("k_conv_rows<bf16,64,64,27>" if rows_kernel else "k_conv_mfma_sk<bf16,64,64>") +
                           f" (SubMConv3d subm2, {frames_in_launch} frames per launch)") if meta["mfma"] else "k_conv_generic",
                "launch_us": round(t_mean * 1e6, 2), "launches_timed": 100, "alg_bytes_per_launch": b_alg,
                "rows": meta["n_out"], "pairs": pairs, "frac_of_6.29TBs_measured_peak": round(ach / 6290.0, 4),
                "same_layer_as_one_full_batch_launch": aux}

    if args.stages and rank == 0:
        stage_times(det, points, offsets)

    if rank == 0:
        frames = WL["batch"] * args.steps * world
        res = {
            "metric": WL["metric"], "value": round(frames / elapsed, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": WL["desc"],
                       "frames_per_step_per_gpu": WL["batch"], "parallelism": f"frame-dp{world}", "launch_mode": args.mode,
                       "graph_branches": args.branches if args.mode == "graph" else None,
                       "steps_in_flight": max(1, args.inflight) if args.mode == "graph" else 1,
                       "single_step_latency_ms": latency_ms},
            "roofline": roof,
            "roofline_mfma": roof_mfma,     # second-largest consumer by kind: the dense RPN conv, MFMA bound
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cpu_state, clouds)
        det_count = int(out["valid"].sum().item())
        res["detections_last_step"] = det_count
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
