#!/usr/bin/env python3
"""Determinism stress: three captured forwards replayed concurrently on three streams, many times.  EVERY op result of every
lane (voxelise, rulebooks, sparse convs, dense, RPN convs, predict stages) is kept and compared bit for bit with the lane's
first replay; the first op in program order that changes is reported."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd"))
import torch
from second_amd import ops, synthetic as syn
from second_amd.models import SecondDetector, CAR_FHD

replays = int(sys.argv[1]) if len(sys.argv) > 1 else 300
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lanes_n = int(os.environ.get("LANES", "3"))
clouds = [syn.syn_kitti_cloud(s, num_points=7000, num_voxels=6000) for s in range(frames)]
pts, offs = syn.batch_clouds(clouds)
pts, offs = torch.from_numpy(pts).cuda(), torch.from_numpy(offs).cuda()
torch.manual_seed(0)
det = SecondDetector(CAR_FHD).cuda().prepare_inference(torch.bfloat16)
calls = []
orig_fp = det._forward_points


def fp(*a, **k):
    calls.clear()
    return orig_fp(*a, **k)


det._forward_points = fp


def flat(res, prefix):
    out = []
    if torch.is_tensor(res):
        out.append((prefix, res))
    elif isinstance(res, dict):
        for k, v in res.items():
            if k == "site_table":            # workspaces: hash-slot placement legitimately depends on the order of the atomics
                continue
            out += flat(v, f"{prefix}.{k}")
    elif isinstance(res, tuple) and len(res) == 2 and torch.is_tensor(res[0]) and torch.is_tensor(res[1]) and res[1].dtype in (torch.int32, torch.bool) \
            and res[0].dim() >= 2 and res[1].shape[0] == res[0].shape[0] and ("keep_defined" in prefix or "masked" in prefix):
        out.append((prefix, res))                      # (values, count-or-mask): compared on the defined part only
    elif isinstance(res, (list, tuple)):
        for i, v in enumerate(res):
            out += flat(v, f"{prefix}[{i}]")
    return out


def meaningful(name, res):
    """Only the defined part of an op's result: keep lists up to num_keep, detections under their valid mask."""
    if name == "nms_sorted":
        keep, num_keep = res
        return {"num_keep": num_keep, "keep_defined": (keep, num_keep)}
    if name == "predict_finalize":
        return {"valid": res["valid"], "masked": {k: (v, res["valid"]) for k, v in res.items() if k != "valid"}}
    return res


ops.set_op_hook(lambda name, fn, a, kw, res: calls.append((name, meaningful(name, res))))
with torch.no_grad():
    det.forward_points(pts, offs)
    det.calibrate(pts, offs)
    lanes = []
    for li in range(lanes_n):
        replay, outs = det.make_graphed(pts, offs)
        tens = []
        for ci, (name, res) in enumerate(calls):
            tens += flat(res, f"{ci:02d}.{name}")
        tens += flat({"valid": outs["valid"], "masked": {k: (v, outs["valid"]) for k, v in outs.items() if k != "valid"}}, "99.out")
        lanes.append((replay, tens, torch.cuda.Stream()))
ops.set_op_hook(None)
print(len(lanes[0][1]), "tensors per lane")


def value(t):
    if torch.is_tensor(t):
        return t
    vals, sel = t
    if sel.dtype == torch.bool:
        m = sel
        while m.dim() < vals.dim():
            m = m.unsqueeze(-1)
        return torch.where(m, vals, torch.zeros_like(vals))
    ar = torch.arange(vals.shape[1], device=vals.device).unsqueeze(0)
    return torch.where(ar < sel.unsqueeze(1), vals, torch.full_like(vals, -7))


def run_all():
    for replay, _, st in lanes:
        with torch.cuda.stream(st):
            replay()
    torch.cuda.synchronize()


run_all()
run_all()
ref = [[value(t).clone() for _, t in tens] for _, tens, _ in lanes]
bad = 0
first_seen = {}
t0 = time.time()
for it in range(replays):
    run_all()
    for li, (_, tens, _) in enumerate(lanes):
        for ti, (name, t) in enumerate(tens):
            t = value(t)
            if not torch.equal(t, ref[li][ti]):
                d = t != ref[li][ti]
                if t.is_floating_point():
                    d &= ~(torch.isnan(t) & torch.isnan(ref[li][ti]))
                if not bool(d.any()):
                    continue
                bad += 1
                key = (li, name)
                first_seen.setdefault(key, it)
                if bad <= 12:
                    nz_now, nz_ref = int((t != 0).sum()), int((ref[li][ti] != 0).sum())
                    vals = t[d][:6].float().tolist()
                    print(f"replay {it} lane {li}: {name} {tuple(t.shape)} changed in {int(d.sum())} elements; nonzero now {nz_now} ref {nz_ref}; ptr {t.data_ptr():#x} bytes {t.numel() * t.element_size()}; values {vals}", flush=True)
                break            # only the first op in program order
print(f"{replays} rounds x {lanes_n} lanes: {bad} replays with a changed tensor, {time.time() - t0:.1f} s")
from collections import Counter
print(Counter(name for (_, name) in first_seen).most_common(10))
