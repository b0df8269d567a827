#!/usr/bin/env python3
"""Uninitialised-read hunt: fill the caching allocator's free blocks with a poison pattern (NaN / huge ints) before every forward,
so that any kernel that reads memory it (or an earlier kernel of the step) did not write changes the result.  Compares eager,
static-capacity and hipGraph forwards of the same clouds, several rounds, different poison patterns."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd"))
import torch
from second_amd import synthetic as syn
from second_amd.models import SecondDetector, CAR_FHD, InFlightRunner


def poison(pattern, mb=3000):
    n = mb * 1024 * 1024 // 4
    t = torch.empty(n, dtype=torch.int32, device="cuda")
    t.fill_(pattern)
    # also a few differently sized blocks so that small requests land on poisoned memory too
    small = [torch.full((k,), pattern, dtype=torch.int32, device="cuda") for k in (256, 4096, 65536, 1 << 20, 1 << 22) for _ in range(8)]
    torch.cuda.synchronize()
    del t, small


def same(a, b):
    if not torch.equal(a["valid"], b["valid"]):
        return "valid differs"
    m = a["valid"]
    if not torch.equal(a["scores"][m], b["scores"][m]):
        return "scores differ"
    if not torch.equal(a["boxes"][m], b["boxes"][m]):
        return f"boxes differ ({int((a['boxes'][m] != b['boxes'][m]).any(-1).sum())} rows)"
    return None


def main():
    torch.manual_seed(0)
    det = SecondDetector(CAR_FHD).cuda().prepare_inference(torch.bfloat16)
    pts, offs = syn.batch_clouds([syn.syn_kitti_cloud(s, num_points=7000, num_voxels=6000) for s in range(3)])
    pts, offs = torch.from_numpy(pts).cuda(), torch.from_numpy(offs).cuda()
    with torch.no_grad():
        ref = det.forward_points(pts, offs)
        ref = {k: v.clone() for k, v in ref.items()}
        det.calibrate(pts, offs)
        bad = 0
        for rnd, pat in enumerate([0x7fc00000, -1, 0x7f7f7f7f, 0x3f803f80, 12345678, 0x7fc00000]):
            poison(pat)
            e = det.forward_points(pts, offs)
            r = same(ref, e)
            if r: bad += 1; print(f"round {rnd} pattern {pat:#x}: EAGER {r}")
            poison(pat)
            s = det.forward_points(pts, offs, static=True)
            r = same(ref, s)
            if r: bad += 1; print(f"round {rnd} pattern {pat:#x}: STATIC {r}")
            poison(pat)
            runner = InFlightRunner(det, pts, offs, inflight=3)
            for _ in range(7):
                runner.step()
            runner.synchronize()
            for li, o in enumerate(runner.outputs):
                r = same(ref, o)
                if r: bad += 1; print(f"round {rnd} pattern {pat:#x}: GRAPH lane {li} {r}")
            del runner
        print("mismatches:", bad)


if __name__ == "__main__":
    main()
