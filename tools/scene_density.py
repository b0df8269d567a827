"""Experiment driver: frames/s of the car.fhd forward on the bench's sparse scene and on the dense seeded scene
(synthetic.syn_kitti_cloud(scene="dense")) for one setting of the RPN live-tile machinery.

    SEC_RPN_LIST_MAX_LIVE=60 python tools/scene_density.py --scene dense --skip 1 --inflight 4

Prints one JSON line.  Clouds are cached under /tmp (the dense generator takes ~3 s per frame)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

from second_amd import synthetic as syn  # noqa: E402
from second_amd.models import InFlightRunner  # noqa: E402


def clouds_of(scene, batch):
    path = f"/tmp/sec_scene_{scene}_{batch}.npz"
    if os.path.exists(path):
        z = np.load(path)
        return [z[f"c{i}"] for i in range(batch)]
    cl = [syn.syn_kitti_cloud(s, scene=scene) for s in range(batch)]
    np.savez(path, **{f"c{i}": c for i, c in enumerate(cl)})
    return cl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="dense")
    ap.add_argument("--skip", type=int, default=1)
    ap.add_argument("--lazy", type=int, default=1)
    ap.add_argument("--inflight", type=int, default=4)
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    det, _ = bench.build_detector(dev, torch.bfloat16, syn.syn_kitti_cloud(0))
    det.rpn.skip_background, det.rpn.lazy_background = bool(args.skip), bool(args.lazy and args.skip)
    pts, offs = syn.batch_clouds(clouds_of(args.scene, 8))
    pts, offs = torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev)
    with torch.no_grad():
        det.calibrate(pts, offs)
        runner = InFlightRunner(det, pts, offs, inflight=args.inflight, serialize_rpn=args.inflight > 1)
        for _ in range(300):
            runner.step()
        torch.cuda.synchronize()
        best = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                runner.step()
            torch.cuda.synchronize()
            best.append(time.perf_counter() - t0)
        runner.synchronize()
        one = det.make_graphed(pts, offs)[0]
        for _ in range(5):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            one()
        torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / 50
    live = det.rpn.last_live_counts.sum(1).tolist() if det.rpn.last_live_counts is not None else None
    print(json.dumps({"scene": args.scene, "skip": args.skip, "lazy": args.lazy, "inflight": args.inflight,
                      "list_max_live": os.environ.get("SEC_RPN_LIST_MAX_LIVE", "75"),
                      "frames_per_s": round(8 * args.steps / float(np.median(best)), 1), "one_step_graph_ms": round(lat * 1e3, 4),
                      "live_tiles": live}), flush=True)


if __name__ == "__main__":
    main()
