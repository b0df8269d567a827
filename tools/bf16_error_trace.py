#!/usr/bin/env python3
"""Per-layer error trace of the benchmarked configuration (8 frames, 16-bit features, static capacities) against the fp32 CPU
forward, and the detection match with every miss attributed (the table committed as profiles/rNN_bf16_error_trace.txt).

    python tools/bf16_error_trace.py [--dtype bf16|fp16] [--frames 8]

Shares tests/e2e_trace.py with tests/test_gpu_e2e.py (the test asserts bounds on the same numbers)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "second.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--frames", type=int, default=8)
    args = ap.parse_args()
    import e2e_trace as T
    from oracle.cpu_forward import forward_frame
    from second_amd import synthetic as syn
    from second_amd.models import SecondDetector, CAR_FHD
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    ulp = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11}[args.dtype]     # unit roundoff (round to nearest)
    clouds = [syn.syn_kitti_cloud(s) for s in range(args.frames)]
    det = T.trained_like_detector(CAR_FHD, clouds[0])
    results = [forward_frame(det, c, collect=True) for c in clouds]
    traces = [r["trace"] for r in results]
    gpu = SecondDetector(CAR_FHD).eval()
    gpu.load_state_dict(det.state_dict())
    gpu = gpu.cuda().prepare_inference(dt)
    pts, offs = syn.batch_clouds(clouds)
    pts, offs = torch.from_numpy(pts).cuda(), torch.from_numpy(offs).cuda()
    with torch.no_grad():
        gpu.calibrate(pts, offs)
    calls, out = T.run_device_trace(gpu, pts, offs)
    rows = T.sparse_stage_errors(calls, traces, ulp, gpu.middle_feature_extractor.sparse_shape)
    rows += T.dense_stage_errors(calls, gpu, det, traces, ulp, single_frames=min(2, args.frames))
    print(f"# car.fhd, {args.frames} synthetic KITTI frames (17 000 points -> 16 000 voxels each), {args.dtype} features, static capacities;")
    print("# reference: oracle/cpu_forward.py in fp32 from the raw points.  cumulative = max |device - cpu| / max |cpu| of the layer output;")
    print("# single = the layer recomputed on the CPU from the device's own 16-bit input, in units of (unit roundoff of the stored result, 2^-8 |x| for bf16, + 1e-4 of range)")
    print(T.format_table(rows))
    found, total, counts, missed = T.match_detections(out, results)
    why = T.attribute_misses(calls, results, missed, CAR_FHD["nms_score_threshold"])
    cand = [len(t["candidate_scores"]) for t in traces]
    print(f"# candidates above the score threshold per frame (CPU): {cand}")
    print(f"# detections: {found} of {total} CPU detections found on the device (0.25 m, 0.05 score); (device, cpu) counts per frame {counts}")
    print(f"# misses by cause: {why}")


if __name__ == "__main__":
    main()
