#!/usr/bin/env python3
"""Which non-library (torch / runtime) kernels still run inside one static forward of the bench configuration?  torch.profiler with
shapes and stacks over one eager static step; prints every kernel that is not one of ours with the op that launched it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "second.pytorch_amd")]
import torch
import bench
from second_amd import synthetic as syn
bench.WL = dict(bench.WORKLOADS["car.fhd"])
dev = torch.device("cuda")
clouds, points, offsets = bench.build_inputs(0, dev)
det, _ = bench.build_detector(dev, torch.bfloat16, syn.syn_kitti_cloud(0))
with torch.no_grad():
    det.calibrate(points, offsets)
    for _ in range(3):
        det.forward_points(points, offsets, static=True)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        det.forward_points(points, offsets, static=True)
        torch.cuda.synchronize()
for e in prof.events():
    if e.device_type.name == "CUDA" and "sec::" not in e.name and "rocprim" not in e.name:
        print("GPU kernel:", e.name[:150], f"{e.device_time:.1f} us")
print("---- CPU ops that launched them")
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    if e.device_time_total > 0 and "sec" not in e.key:
        print(e.key, e.input_shapes, f"{e.device_time_total:.1f}", "\n   ", "\n    ".join(s for s in e.stack[:6]))
