"""A short run of the accelerated drop-in call for rocprofv3:  python tools/dropin_profile.py [fp32|fp16|bf16] [calls]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from reference_standin import build_voxelnet  # noqa: E402
from second_amd import compat, synthetic as syn  # noqa: E402
from second_amd.models import CAR_FHD  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda", 0)
    det, cpu_state = bench.build_detector(dev, torch.float32, syn.syn_kitti_cloud(0))
    clouds, points, offsets = bench.build_inputs(0, dev)
    net = build_voxelnet(CAR_FHD)
    net.load_state_dict(cpu_state)
    net = net.eval().cuda()
    with torch.no_grad():
        vox = net.voxel_generator.generate_device(points, offsets)
    fdt = torch.float16 if mode == "fp16" else torch.float32
    ex = {"voxels": vox["voxels"].to(fdt), "num_points": vox["num_points_per_voxel"], "coordinates": vox["coordinates"],
          "anchors": net.anchors.unsqueeze(0).expand(8, -1, -1).contiguous().to(fdt)}
    if mode == "fp16":
        net.half()
    compat.accelerate_model(net, dtype=torch.bfloat16 if mode == "bf16" else None, graph=os.environ.get("SEC_DROPIN_GRAPH", "1") == "1")
    with torch.no_grad():
        for _ in range(calls):
            r = net(ex)
    torch.cuda.synchronize()
    print(mode, "detections", sum(x["scores"].shape[0] for x in r))


if __name__ == "__main__":
    main()
