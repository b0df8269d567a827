#!/bin/bash
# deferred accelerate_model: frames/s against the number of lanes, three repetitions of an evaluate()-style loop (results collected in a
# list, read after the loop) after a one-second warm-up of call-and-read
export PYTHONUNBUFFERED=1
for L in ${LANES:-2 3 4}; do
SEC_ACCELERATE_LANES=$L timeout 300 python - 2>/dev/null <<'PY'
import gc, json, sys, os, time
sys.argv = ["bench.py"]
import bench, torch
from second_amd import synthetic as syn, compat
sys.path.insert(0, os.path.join(bench.ROOT, "tests"))
from reference_standin import build_voxelnet
from second_amd.models import CAR_FHD
dev = torch.device("cuda")
clouds, points, offsets = bench.build_inputs(0, dev)
det, cpu_state = bench.build_detector(dev, torch.bfloat16, syn.syn_kitti_cloud(0))
net = build_voxelnet(CAR_FHD); net.load_state_dict(cpu_state); net = net.eval().cuda()
with torch.no_grad():
    vox = net.voxel_generator.generate_device(points, offsets)
ex = {"voxels": vox["voxels"], "num_points": vox["num_points_per_voxel"], "coordinates": vox["coordinates"],
      "anchors": net.anchors.unsqueeze(0).expand(8, -1, -1).contiguous()}
compat.accelerate_model(net, dtype=torch.bfloat16, deferred=True)
N = int(os.environ.get("CALLS", "240"))
with torch.no_grad():
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        r = net(ex); len(r[0])
    torch.cuda.synchronize()
    for rep in range(4):
        if os.environ.get("NOGC") == "1": gc.disable()
        t0 = time.perf_counter(); col = []
        for _ in range(N):
            col += net(ex)
        t1 = time.perf_counter()
        for d in col: d["scores"]
        torch.cuda.synchronize(); t2 = time.perf_counter()
        gc.enable()
        print(os.environ["SEC_ACCELERATE_LANES"], "lanes rep", rep, ": issue %.3f ms/call, total %.3f ms/call = %.0f frames/s" % ((t1-t0)/N*1e3, (t2-t0)/N*1e3, 8*N/(t2-t0)), net._second_amd_engine.stats["deferred_redone"], torch.cuda.memory_reserved() >> 20, "MB reserved")
PY
done
