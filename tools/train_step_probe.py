#!/usr/bin/env python3
"""Side measurement (BASELINE config 3 shape): one car.fhd training step -- voxelise, SpMiddleFHD + RPNV2 forward in train
mode (BatchNorm statistics), a surrogate loss on the three heads, backward through the sparse stack (sec_indice_conv_bwd,
sec_dense_to_sparse) and an SGD update -- eager mode, batch 4.  Not a bench line: the reference's target assignment and
losses need ground truth that is not in this environment."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd")); sys.path.insert(0, ROOT)
import torch
from second_amd import synthetic as syn
from second_amd.models import SecondDetector, CAR_FHD

dev = torch.device("cuda", 0)
torch.manual_seed(0)
batch = int(os.environ.get("BATCH", "4"))
det = SecondDetector(CAR_FHD).to(dev).train()
opt = torch.optim.SGD(det.parameters(), lr=1e-4, momentum=0.9)
pts, offs = syn.batch_clouds([syn.syn_kitti_cloud(s) for s in range(batch)])
pts, offs = torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev)


def step(autocast):
    with torch.no_grad():
        vox = det.voxel_generator.generate_device(pts, offs, mean_features=4)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        preds = det.network_forward(vox["mean"], vox["coordinates"], batch)
        loss = sum((v.float() ** 2).mean() for v in preds.values())
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss


for autocast in (False,):
    for _ in range(3):
        l = step(autocast)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        l = step(autocast)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"car.fhd training step, batch {batch}, fp32{' + bf16 autocast' if autocast else ''}: {dt * 1e3:.2f} ms/step "
          f"({batch / dt:.1f} frames/s), loss {l.item():.4f}")
