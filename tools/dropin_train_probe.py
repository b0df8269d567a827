#!/usr/bin/env python3
"""bf16 captured training step vs the fp32 module graph on the same example: error statistics of the head logits and gradients
(the numbers behind the bounds of tests/test_gpu_dropin_train.py::test_fused_training_against_the_fp32_module_graph)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "second.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import test_gpu_dropin_train as T
from second_amd import compat
net = T._net(1); ex = T._example(net, seeds=(2, 3)); plain = T._net(1); plain.load_state_dict(net.state_dict())
compat.accelerate_model(net, train_dtype=torch.bfloat16)
out = net(ex); out["loss"].backward(); po = plain(ex); po["loss"].backward()
a, b = out["cls_preds"].reshape(-1).float(), po["cls_preds"].reshape(-1).float().detach()
d = (a - b).abs()
print("cls_preds: max|ref|", b.abs().max().item(), "rms ref", b.pow(2).mean().sqrt().item(), "L2 rel", (d.pow(2).sum().sqrt() / b.pow(2).sum().sqrt()).item())
print("quantiles of |diff|:", [round(torch.quantile(d[::7], q).item(), 5) for q in (0.5, 0.9, 0.99, 0.999)], "max", d.max().item())
print("frac > 0.1:", (d > 0.1).float().mean().item(), " frac > 0.3:", (d > 0.3).float().mean().item())
ga, gb = T._grads(net), T._grads(plain)
for n in gb:
    print(f"{n:60s} rel-max {((ga[n]-gb[n]).abs().max()/(gb[n].abs().max()+1e-20)).item():.4f}  L2 rel {((ga[n]-gb[n]).norm()/(gb[n].norm()+1e-20)).item():.4f}")
