#!/usr/bin/env python3
"""CPU study for DESIGN.md section 9 (dense RPN conv, next step): how much accuracy would a bf16 Winograd F(2x2,3x3) form of the
RPN's 3x3 128->128 conv lose against the direct bf16 MFMA form?  Operands are bf16-valued, products accumulate in fp32 (what
the matrix cores do); the Winograd form additionally rounds the transformed input V = B^T d B and weights U = G g G^T to bf16
before the 16 per-position GEMMs.  Prints absolute errors against an exact (float64) convolution of the same operands,
before and after the bias + ReLU + bf16 output rounding that both forms share."""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
C, H, W = 128, 48, 48
x = torch.relu(torch.randn(1, C, H, W)).bfloat16().float()       # post-ReLU activations, as between RPN layers
w = (torch.randn(C, C, 3, 3) / 34).bfloat16().float()
b = torch.randn(C)
ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
direct = F.conv2d(x, w, b, padding=1)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
patches = F.pad(x, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)  # [1, C, tiles_y, tiles_x, 4, 4]
V = torch.einsum('ij,bcyxjk,lk->bcyxil', Bt, patches, Bt)
U = torch.einsum('ij,ocjk,lk->ocil', G, w, G)


def winograd(round_v, round_u):
    vq = V.bfloat16().float() if round_v else V
    uq = U.bfloat16().float() if round_u else U
    m = torch.einsum('ocil,bcyxil->boyxil', uq, vq)
    y = torch.einsum('ij,boyxjk,lk->boyxil', At, m, At)
    ty, tx = y.shape[2], y.shape[3]
    return y.permute(0, 1, 2, 4, 3, 5).reshape(1, C, ty * 2, tx * 2) + b.view(1, -1, 1, 1)


print(f"reference rms {ref.pow(2).mean().sqrt():.3f} max {ref.abs().max():.2f}")
for name, o in [("direct, fp32 accumulate", direct), ("winograd, fp32 transforms", winograd(False, False)),
                ("winograd, V -> bf16", winograd(True, False)), ("winograd, U and V -> bf16", winograd(True, True))]:
    e = (o.double() - ref).abs()
    eb = (torch.relu(o).bfloat16().double() - torch.relu(ref)).abs()
    print(f"{name:28s} pre-rounding max {e.max():.3e} rms {e.pow(2).mean().sqrt():.3e} | after ReLU + bf16 store: "
          f"max {eb.max():.3e} rms {eb.pow(2).mean().sqrt():.3e}")
