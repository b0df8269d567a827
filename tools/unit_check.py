#!/usr/bin/env python3
"""Which unit returns different results beside the RPN conv kernel?  (needs the -DSEC_NMS_DEBUG build via SEC_HIP_LIB)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd"))
import torch
from second_amd import ops, runtime as rt
l = rt.lib()
torch.manual_seed(0)
data = (torch.rand(3000, 6, device="cuda") * 3 + 0.5).contiguous()
x = torch.relu(torch.randn(8, 128, 200, 176, device="cuda")).bfloat16().contiguous(memory_format=torch.channels_last)
w = (torch.randn(128, 128, 3, 3, device="cuda") / 34).bfloat16()
pk = ops.conv2d_pack_weight(w)
bias = torch.randn(128, device="cuda")
ma = torch.randn(4096, 4096, device="cuda").bfloat16(); mb = torch.randn(4096, 4096, device="cuda").bfloat16()
cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
for kind in ("none", "mm", "conv"):
    cnt.zero_()
    torch.cuda.synchronize()
    for it in range(150):
        for s in (s1, s3):
            with torch.cuda.stream(s):
                for _ in range(3):
                    if kind == "conv":
                        ops.conv2d_nhwc(x, pk, bias, 128, 3, 1, 1, relu=True)
                    elif kind == "mm":
                        torch.mm(ma, mb)
        with torch.cuda.stream(s2):
            l.sec__debug_unit_check(ctypes.c_void_p(data.data_ptr()), 3000, 768, 400, ctypes.c_void_p(cnt.data_ptr()),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
    print(f"load {kind:5s}: repeated loads differ {cnt[0].item()}, sin/cos differ {cnt[1].item()}, fp32 arithmetic differs {cnt[2].item()}, LDS read-back differs {cnt[3].item()}")
