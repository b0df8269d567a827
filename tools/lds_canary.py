#!/usr/bin/env python3
"""Does the RPN conv kernel write outside its LDS allocation?  (needs the -DSEC_CONV_TIMELINE build, SEC_HIP_LIB=...)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd"))
import torch
from second_amd import ops, runtime as rt
l = rt.lib()
x = torch.relu(torch.randn(8, 128, 200, 176, device="cuda")).bfloat16().contiguous(memory_format=torch.channels_last)
w = (torch.randn(128, 128, 3, 3, device="cuda") / 34).bfloat16()
pk = ops.conv2d_pack_weight(w)
bias = torch.randn(128, device="cuda")
err = torch.zeros(1, dtype=torch.int32, device="cuda")
first = torch.full((1,), 1 << 30, dtype=torch.int32, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for kind in ("none", "conv"):
    err.zero_(); first.fill_(1 << 30)
    torch.cuda.synchronize()
    for it in range(200):
        if kind == "conv":
            with torch.cuda.stream(s1):
                for _ in range(3):
                    ops.conv2d_nhwc(x, pk, bias, 128, 3, 1, 1, relu=True)
        with torch.cuda.stream(s2):
            l.sec__debug_lds_canary(ctypes.c_void_p(err.data_ptr()), ctypes.c_void_p(first.data_ptr()), 512, 60000,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
    print(f"load {kind}: corrupted canary words {int(err.item())}, lowest index {int(first.item())}")
