#!/usr/bin/env python3
"""Self-checking probe: do packed fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) return wrong results while
another wave of the CU runs a dense MFMA loop?  (the system-level finding is tools/nms_stress.py's.)

Three scenarios, the victim kernel (tools/pkfp32_repro.hip::k_victim, bitwise packed-vs-scalar comparison in every lane) launched
back to back on one stream while other streams run
    alone      nothing,
    mfma       a generic back-to-back v_mfma_f32_32x32x16_bf16 loop (k_mfma_loop: matrix pipe saturated, no LDS / memory),
    rpnconv    the product's RPN 3x3 conv, k_conv2d_halo_reg via sec_conv2d_nhwc (the kernel beside which round 2 saw the fault).
Prints one JSON line per scenario: victim launches, lane-evaluations checked, mismatches per instruction kind and per lane group.
Any mismatch = the hazard reproduced in isolation; none in `mfma` but some in `rpnconv` = specific to what that kernel does."""
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd"))
SO = os.path.join(ROOT, "second.pytorch_amd", "lib", "libpkfp32_repro.so")


def build():
    src = os.path.join(ROOT, "tools", "pkfp32_repro.hip")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-o", SO, src])
    return SO


def main():
    import torch
    from second_amd import ops
    lib = ctypes.CDLL(build())
    lib.pk_launch_victim.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
    lib.pk_launch_victim_load.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.pk_launch_mfma.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    secs = float(os.environ.get("SECONDS_PER_SCENARIO", "3"))
    blocks, iters = 2048, 4000                  # 8 waves of victim work per CU, ~100 us per launch
    x = torch.relu(torch.randn(8, 128, 200, 176, device="cuda")).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(128, 128, 3, 3, device="cuda") / 34).bfloat16()
    pk, bias = ops.conv2d_pack_weight(w), torch.randn(128, device="cuda")
    sink = torch.zeros(16, device="cuda")
    data = torch.randn(64 * 1024 * 1024, device="cuda")            # 256 MB: the loads miss the L2
    s_v, s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    for scen in ("alone", "mfma", "rpnconv", "alone"):
        err = torch.zeros(76, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        t0, launches = time.time(), 0
        while time.time() - t0 < secs:
            for _ in range(10):
                if scen == "mfma":
                    for s in (s_a, s_b):
                        assert lib.pk_launch_mfma(ctypes.c_void_p(sink.data_ptr()), 1024, 3000, ctypes.c_void_p(s.cuda_stream)) == 0
                elif scen == "rpnconv":
                    for s in (s_a, s_b):
                        with torch.cuda.stream(s):
                            ops.conv2d_nhwc(x, pk, bias, 128, 3, 1, 1, relu=True)
                assert lib.pk_launch_victim(ctypes.c_void_p(err.data_ptr()), blocks, iters, 1.0 + 0.001 * launches,
                                            ctypes.c_void_p(s_v.cuda_stream)) == 0
                assert lib.pk_launch_victim_load(ctypes.c_void_p(data.data_ptr()), data.numel() // 4, ctypes.c_void_p(err.data_ptr()),
                                                 blocks, 400, ctypes.c_void_p(s_v.cuda_stream)) == 0
                launches += 1
            torch.cuda.synchronize()
        e = err.cpu().tolist()
        lanes = e[:64]
        print(json.dumps({"scenario": scen, "victim_launches": launches, "completed": e[67],
                          "checked_results": launches * blocks * 256 * iters * 16,
                          "mismatches": {"v_pk_fma_f32": e[64], "v_pk_mul_f32": e[65], "v_pk_add_f32": e[66],
                                         "in place, op_sel_hi:[0,1] (dst pair == src0 pair)": e[68],
                                         "in place, op_sel:[1,0] op_sel_hi:[0,0]": e[69],
                                         "distinct dst, op_sel_hi:[0,1] (control)": e[70],
                                         "inline constant, op_sel_hi:[1,0]": e[71], "SGPR pair overwritten by v_cmp 2 instructions later": e[72],
                                         "packed multiply as first consumer of a global_load_dwordx4 (of launches * blocks * 256 * 400 * 2)": e[73]},
                          "mismatches_by_lane_group": {"0-15": sum(lanes[:16]), "16-31": sum(lanes[16:32]), "32-47": sum(lanes[32:48]),
                                                       "48-63": sum(lanes[48:])}}), flush=True)


if __name__ == "__main__":
    main()
