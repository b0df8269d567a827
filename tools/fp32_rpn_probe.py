"""Probe: where the fp32 drop-in call spends its time, and how fast torch/MIOpen runs the fp32 RPN block in three forms.
    python tools/fp32_rpn_probe.py            (prints JSON lines)"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from second_amd import synthetic as syn  # noqa: E402
from second_amd.models import RPNV2, RPNInference, fold_conv_bn_  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    rpn = RPNV2().eval().to(dev)
    x = torch.randn(8, 128, 200, 176, device=dev)
    with torch.no_grad():
        a = timeit(lambda: rpn(x))
        xcl = x.contiguous(memory_format=torch.channels_last)
        inf = RPNInference(rpn, torch.float32, backend="miopen").to(dev)
        b = timeit(lambda: inf(xcl))
        import copy
        r2 = copy.deepcopy(rpn)
        r2.blocks = torch.nn.ModuleList([fold_conv_bn_(blk) for blk in r2.blocks])
        r2.deblocks = torch.nn.ModuleList([fold_conv_bn_(blk) for blk in r2.deblocks])
        c = timeit(lambda: r2(x))
        w = rpn.blocks[0][4].weight
        d = timeit(lambda: F.conv2d(x, w, None, 1, 1))
        wcl = w.contiguous(memory_format=torch.channels_last)
        e = timeit(lambda: F.conv2d(xcl, wcl, None, 1, 1))
        xb, wb = x.bfloat16(), w.bfloat16()
        f = timeit(lambda: F.conv2d(xb, wb, None, 1, 1))
    print(json.dumps({"rpn_ms": {"modules_nchw_unfolded": round(a, 3), "folded_channels_last": round(b, 3), "folded_nchw": round(c, 3)},
                      "one_conv3x3_128_ms": {"fp32_nchw": round(d, 3), "fp32_nhwc": round(e, 3), "bf16_nchw": round(f, 3)}}), flush=True)


if __name__ == "__main__":
    main()
