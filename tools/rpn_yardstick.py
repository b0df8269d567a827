#!/usr/bin/env python3
"""Yardsticks and a power / clock trace for the RPN 3x3 128->128 conv on 8 x 200 x 176 (83.05 GFLOP per launch, bf16), so that
"what this chip sustains on this shape" is evidence rather than inference:

  * sec_conv2d_nhwc           the hand-written kernel (k_conv2d_halo_reg)
  * hipBLASLt (torch.mm)      the equivalent plain GEMM  [281 600 x 1152] x [1152 x 128]  -- the im2col product WITHOUT any of the
                              conv's halo reuse, i.e. an upper-bound yardstick for an implicit-GEMM formulation of this shape
  * hipBLASLt, square         4096^3 bf16, the library's comfortable case on this chip
  * MIOpen (torch conv2d)     the same convolution, channels_last bf16, + our fused bias / ReLU pass

each run for ~1.5 s back to back while a sampler thread reads socket power and shader clock through amdsmi (rocm-smi fallback);
random (post-ReLU-like) data -- zero-filled inputs clock ~19 % higher (MI355X_MICROARCH.md, DVFS give-back)."""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "second.pytorch_amd"))
import torch  # noqa: E402
from second_amd import ops  # noqa: E402


class Sampler(threading.Thread):
    def __init__(self, period=0.05):
        super().__init__(daemon=True)
        self.period, self.samples, self.stop_flag, self.label = period, [], False, "idle"
        self.h = None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self.amdsmi, self.h = amdsmi, amdsmi.amdsmi_get_processor_handles()[0]
        except Exception as e:  # noqa: BLE001
            self.err = repr(e)

    def read(self):
        if self.h is not None:
            a = self.amdsmi
            try:
                m = a.amdsmi_get_gpu_metrics_info(self.h)
                def pick(*keys):
                    for k in keys:
                        v = m.get(k)
                        if v not in (None, "N/A", 0, 65535):
                            return v
                    return None
                return {"power_w": pick("current_socket_power", "average_socket_power"), "raw_keys": sorted(m)[:80] if not self.samples else None,
                        "gfxclk_mhz": m.get("current_gfxclk") or m.get("average_gfxclk_frequency"),
                        "gfxclks_mhz": (m.get("current_gfxclks") or [])[:8], "temp_c": m.get("temperature_hotspot")}
            except Exception:  # noqa: BLE001
                try:
                    p = a.amdsmi_get_power_info(self.h)
                    c = a.amdsmi_get_clock_info(self.h, a.AmdSmiClkType.GFX)
                    return {"power_w": p.get("average_socket_power") or p.get("current_socket_power"), "gfxclk_mhz": c.get("clk") or c.get("cur_clk")}
                except Exception as e:  # noqa: BLE001
                    return {"error": repr(e)}
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            return {"rocm_smi": json.loads(r.stdout)}
        except Exception as e:  # noqa: BLE001
            return {"error": repr(e)}

    def run(self):
        while not self.stop_flag:
            s = self.read()
            s["t"], s["label"] = time.time(), self.label
            self.samples.append(s)
            time.sleep(self.period)


def loop(fn, seconds, flop, sampler, label):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    sampler.label = label
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t0 = time.time()
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(100):
            fn()
        n += 100
        torch.cuda.synchronize()       # bounds the queue; 100 launches of >= 50 us hide the sync
    e1.record()
    torch.cuda.synchronize()
    sampler.label = "idle"
    us = e0.elapsed_time(e1) * 1e3 / n
    def num(v):
        try:
            return float(v)
        except (TypeError, ValueError):
            return None
    ss = [s for s in sampler.samples if s["label"] == label]
    pw = [v for v in (num(s.get("power_w")) for s in ss) if v]
    ck = [v for v in (num(s.get("gfxclk_mhz")) for s in ss) if v]
    if not ck:      # per-XCD clocks
        ck = [sum(c) / len(c) for c in ([v for v in map(num, s.get("gfxclks_mhz") or []) if v] for s in ss) if c]
    res = {"what": label, "us": round(us, 2), "tflops": round(flop / us / 1e6, 1), "frac_of_2.5PF": round(flop / us / 1e6 / 2500, 4), "launches": n,
           "power_w_mean": round(sum(pw) / len(pw), 1) if pw else None, "power_w_max": max(pw) if pw else None,
           "gfxclk_mhz_mean": round(sum(ck) / len(ck)) if ck else None, "gfxclk_mhz_min": min(ck) if ck else None, "samples": len(ss)}
    print(json.dumps(res), flush=True)
    time.sleep(0.5)
    return res


def main():
    secs = float(os.environ.get("SECONDS_PER_RUN", "1.5"))
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    x = torch.relu(torch.randn(8, 128, 200, 176, generator=g)).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(128, 128, 3, 3, generator=g) / 34).to(dev).bfloat16()
    b = torch.randn(128, generator=g).to(dev)
    pk = ops.conv2d_pack_weight(w)
    flop = 2.0 * 8 * 200 * 176 * 128 * 128 * 9
    sm = Sampler()
    sm.start()
    time.sleep(0.3)
    out = [loop(lambda: ops.conv2d_nhwc(x, pk, b, 128, 3, 1, 1, relu=True), secs, flop, sm, "sec_conv2d_nhwc " + "")]
    out[-1]["kernel"] = ops.last_kernel_name()
    a = torch.relu(torch.randn(8 * 200 * 176, 1152, generator=g)).to(dev).bfloat16()
    bm = (torch.randn(1152, 128, generator=g) / 34).to(dev).bfloat16()
    c = torch.empty(8 * 200 * 176, 128, device=dev, dtype=torch.bfloat16)
    out.append(loop(lambda: torch.mm(a, bm, out=c), secs, flop, sm, "hipBLASLt torch.mm [281600 x 1152] x [1152 x 128] (im2col-equivalent GEMM, no halo reuse)"))
    bt = bm.t().contiguous()
    out.append(loop(lambda: torch.mm(a, bt.t(), out=c), secs, flop, sm, "hipBLASLt torch.mm, B given as [128 x 1152]^T (NT layout)"))
    del a, c
    s1, s2 = torch.randn(4096, 4096, generator=g).to(dev).bfloat16(), torch.randn(4096, 4096, generator=g).to(dev).bfloat16()
    s3 = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)
    out.append(loop(lambda: torch.mm(s1, s2, out=s3), secs, 2.0 * 4096 ** 3, sm, "hipBLASLt torch.mm 4096^3 bf16 (library yardstick)"))
    wcl = w.contiguous(memory_format=torch.channels_last)
    out.append(loop(lambda: ops.bias_act_(torch.nn.functional.conv2d(x, wcl, None, 1, 1), b, True), secs, flop, sm,
                    "MIOpen conv2d (torch, channels_last bf16) + fused bias / ReLU pass"))
    sm.stop_flag = True
    idle = [s for s in sm.samples if s["label"] == "idle"]
    print(json.dumps({"idle_power_w": [s.get("power_w") for s in idle[:3]], "sampler": "amdsmi" if sm.h is not None else "rocm-smi",
                      "sampler_error": getattr(sm, "err", None), "first_sample": sm.samples[0] if sm.samples else None}))
    series = [{k: s.get(k) for k in ("t", "label", "power_w", "gfxclk_mhz")} for s in sm.samples]
    t0 = series[0]["t"] if series else 0
    for s in series:
        s["t"] = round(s["t"] - t0, 3)
    if os.environ.get("SERIES_OUT"):
        json.dump(series, open(os.environ["SERIES_OUT"], "w"))


if __name__ == "__main__":
    main()
