#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy table of the kernels of one HIP source (hipcc -S, no GPU needed).

    python tools/kernel_resources.py second.pytorch_amd/csrc/indice_conv.hip [name-filter] [-- extra hipcc flags]

Used after every kernel change: a non-zero ScratchSize (spilled or dynamically indexed registers) or an occupancy
drop is visible here before any GPU time is spent.
"""
import os
import re
import subprocess
import sys
import tempfile


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    src = args[0]
    flt = args[1] if len(args) > 1 else ""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-S",
                               "--cuda-device-only", "-Wno-unused-command-line-argument", *extra, "-o", out, src])
        txt = open(out).read()
    names = re.findall(r"^(_Z\w+):\s*; @", txt, re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scratch':>8} {'lds':>7} {'occ':>4}  kernel")
    for name, pretty in zip(names, dem):
        if flt and flt not in pretty:
            continue
        start = txt.index(name + ":")
        blk = txt[start:]
        end = blk.find("; -- End function")
        blk = blk[:end + 3000]

        def g(k):
            m = re.search(r"; " + k + r":\s*(\S+)", blk)
            return m.group(1) if m else "?"
        short = re.sub(r"\(.*", "", pretty).replace("void sec::", "")
        print(f"{g('NumVgprs'):>5} {g('NumAgprs'):>5} {g('NumSgprs'):>5} {g('ScratchSize'):>8} {g('LDSByteSize'):>7} {g('Occupancy'):>4}  {short}")


if __name__ == "__main__":
    main()
