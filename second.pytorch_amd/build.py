"""Build libsecond_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

    python second.pytorch_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The library lands in second.pytorch_amd/lib/ (git-ignored, but it
travels to the GPU box with the gpurun snapshot).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in
       ("common.hip", "voxelize.hip", "rulebook.hip", "indice_conv.hip", "scatter.hip", "nms.hip", "dense.hip", "pillars.hip", "predict.hip")]
HDR = [os.path.join(HERE, "csrc", "common.hpp"), os.path.join(HERE, "..", "include", "second_hip.h")]
OUT = os.path.join(HERE, "lib", "libsecond_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def build(force=False, verbose=True):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT):
        newest = max(os.path.getmtime(p) for p in SRC + HDR)
        if os.path.getmtime(OUT) >= newest:
            return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("SEC_EXTRA_HIPCC_FLAGS", "").split()   # e.g. -DSEC_CONV_ABLATIONS for profiling builds
    # one translation unit per source, compiled concurrently (indice_conv.hip / dense.hip dominate), then one link
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc, *cflags, *extra, "-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SRC), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SRC))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-o", OUT, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
