"""second_amd -- host side of the MI355X-native SECOND hot path.

The arithmetic lives in ``lib/libsecond_hip.so`` (hand-written HIP for gfx950, C ABI declared in
``include/second_hip.h``).  PyTorch is used only as plumbing: device memory, streams, RCCL.
There is NO CPU fallback: importing :mod:`second_amd.runtime` without the built library, or calling an
op on a non-GPU tensor, raises.
"""
__version__ = "0.1.0"
