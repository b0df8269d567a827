"""Drop-in replacement for the `spconv` (traveller59/spconv v1.x) Python package, backed by
libsecond_hip.so on MI355X.  Same public names, argument meaning and tensor layouts as the package the
reference imports (second/pytorch/models/middle.py:4, second/builder/voxel_builder.py:3,
second/pytorch/core/box_torch_ops.py:13, second/core/non_max_suppression/nms_cpu.py:5-6, nms_gpu.py:8,
second/core/box_np_ops.py:5), so second/pytorch/train.py runs unmodified with this directory first on
sys.path.  GPU only: ops on CPU tensors raise (no fallback).
"""
from .tensor import SparseConvTensor, Rulebook
from .modules import SparseModule, SparseSequential
from .conv import SparseConvolution, SubMConv3d, SparseConv3d
from . import functional, ops, utils

__version__ = "1.1+second.pytorch_amd"
__all__ = ["SparseConvTensor", "Rulebook", "SparseModule", "SparseSequential", "SparseConvolution",
           "SubMConv3d", "SparseConv3d", "functional", "ops", "utils"]
