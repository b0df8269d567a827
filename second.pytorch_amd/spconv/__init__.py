"""Drop-in replacement for the `spconv` (traveller59/spconv v1.x) Python package, backed by
libsecond_hip.so on MI355X.  Same public names, argument meaning and tensor layouts as the package the
reference imports (second/pytorch/models/middle.py:4, second/builder/voxel_builder.py:3,
second/pytorch/core/box_torch_ops.py:13, second/core/non_max_suppression/nms_cpu.py:5-6, nms_gpu.py:8,
second/core/box_np_ops.py:5), so second/pytorch/train.py runs unmodified with this directory first on
sys.path.  GPU only: ops on CPU tensors raise (no fallback).
"""
from second_amd.compat import spawn_dataloader_workers as _spawn_dataloader_workers
from second_amd import runtime as _runtime  # noqa: F401  (registers the fork guard)

# Zero-edit path: the reference forks its DataLoader workers (second/pytorch/train.py:262-277) and the workers call
# spconv.utils.VoxelGeneratorV2.generate, which needs a HIP context here.  Importing spconv is all the reference does, so the
# import itself switches THE REFERENCE'S loaders (dataset or collate function from its `second` package) to spawned workers;
# every other DataLoader of the host program keeps its start method (idempotent; an explicit multiprocessing_context wins;
# SEC_KEEP_FORK=1 opts out; second_amd.compat.install() switches all loaders).
import os as _os
if _os.environ.get("SEC_KEEP_FORK", "0") != "1":
    _spawn_dataloader_workers("reference")

# SEC_ACCELERATE_MODEL=1: the fused static-capacity pipeline behind the reference's own VoxelNet.forward(example), without a call
# (second_amd.dropin: the class is wrapped when second.pytorch.models.voxelnet is imported; eval mode only, networks outside the
# fused path keep their forward).  Off by default: `compat.accelerate_model(net)` / `second_amd.launch evaluate` are the explicit routes.
if _os.environ.get("SEC_ACCELERATE_MODEL", "0") == "1":
    from second_amd.dropin import install_import_hook as _install_import_hook
    _install_import_hook()

from .tensor import SparseConvTensor, Rulebook
from .modules import SparseModule, SparseSequential
from .conv import SparseConvolution, SubMConv3d, SparseConv3d
from . import functional, ops, utils

__version__ = "1.1+second.pytorch_amd"
__all__ = ["SparseConvTensor", "Rulebook", "SparseModule", "SparseSequential", "SparseConvolution",
           "SubMConv3d", "SparseConv3d", "functional", "ops", "utils"]
