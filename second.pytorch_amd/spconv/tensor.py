"""spconv.SparseConvTensor (spconv/__init__.py upstream; constructed at
second/pytorch/models/middle.py:199-200)."""
import numpy as np
import torch

from second_amd import ops as _ops


class Rulebook:
    """Cached result of a rulebook build, shared between layers through ``indice_key``.
    Holds the gather tables our kernels consume and (lazily) spconv's pair lists."""

    def __init__(self, out_indices, in_indices, nbr_out, nbr_in, num_out, in_shape, out_shape, subm,
                 pairs=None, pair_num=None, num_out_dev=None):
        self.num_out_dev = num_out_dev  # static-capacity mode: device count of live output rows
        self.out_indices, self.in_indices = out_indices, in_indices
        self.nbr_out, self.nbr_in = nbr_out, nbr_in
        self.num_out, self.in_shape, self.out_shape, self.subm = num_out, in_shape, out_shape, subm
        self.pairs, self.pair_num = pairs, pair_num

    def as_spconv_tuple(self):
        """(outids, indices, indice_pairs, indice_pair_num, spatial_shape) like spconv's indice_dict entry."""
        return (self.out_indices, self.in_indices, self.pairs, self.pair_num, self.in_shape)


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, num_active_dev=None):
        """features [N,C]; indices [N,4] int32 (batch, z, y, x); spatial_shape (z,y,x).

        ``num_active_dev`` (extension, device int32[>=1]): static-capacity mode -- the tensors are sized for
        a capacity, only the first num_active_dev[0] rows are live; every layer then runs without host
        synchronisation (hipGraph-capturable).  None = the reference's dynamic-shape behaviour."""
        self.num_active_dev = num_active_dev
        self.overflow_checks = []  # [(device int32[2] = (clamped, raw), capacity)] of strided layers upstream
        self.planned = None   # {id(conv): (Rulebook, event or None)} from SparseSequential.plan_chain
        self.features = features
        self.indices = indices
        if self.indices.dtype != torch.int32:
            self.indices = self.indices.int()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid  # accepted for API parity; the hash-table rulebook never needs the dense grid

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def dense(self, channels_first=True):
        if torch.is_grad_enabled() and self.features.requires_grad:   # training: differentiable scatter
            from .functional import SparseToDenseFunction
            out = SparseToDenseFunction.apply(self.features, self.indices.contiguous(), self.batch_size, self.spatial_shape,
                                              self.num_active_dev)
        else:
            out = _ops.sparse_to_dense(self.features, self.indices.contiguous(), self.batch_size, self.spatial_shape,
                                       num_dev=self.num_active_dev)
        if not channels_first:
            return out.permute(0, 2, 3, 4, 1).contiguous()
        return out

    def dense_channels_last_2d(self):
        """[B, C*D, H, W] in channels_last memory format == dense().view(B, C*D, H, W) values
        (the RPN input of second/pytorch/models/middle.py:206-210) without the permute copy."""
        if torch.is_grad_enabled() and self.features.requires_grad:   # training: differentiable, gradient gathered from its own strides
            from .functional import SparseToDenseFunction
            return SparseToDenseFunction.apply(self.features, self.indices.contiguous(), self.batch_size, self.spatial_shape,
                                               self.num_active_dev, True)
        return _ops.sparse_to_dense(self.features, self.indices.contiguous(), self.batch_size, self.spatial_shape,
                                    channels_last_2d=True, num_dev=self.num_active_dev)

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size
