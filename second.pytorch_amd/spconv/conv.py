"""spconv.SubMConv3d / SparseConv3d (spconv/conv.py upstream; used at
second/pytorch/models/middle.py:146-189 and resnet.py:10-29).  Leaf nn.Modules with Parameters
``weight`` [kD,kH,kW,Cin,Cout] and ``bias`` [Cout] so .tckpt checkpoints interchange."""
import math

import torch
from torch import nn
from torch.nn import init

from second_amd import ops as _ops
from . import functional as Fsp
from .modules import SparseModule
from .tensor import Rulebook, SparseConvTensor


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3, v
        return [int(x) for x in v]
    return [int(v)] * 3


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None):
        super().__init__()
        assert ndim == 3, "only 3-D sparse convolutions are on the SECOND hot path"
        assert groups == 1 and not transposed and not inverse, "unused by the reference (SURVEY 2.2)"
        self.ndim, self.in_channels, self.out_channels = ndim, in_channels, out_channels
        self.kernel_size, self.stride = _triple(kernel_size), _triple(stride)
        self.padding, self.dilation = _triple(padding), _triple(dilation)
        self.conv1x1 = all(k == 1 for k in self.kernel_size)
        self.subm, self.indice_key = subm, indice_key
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self._packed = None
        self._packed_key = None
        # static-capacity mode (see SparseConvTensor.num_active_dev): rows reserved for this layer's output.
        # Default = growth x input capacity; `static_out_rows` (e.g. from SecondDetector.calibrate) overrides.
        self.static_growth = 2.0
        self.static_out_rows = None
        self.last_num_out = None
        self.reset_parameters()

    def reset_parameters(self):
        fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        bound = math.sqrt(6.0 / ((1 + 5) * fan_in))  # kaiming_uniform_(a=sqrt(5))
        init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            b = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -b, b)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, subm={self.subm}, indice_key={self.indice_key}")

    # -- rulebook ---------------------------------------------------------------------------------
    def _rulebook(self, x):
        planned = getattr(x, "planned", None)
        if planned is not None and id(self) in planned:      # built ahead of time by the fused chain (SparseSequential.plan_chain)
            rb, event = planned[id(self)]
            if event is not None:
                torch.cuda.current_stream().wait_event(event)
            return rb
        rb = x.find_indice_pair(self.indice_key)
        if rb is not None and self.subm:
            return rb
        indices = x.indices.contiguous()
        nd = x.num_active_dev
        if self.subm:
            # sites produced by a strided conv: its output hash table is still around -> no re-hash (one use only)
            tbl = x.__dict__.pop("site_table", None)
            if tbl is not None and tbl[0] != (indices.data_ptr(), indices.shape[0]):
                tbl = None
            r = _ops.rulebook_subm(indices, x.batch_size, x.spatial_shape, self.kernel_size, self.dilation, n_dev=nd,
                                   site_table=tbl[1] if tbl else None)
        elif nd is not None:  # static capacity: no host sync, overflow is reported through num_out_dev[1]
            cap = self.static_out_rows or int(indices.shape[0] * self.static_growth)
            hint = max(1, -(-cap // max(indices.shape[0], 1)))
            r = _ops.rulebook_conv(indices, x.batch_size, x.spatial_shape, self.kernel_size, self.stride, self.padding,
                                   self.dilation, n_dev=nd, out_cap=cap, out_per_in_hint=hint,
                                   want_nbr_in=torch.is_grad_enabled(), in_sites=self._in_sites(x, indices))
        else:
            r = _ops.rulebook_conv(indices, x.batch_size, x.spatial_shape, self.kernel_size, self.stride, self.padding,
                                   self.dilation, want_nbr_in=torch.is_grad_enabled(), in_sites=self._in_sites(x, indices))
        rb = Rulebook(r["out_indices"], indices, r["nbr_out"], r["nbr_in"], r["num_out"], x.spatial_shape,
                      r["out_shape"], self.subm, num_out_dev=r["num_out_dev"])
        if r.get("site_table") is not None:
            rb._site_table = ((r["out_indices"].data_ptr(), r["out_indices"].shape[0]), r["site_table"])
            if isinstance(r["site_table"][0], str):      # its bitmap also serves the NEXT strided build (kept until then)
                rb._site_bitmap = rb._site_table
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = rb
        if nd is None:
            self.last_num_out = rb.num_out
        return rb

    @staticmethod
    def _in_sites(x, indices):
        """The sorted-numbering site table of the strided layer that produced these sites (carried through the SubM layers
        of the stage), if it still describes exactly this index tensor."""
        tbl = getattr(x, "site_bitmap", None)
        if tbl is not None and tbl[0] == (indices.data_ptr(), indices.shape[0]):
            return tbl[1]
        return None

    def packed_weight(self):
        w = self.weight
        key = (w._version, w.dtype, w.device, w.data_ptr(), _ops.get_fp32_mode() if (w.is_cuda and w.dtype == torch.float32) else None)
        if self._packed_key != key:
            self._packed = _ops.pack_weight(w.detach().contiguous()) if w.is_cuda else None
            self._packed_key = key
        return self._packed

    def _wrap(self, x, feats, rb):
        out = SparseConvTensor(feats, rb.out_indices, rb.out_shape, x.batch_size, x.grid, rb.num_out_dev)
        out.planned = getattr(x, "planned", None)
        out.overflow_checks = getattr(x, "overflow_checks", [])
        if rb.num_out_dev is not None and not rb.subm:
            out.overflow_checks = out.overflow_checks + [(rb.num_out_dev, rb.out_indices.shape[0])]
        out.indice_dict = x.indice_dict
        if not rb.subm:
            out.site_table = rb.__dict__.pop("_site_table", None)   # handed to the next SubM rulebook build, then dropped
            out.site_bitmap = rb.__dict__.pop("_site_bitmap", None)
            out.site_map_tensor = rb.__dict__.get("_site_map")      # fused chain build: the BEV site map of the last level
        else:
            out.site_bitmap = getattr(x, "site_bitmap", None)       # same sites: still valid for the next strided build
        return out

    # -- forward ----------------------------------------------------------------------------------
    def forward(self, x):
        assert isinstance(x, SparseConvTensor)
        if self.conv1x1 and not self.subm and all(s == 1 for s in self.stride) and all(p == 0 for p in self.padding):
            feats = torch.mm(x.features, self.weight.view(self.in_channels, self.out_channels))
            if self.bias is not None:
                feats = feats + self.bias
            out = SparseConvTensor(feats, x.indices, x.spatial_shape, x.batch_size, x.grid)
            out.indice_dict = x.indice_dict
            return out
        rb = self._rulebook(x)
        if self.weight.dtype == x.features.dtype:
            w, packed = self.weight, self.packed_weight()
        else:
            # mixed precision (fp32 master weights, 16-bit features): IndiceConvFunction casts + packs and returns dW in fp32
            w, packed = self.weight, None
        feats = Fsp.indice_conv(x.features, w, rb, packed)
        if self.bias is not None:
            feats = feats + self.bias.to(feats.dtype)
        return self._wrap(x, feats, rb)

    def forward_fused(self, x, scale, shift, relu):
        """Inference: conv + per-channel scale/shift (folded BatchNorm1d, bias) + ReLU in ONE launch."""
        rb = self._rulebook(x)
        w = self.weight.detach()
        packed = self.packed_weight()
        if w.dtype != x.features.dtype:
            w, packed = w.to(x.features.dtype), None
        feats = _ops.indice_conv(x.features.contiguous(), w.contiguous(), rb.nbr_out, rb.num_out, packed=packed,
                                 scale=scale, shift=shift, relu=relu, num_out_dev=rb.num_out_dev)
        return self._wrap(x, feats, rb)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)
