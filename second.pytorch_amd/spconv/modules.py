"""spconv.SparseModule / SparseSequential (spconv/modules.py upstream; second/pytorch/models/middle.py:145)."""
from collections import OrderedDict

import torch
from torch import nn


class SparseModule(nn.Module):
    """Marker base class: modules that take and return a SparseConvTensor."""


def _is_sparse(m):
    return isinstance(m, SparseModule)


def _sec_ops():
    from second_amd import ops
    return ops


class SparseSequential(SparseModule):
    """nn.Sequential that applies SparseModules to the tensor and plain modules to ``.features``.

    Inference peephole (API unchanged): ``SparseConvolution -> BatchNorm1d [-> ReLU]`` runs as ONE fused
    launch with the BatchNorm folded into a per-channel scale/shift epilogue (``fuse_inference``)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)
        self.fuse_inference = True
        self.fuse_train_bn = True   # training, 16-bit rows: fused BatchNorm1d + ReLU kernels (False: torch's BatchNorm1d on the rows)
        self._fold_cache = {}

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError(f"index {idx} is out of range")
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def plan_chain(self, x):
        """Static-capacity inference: ALL rulebooks of this sequence from ONE fused build (ops.rulebook_chain: 4 + (levels - 1)
        launches instead of ~25 dependent ones) when the stack is what SECOND builds -- 3x3x3 SubM layers between 3x3x3 stride-2 /
        (3,1,1) stride-(2,1,1) convs (middle.py:146-189) -- and the rulebook numbering in force is "sorted".  Returns the
        ``planned`` dict ({id(conv): (Rulebook, None)}) or None (then every layer builds its own rulebook as before)."""
        from .conv import SparseConvolution
        from .tensor import Rulebook
        ops = _sec_ops()
        if x.num_active_dev is None or torch.is_grad_enabled() or ops._numbering != "sorted" or not x.features.is_cuda:
            return None
        convs, want_subm, plan, cap = [], [False], [], x.indices.shape[0]
        for m in self._modules.values():
            if not isinstance(m, SparseConvolution) or m.conv1x1:
                continue
            if m.subm:
                if m.kernel_size != [3, 3, 3] or m.dilation != [1, 1, 1]:
                    return None
                want_subm[-1] = True
                plan.append((m, len(convs), True))
            else:
                if m.dilation != [1, 1, 1]:
                    return None
                cap = m.static_out_rows or int(cap * m.static_growth)
                convs.append((m.kernel_size, m.stride, m.padding, cap))
                want_subm.append(False)
                plan.append((m, len(convs), False))
        if not convs:
            return None
        tbl = getattr(x, "site_table", None)
        tbl = tbl[1] if (tbl is not None and tbl[0] == (x.indices.data_ptr(), x.indices.shape[0])) else None
        if not (tbl is not None and isinstance(tbl[0], str) and tbl[0] == "vox"):
            # level-0 rows that do not come straight from the voxeliser (block-filtered voxels of nuscenes/all.fhd): their SubM
            # layers hash the sites themselves, the levels above still come from the fused build
            tbl, want_subm[0] = None, False
            plan = [p for p in plan if not (p[2] and p[1] == 0)]
        r = ops.rulebook_chain(x.indices.contiguous(), x.batch_size, x.spatial_shape, convs, n_dev=x.num_active_dev, site_table=tbl,
                               want_subm=want_subm, want_site_map=int(x.spatial_shape[0]) > 0)
        if r is None:
            return None
        lv = r["levels"]
        planned = {}
        for m, level, subm in plan:
            L = lv[level]
            if subm:
                rb = Rulebook(L["indices"], L["indices"], L["subm_nbr"], None, L["cap"], L["shape"], L["shape"], True,
                              num_out_dev=L["num_dev"])
            else:
                I = lv[level - 1]
                rb = Rulebook(L["indices"], I["indices"], L["nbr_out"], None, L["cap"], I["shape"], L["shape"], False,
                              num_out_dev=L["num_dev"])
                if level == len(convs):
                    rb._site_map = r["site_map"]
            rb._chain_workspace = r["workspace"]
            planned[id(m)] = (rb, None)
        return planned

    def _folded(self, conv, bn):
        key = (id(conv), id(bn), bn.weight._version if bn.weight is not None else 0,
               bn.bias._version if bn.bias is not None else 0, bn.running_mean._version, bn.running_var._version,
               conv.bias._version if conv.bias is not None else -1, bn.running_mean.device)
        hit = self._fold_cache.get(id(conv))
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        with torch.no_grad():
            var = bn.running_var.float()
            scale = torch.rsqrt(var + bn.eps)
            if bn.weight is not None:
                scale = scale * bn.weight.float()
            shift = -bn.running_mean.float() * scale
            if bn.bias is not None:
                shift = shift + bn.bias.float()
            if conv.bias is not None:
                shift = shift + conv.bias.float() * scale
            scale, shift = scale.contiguous(), shift.contiguous()
        self._fold_cache[id(conv)] = (key, scale, shift)
        return scale, shift

    def forward(self, input):
        from .conv import SparseConvolution
        from .tensor import SparseConvTensor
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if (self.fuse_inference and not self.training and isinstance(m, SparseConvolution) and not m.conv1x1
                    and isinstance(input, SparseConvTensor) and input.features.is_cuda
                    and not torch.is_grad_enabled()
                    and i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d)
                    and mods[i + 1].track_running_stats and not mods[i + 1].training):
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                scale, shift = self._folded(m, mods[i + 1])
                input = m.forward_fused(input, scale, shift, relu)
                i += 3 if relu else 2
                continue
            if (self.fuse_train_bn and isinstance(m, nn.BatchNorm1d) and m.training and m.affine and m.track_running_stats
                    and isinstance(input, SparseConvTensor) and input.indices.shape[0] != 0 and input.features.is_cuda
                    and _sec_ops().bn_train_supported(m.num_features, input.features.dtype)):
                # training with 16-bit features: BatchNorm1d (batch statistics) + ReLU in two passes + a finalize on the rows
                # (sec_bn_relu_fwd_nhwc / _bwd_nhwc) instead of torch's statistics / transform / ReLU kernels and their backward
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                mom = m.momentum if m.momentum is not None else 0.1
                input.features = _sec_ops().BatchNormReluFunction.apply(input.features.contiguous(), m.weight, m.bias, m.running_mean,
                                                                       m.running_var, m.eps, mom, relu, input.num_active_dev)
                _sec_ops().bump_bn_counter(m)
                i += 2 if relu else 1
                continue
            if _is_sparse(m):
                input = m(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    if input.num_active_dev is not None and isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training:
                        raise RuntimeError("static-capacity sparse tensor through a torch BatchNorm in training mode: its statistics "
                                           "would include the rows past the live count (the fused BatchNorm1d path needs 16-bit "
                                           "features and a supported channel count)")
                    input.features = m(input.features)
            else:
                input = m(input)
            i += 1
        return input
