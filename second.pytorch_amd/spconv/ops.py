"""spconv.ops (spconv/ops.py upstream): pair-list style functional API kept for parity.
The modules in conv.py use the gather-table form directly; these wrappers serve external callers."""
import torch

from second_amd import ops as _ops


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    ndim = len(input_size)
    out = []
    for i in range(ndim):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
        out.append(1 if kernel_size[i] == -1 else size)
    return out


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1, out_padding=0,
                     subm=False, transpose=False, grid=None):
    """-> (outids [M,4], indice_pairs [K,2,N], indice_pair_num [K]) in spconv's canonical CPU order."""
    assert not transpose and out_padding in (0, [0, 0, 0], (0, 0, 0)), "transposed conv is unused by the reference"
    indices = indices.int().contiguous()
    if subm:
        r = _ops.rulebook_subm(indices, batch_size, spatial_shape, ksize, dilation, want_pairs=True)
    else:
        r = _ops.rulebook_conv(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, want_pairs=True)
    return r["out_indices"], r["pairs"], r["pair_num"]


def _table_from_pairs(indice_pairs, indice_pair_num, num_out, inverse=False):
    k, _, n = indice_pairs.shape
    src, dst = (1, 0) if inverse else (0, 1)
    nbr = torch.full((num_out, k), -1, dtype=torch.int32, device=indice_pairs.device)
    ar = torch.arange(n, device=indice_pairs.device).unsqueeze(0)
    valid = ar < indice_pair_num.unsqueeze(1).to(ar.dtype)
    kk = torch.arange(k, device=indice_pairs.device).unsqueeze(1).expand(k, n)
    nbr[indice_pairs[:, dst][valid].long(), kk[valid]] = indice_pairs[:, src][valid]
    return nbr


def indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse=False, subm=False):
    nbr = _table_from_pairs(indice_pairs, indice_pair_num, int(num_activate_out), inverse)
    return _ops.indice_conv(features.contiguous(), filters.contiguous(), nbr, int(num_activate_out),
                            packed=_ops.pack_weight(filters.contiguous()))


def indice_conv_backward(features, filters, out_bp, indice_pairs, indice_pair_num, inverse=False, subm=False):
    n_out = out_bp.shape[0]
    nbr_out = _table_from_pairs(indice_pairs, indice_pair_num, n_out, inverse)
    nbr_in = _table_from_pairs(indice_pairs, indice_pair_num, features.shape[0], not inverse)
    return _ops.indice_conv_backward(features.contiguous(), filters.contiguous(), nbr_out, nbr_in, out_bp.contiguous())


def nms(boxes, scores, pre_max_size, post_max_size, thresh, eps):
    """torch.ops.spconv.nms stand-in (only referenced by the dead nms_v2, box_torch_ops.py:479-489)."""
    from .utils import _nms_tensor
    return _nms_tensor(boxes, scores, pre_max_size, post_max_size, thresh, eps)
