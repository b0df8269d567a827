"""TEST-ONLY backend: routes ``second_amd.ops`` to the CPU oracle so that the *host logic* of the drop-in
``spconv`` package and of ``second_amd.models`` (module classes, rulebook caching by indice_key, BN folding,
state-dict compatibility, predict post-processing) can be exercised in the CPU-only build container against
the unmodified reference.  The product never imports this file; on a GPU box the HIP library is the backend.

    with oracle_backend.installed(): ...
"""
import contextlib

import numpy as np
import torch

from oracle import oracle as orc


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _tables(pairs, pair_num, n_in, n_out):
    k = pairs.shape[0]
    nbr_out = -np.ones((n_out, k), np.int32)
    nbr_in = -np.ones((n_in, k), np.int32)
    for kk in range(k):
        i, o = pairs[kk, 0, :pair_num[kk]], pairs[kk, 1, :pair_num[kk]]
        nbr_out[o, kk] = i
        nbr_in[i, kk] = o
    return nbr_out, nbr_in


def _pairs_from_nbr(nbr_out, n_in):
    nbr_out = _np(nbr_out)
    n_out, k = nbr_out.shape
    pairs = -np.ones((k, 2, n_in), np.int32)
    num = np.zeros((k,), np.int32)
    for kk in range(k):
        o = np.nonzero(nbr_out[:, kk] >= 0)[0]
        i = nbr_out[o, kk]
        order = np.argsort(i, kind="stable")
        num[kk] = len(o)
        pairs[kk, 0, :len(o)] = i[order]
        pairs[kk, 1, :len(o)] = o[order]
    return pairs, num


def voxelize(points, point_offsets, point_cloud_range, voxel_size, max_points, max_voxels, cap_mode="break",
             mean_features=0, sync=True, mean_dtype=None, fill=True):
    pts, offs = _np(points), _np(point_offsets)
    outs = {"voxels": [], "coordinates": [], "num_points_per_voxel": []}
    voff = [0]
    for b in range(len(offs) - 1):
        r = orc.points_to_voxel(pts[offs[b]:offs[b + 1]], voxel_size, point_cloud_range, max_points, max_voxels, cap_mode)
        outs["voxels"].append(r["voxels"])
        outs["coordinates"].append(np.concatenate([np.full((r["voxel_num"], 1), b, np.int32), r["coordinates"]], 1))
        outs["num_points_per_voxel"].append(r["num_points_per_voxel"])
        voff.append(voff[-1] + r["voxel_num"])
    res = {k: torch.from_numpy(np.concatenate(v)) for k, v in outs.items()}
    res["voxel_offsets"] = torch.tensor(voff, dtype=torch.int32)
    res["voxel_num"] = voff[-1]
    if mean_features:
        res["mean"] = torch.from_numpy(orc.simple_voxel_mean(_np(res["voxels"]), _np(res["num_points_per_voxel"]), mean_features))
        if mean_dtype is not None:
            res["mean"] = res["mean"].to(mean_dtype)
    return res


def voxel_block_filter(vox, grid_size_xy, block_factor, block_size, height_threshold, height_high_threshold=3.0,
                       sync=True):
    voff = _np(vox["voxel_offsets"])
    outs = {"voxels": [], "coordinates": [], "num_points_per_voxel": []}
    ooff = [0]
    for b in range(len(voff) - 1):
        sl = slice(int(voff[b]), int(voff[b + 1]))
        v, c, n = _np(vox["voxels"])[sl], _np(vox["coordinates"])[sl], _np(vox["num_points_per_voxel"])[sl]
        keep = orc.block_filter(v, c[:, 1:], n, grid_size_xy, block_factor, block_size, height_threshold, height_high_threshold)
        outs["voxels"].append(v[keep]); outs["coordinates"].append(c[keep]); outs["num_points_per_voxel"].append(n[keep])
        ooff.append(ooff[-1] + int(keep.sum()))
    res = {k: torch.from_numpy(np.concatenate(v)) for k, v in outs.items()}
    res["voxel_offsets"] = torch.tensor(ooff, dtype=torch.int32)
    res["voxel_num"] = ooff[-1]
    return res


def rulebook_subm(indices, batch_size, spatial_shape, ksize=3, dilation=1, want_pairs=False, n_dev=None, **kw):
    idx = _np(indices)
    _, pairs, num = orc.rulebook_subm(idx, batch_size, spatial_shape, ksize, dilation)
    nbr_out, _ = _tables(pairs, num, len(idx), len(idx))
    return {"nbr_out": torch.from_numpy(nbr_out), "nbr_in": None, "pairs": torch.from_numpy(pairs),
            "pair_num": torch.from_numpy(num), "out_indices": indices, "num_out": len(idx), "num_out_dev": None,
            "out_shape": [int(s) for s in spatial_shape]}


def rulebook_conv(indices, batch_size, spatial_shape, ksize, stride, padding, dilation=1, want_pairs=False, **kw):
    idx = _np(indices)
    out_idx, pairs, num, out_shape = orc.rulebook_conv(idx, batch_size, spatial_shape, ksize, stride, padding, dilation)
    nbr_out, nbr_in = _tables(pairs, num, len(idx), len(out_idx))
    return {"nbr_out": torch.from_numpy(nbr_out), "nbr_in": torch.from_numpy(nbr_in), "pairs": torch.from_numpy(pairs),
            "pair_num": torch.from_numpy(num), "out_indices": torch.from_numpy(out_idx), "num_out": len(out_idx),
            "num_out_dev": None, "out_shape": [int(s) for s in out_shape]}


def pack_weight(weight):
    return None


def indice_conv(features, weight, nbr_out, num_out, packed=None, scale=None, shift=None, relu=False, out_dtype=None,
                num_out_dev=None):
    pairs, num = _pairs_from_nbr(nbr_out, features.shape[0])
    y = orc.indice_conv(_np(features.float()), _np(weight.float()), pairs, num, int(num_out), acc64=True)
    if scale is not None:
        y = y * _np(scale)
    if shift is not None:
        y = y + _np(shift)
    if relu:
        y = np.maximum(y, 0)
    return torch.from_numpy(y.astype(np.float32)).to(out_dtype or features.dtype)


def indice_conv_backward(features, weight, nbr_out, nbr_in, dout, need_dfeat=True, need_dweight=True, dweight_dtype=None, packed_dgrad=None,
                         dweight_out=None):
    pairs, num = _pairs_from_nbr(nbr_out, features.shape[0])
    dfeat, dw = orc.indice_conv_backward(_np(features.float()), _np(weight.float()), pairs, num, _np(dout.float()))
    return torch.from_numpy(dfeat).to(features.dtype), torch.from_numpy(dw).to(dweight_dtype or weight.dtype)


def sparse_to_dense(features, indices, batch_size, spatial_shape, channels_last_2d=False, num_dev=None):
    d = torch.from_numpy(orc.sparse_to_dense(_np(features.float()), _np(indices), batch_size, spatial_shape)).to(features.dtype)
    if channels_last_2d:
        b, c, dd, h, w = d.shape
        return d.view(b, c * dd, h, w).contiguous(memory_format=torch.channels_last)
    return d


def pillar_scatter(features, coords, batch_size, ny, nx, channels_last=False, num_dev=None):
    return torch.from_numpy(orc.pillar_scatter(_np(features.float()), _np(coords), batch_size, ny, nx)).to(features.dtype)


def dense_to_sparse(dense, indices, num_dev=None, depth=0):
    idx = indices.long()
    if dense.dim() == 5:
        return dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]].contiguous()
    if depth:                                     # [B, C * D, H, W], channel = c * D + z
        b, cd, h, w = dense.shape
        return dense.reshape(b, cd // depth, depth, h, w)[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]].contiguous()
    return dense[idx[:, 0], :, idx[:, 2], idx[:, 3]].contiguous()


def rotate_iou(boxes, qboxes, criterion=-1):
    return torch.from_numpy(orc.rotate_iou(_np(boxes), _np(qboxes), criterion))


def nms_sorted(dets, counts, thresh, kind="rotate", semantics="numba", eps=1.0, post_max=0):
    d, c = _np(dets), _np(counts)
    b, max_n, _ = d.shape
    keep = np.zeros((b, max_n), np.int32)
    num = np.zeros((b,), np.int32)
    for i in range(b):
        if kind == "rotate":
            k = orc.rotate_nms_sorted(d[i, :c[i]], thresh, semantics)
        else:
            k = orc.nms_sorted(d[i, :c[i]], thresh, semantics, eps)
        if post_max:
            k = k[:post_max]
        keep[i, :len(k)] = k
        num[i] = len(k)
    return torch.from_numpy(keep), torch.from_numpy(num)


_NAMES = ["voxelize", "rulebook_subm", "rulebook_conv", "pack_weight", "indice_conv", "indice_conv_backward",
          "sparse_to_dense", "pillar_scatter", "rotate_iou", "nms_sorted", "voxel_block_filter", "dense_to_sparse"]


@contextlib.contextmanager
def installed():
    from second_amd import ops
    import spconv.utils as su
    saved = {n: getattr(ops, n) for n in _NAMES}
    saved_dev = su._dev
    try:
        for n in _NAMES:
            setattr(ops, n, globals()[n])
        su._dev = lambda: torch.device("cpu")
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
        su._dev = saved_dev
