"""ctypes front-end of the CPU oracle (``oracle/second_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py``.  Nothing under
``second.pytorch_amd/`` may import this module.

numpy in / numpy out; every function cites the oracle C function, which in
turn cites the reference file:line it restates.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsecond_oracle.so")
_lib = None

_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile the oracle with gcc (Makefile next to this file)."""
    src = os.path.join(_HERE, "second_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_rotate_iou_pair.restype = ctypes.c_float
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _ip(a):
    return a.ctypes.data_as(_i32p)


def grid_size(point_cloud_range, voxel_size):
    rng, vs = _f(point_cloud_range), _f(voxel_size)
    g = np.zeros(3, np.int32)
    lib().orc_grid_size(_fp(rng), _fp(vs), _ip(g))
    return g.astype(np.int64)


def points_to_voxel(points, voxel_size, point_cloud_range, max_points, max_voxels, cap_mode="break"):
    """orc_points_to_voxel.  Returns dict like spconv's VoxelGeneratorV2.generate."""
    points = _f(points)
    n, nf = points.shape
    vs, rng = _f(voxel_size), _f(point_cloud_range)
    voxels = np.zeros((max_voxels, max_points, nf), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    npv = np.zeros((max_voxels,), np.int32)
    mode = {"break": 0, "continue": 1}[cap_mode]
    num = lib().orc_points_to_voxel(_fp(points), n, nf, _fp(vs), _fp(rng), int(max_points),
                                    int(max_voxels), mode, _fp(voxels), _ip(coors), _ip(npv))
    if num < 0:
        raise MemoryError("oracle voxel lookup grid allocation failed")
    return {"voxels": voxels[:num], "coordinates": coors[:num], "num_points_per_voxel": npv[:num],
            "voxel_num": num}


def simple_voxel_mean(voxels, num_points, nf):
    voxels, num_points = _f(voxels), _i(num_points)
    v, t, f = voxels.shape
    out = np.zeros((v, nf), np.float32)
    lib().orc_simple_voxel_mean(_fp(voxels), _ip(num_points), v, t, f, nf, _fp(out))
    return out


def _triple(x):
    if np.isscalar(x):
        return _i([x, x, x])
    x = _i(list(x))
    assert x.shape == (3,)
    return x


def conv_output_size(in_shape, ksize, stride, padding, dilation):
    out = np.zeros(3, np.int32)
    lib().orc_conv_output_size(_ip(_triple(in_shape)), _ip(_triple(ksize)), _ip(_triple(stride)),
                               _ip(_triple(padding)), _ip(_triple(dilation)), _ip(out))
    return out


def rulebook_subm(indices, batch_size, spatial_shape, ksize=3, dilation=1):
    """orc_rulebook_subm -> (out_indices, pairs[K,2,N], pair_num[K])."""
    indices = _i(indices)
    n = indices.shape[0]
    ks, dl, shp = _triple(ksize), _triple(dilation), _triple(spatial_shape)
    k = int(np.prod(ks))
    pairs = np.zeros((k, 2, n), np.int32)
    pair_num = np.zeros((k,), np.int32)
    r = lib().orc_rulebook_subm(_ip(indices), n, int(batch_size), _ip(shp), _ip(ks), _ip(dl),
                                _ip(pairs), _ip(pair_num))
    if r < 0:
        raise MemoryError
    return indices.copy(), pairs, pair_num


def rulebook_conv(indices, batch_size, spatial_shape, ksize, stride, padding, dilation=1):
    """orc_rulebook_conv -> (out_indices[M,4], pairs[K,2,N], pair_num[K], out_shape[3])."""
    indices = _i(indices)
    n = indices.shape[0]
    ks, st, pd, dl = _triple(ksize), _triple(stride), _triple(padding), _triple(dilation)
    shp = _triple(spatial_shape)
    out_shape = conv_output_size(shp, ks, st, pd, dl)
    k = int(np.prod(ks))
    out_idx = np.zeros((max(n * k, 1), 4), np.int32)
    pairs = np.zeros((k, 2, n), np.int32)
    pair_num = np.zeros((k,), np.int32)
    m = lib().orc_rulebook_conv(_ip(indices), n, int(batch_size), _ip(shp), _ip(out_shape), _ip(ks),
                                _ip(st), _ip(pd), _ip(dl), _ip(out_idx), _ip(pairs), _ip(pair_num))
    if m < 0:
        raise MemoryError
    return out_idx[:m].copy(), pairs, pair_num, out_shape


def rulebook_conv_sorted(indices, batch_size, spatial_shape, ksize, stride, padding, dilation=1):
    """The strided rulebook in spconv's GPU output numbering (SURVEY.md Appendix A.4 [recall]: the CUDA path of
    spconv src/spconv/indice.cu numbers the outputs by sort / unique of the linear cell index
    lin = ((b * D + z) * H + y) * W + x, ascending; its pair order inside an offset is atomic-arrival order, for which the
    canonical comparison form is ascending input row).  Restated as a relabelling of the first-touch rulebook."""
    out_idx, pairs, pair_num, out_shape = rulebook_conv(indices, batch_size, spatial_shape, ksize, stride, padding, dilation)
    d, h, w = (int(v) for v in out_shape)
    lin = ((out_idx[:, 0].astype(np.int64) * d + out_idx[:, 1]) * h + out_idx[:, 2]) * w + out_idx[:, 3]
    perm = np.argsort(lin, kind="stable")
    rank = np.empty(len(perm), np.int32)
    rank[perm] = np.arange(len(perm), dtype=np.int32)
    pairs = pairs.copy()
    for k in range(pairs.shape[0]):
        c = int(pair_num[k])
        pairs[k, 1, :c] = rank[pairs[k, 1, :c]]
    return out_idx[perm].copy(), pairs, pair_num, out_shape


def indice_conv(features, weight, pairs, pair_num, num_out, acc64=True):
    """orc_indice_conv_fwd.  weight [kD,kH,kW,Cin,Cout] or [K,Cin,Cout]."""
    features = _f(features)
    n_in, cin = features.shape
    weight = _f(weight).reshape(-1, cin, weight.shape[-1])
    k, _, cout = weight.shape
    pairs, pair_num = _i(pairs), _i(pair_num)
    assert pairs.shape == (k, 2, n_in), (pairs.shape, (k, 2, n_in))
    out = np.zeros((num_out, cout), np.float32)
    lib().orc_indice_conv_fwd(_fp(features), n_in, cin, _fp(weight), k, cout, _ip(pairs),
                              _ip(pair_num), int(num_out), int(bool(acc64)), _fp(out))
    return out


def indice_conv_backward(features, weight, pairs, pair_num, dout):
    features, dout = _f(features), _f(dout)
    n_in, cin = features.shape
    wshape = weight.shape
    weight = _f(weight).reshape(-1, cin, wshape[-1])
    k, _, cout = weight.shape
    pairs, pair_num = _i(pairs), _i(pair_num)
    dfeat = np.zeros_like(features)
    dw = np.zeros_like(weight)
    lib().orc_indice_conv_bwd(_fp(features), n_in, cin, _fp(weight), k, cout, _ip(pairs),
                              _ip(pair_num), _fp(dout), _fp(dfeat), _fp(dw))
    return dfeat, dw.reshape(wshape)


def sparse_to_dense(features, indices, batch_size, spatial_shape):
    features, indices = _f(features), _i(indices)
    n, c = features.shape
    shp = _triple(spatial_shape)
    out = np.zeros((batch_size, c, *[int(s) for s in shp]), np.float32)
    lib().orc_sparse_to_dense(_fp(features), _ip(indices), n, c, int(batch_size), _ip(shp), _fp(out))
    return out


def pillar_scatter(features, coords, batch_size, ny, nx):
    features, coords = _f(features), _i(coords)
    p, c = features.shape
    out = np.zeros((batch_size, c, ny, nx), np.float32)
    lib().orc_pillar_scatter(_fp(features), _ip(coords), p, c, int(batch_size), int(ny), int(nx), _fp(out))
    return out


def rotate_iou(boxes, qboxes, criterion=-1):
    boxes, qboxes = _f(boxes), _f(qboxes)
    n, k = boxes.shape[0], qboxes.shape[0]
    out = np.zeros((n, k), np.float32)
    if n and k:
        lib().orc_rotate_iou(_fp(boxes), n, _fp(qboxes), k, int(criterion), _fp(out))
    return out


def standup_iou(dets):
    dets = _f(dets)
    n = dets.shape[0]
    sb = np.zeros((n, 4), np.float32)
    iou = np.zeros((n, n), np.float32)
    if n:
        lib().orc_standup_boxes(_fp(dets), n, dets.shape[1], _fp(sb))
        lib().orc_standup_iou(_fp(sb), n, ctypes.c_float(0.0), _fp(iou))
    return sb, iou


def rotate_nms_sorted(dets, thresh, semantics="numba"):
    """dets sorted by descending score, columns (x,y,w,l,r[,score]).  Returns kept positions."""
    dets = _f(dets)
    n = dets.shape[0]
    keep = np.zeros((max(n, 1),), np.int32)
    sem = {"numba": 0, "cpu": 1}[semantics]
    k = lib().orc_rotate_nms_sorted(_fp(dets), n, dets.shape[1], ctypes.c_float(thresh), sem, _ip(keep)) if n else 0
    return keep[:k].copy()


def nms_sorted(dets, thresh, semantics="numba", eps=1.0):
    dets = _f(dets)
    n = dets.shape[0]
    keep = np.zeros((max(n, 1),), np.int32)
    sem = {"numba": 0, "cpu": 1}[semantics]
    k = lib().orc_nms_sorted(_fp(dets), n, dets.shape[1], ctypes.c_float(thresh), sem,
                             ctypes.c_float(eps), _ip(keep)) if n else 0
    return keep[:k].copy()


def box_decode(encodings, anchors):
    enc, anc = _f(encodings).reshape(-1, 7), _f(anchors).reshape(-1, 7)
    out = np.zeros_like(enc)
    lib().orc_box_decode(_fp(enc), _fp(anc), enc.shape[0], _fp(out))
    return out


def pfn_forward(voxels, num_points, coords, weight_t, scale, shift, vx, vy, x_offset, y_offset):
    """orc_pfn_fwd: weight_t [F+5, C] (= PFNLayer.linear.weight.T), scale/shift = folded eval BatchNorm1d."""
    voxels, num_points, coords = _f(voxels), _i(num_points), _i(coords)
    p, t, f = voxels.shape
    weight_t, scale, shift = _f(weight_t), _f(scale), _f(shift)
    c = weight_t.shape[1]
    out = np.zeros((p, c), np.float32)
    cf = ctypes.c_float
    lib().orc_pfn_fwd(_fp(voxels), _ip(num_points), _ip(coords), p, t, f, _fp(weight_t), _fp(scale), _fp(shift), c,
                      cf(vx), cf(vy), cf(x_offset), cf(y_offset), _fp(out))
    return out


def block_filter(voxels, coors, num_points, grid_size_xy, block_factor, block_size, height_threshold,
                 height_high_threshold=3.0):
    """orc_block_filter -> bool mask [V] (coors are (z,y,x))."""
    voxels, coors, num_points = _f(voxels), _i(coors), _i(num_points)
    v, t, f = voxels.shape
    keep = np.zeros((v,), np.uint8)
    lib().orc_block_filter(_fp(voxels), _ip(coors), _ip(num_points), v, t, f, int(grid_size_xy[0]), int(grid_size_xy[1]),
                           int(block_factor), int(block_size), ctypes.c_float(height_threshold),
                           ctypes.c_float(height_high_threshold), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
    return keep.astype(bool)
