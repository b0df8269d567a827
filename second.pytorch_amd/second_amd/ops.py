"""Tensor-level wrappers over the C ABI (include/second_hip.h).

torch here is plumbing only: it owns device memory and the current HIP stream.  Each function cites
the reference interface its kernel replaces (see the header for file:line).
"""
import numpy as np
import torch

import ctypes
import functools
import os

from . import runtime as rt

# Measurement hook (bench.py's per-kernel roofline table): ``hook(name, fn, args, kwargs, result)`` is called after every traced
# op so that the very same launch (same tensors, same tables) can be re-issued between two HIP events.  None = disabled.
_op_hook = None


def set_op_hook(hook):
    global _op_hook
    _op_hook = hook


def last_kernel_name():
    """The kernel instantiation the last indice_conv / conv2d_nhwc call of this thread dispatched to, as a profiler prints it
    (sec_last_kernel_name); "" for kernels that do not report themselves."""
    return rt.lib().sec_last_kernel_name().decode()


def _traced(name):
    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            res = fn(*args, **kwargs)
            if _op_hook is not None:
                _op_hook(name, fn, args, kwargs, res)
            return res
        wrapper.__wrapped_op__ = fn
        return wrapper
    return deco


# ----------------------------------------------------------------------------- voxelisation
@_traced("voxelize")
def voxelize(points, point_offsets, point_cloud_range, voxel_size, max_points, max_voxels,
             cap_mode="break", mean_features=0, sync=True, mean_dtype=None, fill=True):
    """Batched points_to_voxel (spconv VoxelGeneratorV2.generate; second/data/preprocess.py:301-316).

    points [N,F] float32 cuda (clouds concatenated), point_offsets [B+1] int32 cuda.
    Returns dict(voxels, coordinates [M,4]=(b,z,y,x), num_points_per_voxel, voxel_offsets [B+1], mean?).
    With ``sync=True`` the outputs are sliced to the total voxel count (one D2H of B+1 ints);
    with ``sync=False`` they keep capacity B*max_voxels and only rows < voxel_offsets[B] are defined.
    ``fill=False``: no ``voxels`` tensor (None in the result); the per-voxel point lists stay in the workspace behind
    ``site_table`` for :func:`pfn_forward_slots` (``points`` must stay alive and unchanged until then).
    """
    rt.require_gpu(points, point_offsets)
    assert points.dtype == torch.float32 and points.dim() == 2 and points.is_contiguous()
    assert point_offsets.dtype == torch.int32
    n, f = points.shape
    batch = point_offsets.numel() - 1
    dev = points.device
    if int(max_points) > 256:
        raise rt.SecondHipError(f"voxelize: max_points = {max_points} > 256 points per voxel is not supported (include/second_hip.h)")
    cap = batch * max_voxels
    rows = min(cap, max(n, 1))
    assert fill or not mean_features, "the SimpleVoxel mean is an epilogue of the fill"
    voxels = torch.empty((rows, max_points, f), dtype=torch.float32, device=dev) if fill else None
    coors = torch.empty((rows, 4), dtype=torch.int32, device=dev)
    npv = torch.empty((rows,), dtype=torch.int32, device=dev)
    voff = torch.empty((batch + 1,), dtype=torch.int32, device=dev)
    mean_dtype = mean_dtype or torch.float32      # SimpleVoxel's output in the dtype of the stack that consumes it (no cast launch)
    mean = torch.empty((rows, mean_features), dtype=mean_dtype, device=dev) if mean_features else None
    l = rt.lib()
    ws_bytes = l.sec_voxelize_workspace_bytes(n, batch, max_voxels, max_points)
    ws = rt.workspace(ws_bytes, dev)
    rc = l.sec_voxelize_f32(rt.ptr(points), rt.ptr(point_offsets), n, f, batch, rt.f_arr(point_cloud_range),
                            rt.f_arr(voxel_size), int(max_points), int(max_voxels),
                            {"break": 0, "continue": 1}[cap_mode], rt.ptr(voxels), rt.ptr(coors), rt.ptr(npv),
                            rt.ptr(voff), rt.ptr(mean), int(mean_features), rt.dtype_code(mean_dtype), rt.ptr(ws), ws.numel(), rt.stream())
    rt.check(rc, "sec_voxelize_f32")
    out = {"voxels": voxels, "coordinates": coors, "num_points_per_voxel": npv, "voxel_offsets": voff}
    # the hash table this call leaves in its workspace (cell -> voxel row) is the site lookup of the first SubM rulebook
    r6, v3 = np.asarray(point_cloud_range, np.float32), np.asarray(voxel_size, np.float32)
    grid_zyx = [int(v) for v in np.round((r6[3:] - r6[:3]) / v3).astype(np.int64)[::-1]]
    out["site_table"] = ("vox", ws, int(n), int(max_voxels), int(max_points), grid_zyx)
    if mean is not None:
        out["mean"] = mean
    if sync:
        total = int(voff[-1].item())
        for k in ("voxels", "coordinates", "num_points_per_voxel", "mean"):
            if out.get(k) is not None:
                out[k] = out[k][:total]
        out["voxel_num"] = total
    return out


# ----------------------------------------------------------------------------- rulebooks
def conv_output_shape(in_shape, ksize, stride, padding, dilation):
    import ctypes
    out = (ctypes.c_int * 3)()
    rt.lib().sec_conv_output_shape(rt.i3(in_shape), rt.i3(ksize), rt.i3(stride), rt.i3(padding), rt.i3(dilation), out)
    return [int(v) for v in out]


def _kvol(ksize):
    return int(np.prod([int(k) for k in ((ksize,) * 3 if isinstance(ksize, int) else ksize)]))


@_traced("rulebook_subm")
def rulebook_subm(indices, batch_size, spatial_shape, ksize=3, dilation=1, want_pairs=False, n_dev=None, site_table=None):
    """SubMConv3d rulebook (spconv.ops.get_indice_pairs(subm=True)).  indices [N,4] int32 (b,z,y,x).
    ``n_dev`` (device int32[1]) switches to static-capacity mode: only the first n_dev rows are live.
    ``site_table``: the ``site_table`` entry of the :func:`rulebook_conv` result whose out_indices these are -- its
    hash table is reused instead of re-hashing the sites."""
    rt.require_gpu(indices)
    assert indices.dtype == torch.int32 and indices.is_contiguous()
    n = indices.shape[0]
    k = _kvol(ksize)
    dev = indices.device
    if site_table is not None and not want_pairs and n > 0 and isinstance(site_table[0], str) and site_table[0] == "vox":
        _, ws, vn, vmv, vmp, vgrid = site_table
        ws.record_stream(torch.cuda.current_stream())
        nbr = torch.empty((n, k), dtype=torch.int32, device=dev)
        rc = rt.lib().sec_rulebook_subm3d_after_voxelize(rt.ptr(indices), n, rt.ptr(n_dev), int(batch_size), rt.i3(spatial_shape),
                                                         rt.i3(ksize), rt.i3(dilation), rt.ptr(nbr), rt.ptr(ws), ws.numel(), vn, vmv,
                                                         vmp, rt.i3(vgrid), rt.stream())
        rt.check(rc, "sec_rulebook_subm3d_after_voxelize")
        return {"nbr_out": nbr, "nbr_in": None, "pairs": None, "pair_num": None, "out_indices": indices,
                "num_out": n, "num_out_dev": n_dev, "out_shape": [int(s) for s in spatial_shape]}
    if site_table is not None and not want_pairs and n > 0 and isinstance(site_table[0], str) and site_table[0] == "sorted":
        ws = site_table[1]
        ws.record_stream(torch.cuda.current_stream())
        pre = site_table[2] if len(site_table) > 2 else None     # table already filled with -1 by the strided build's init launch
        if pre is not None and tuple(pre.shape) == (n, k):
            nbr, prefilled = pre, 1
            nbr.record_stream(torch.cuda.current_stream())
        else:
            nbr, prefilled = torch.empty((n, k), dtype=torch.int32, device=dev), 0
        rc = rt.lib().sec_rulebook_subm3d_after_conv_sorted(rt.ptr(indices), n, rt.ptr(n_dev), int(batch_size), rt.i3(spatial_shape),
                                                            rt.i3(ksize), rt.i3(dilation), rt.ptr(nbr), prefilled, rt.ptr(ws), ws.numel(),
                                                            rt.stream())
        rt.check(rc, "sec_rulebook_subm3d_after_conv_sorted")
        return {"nbr_out": nbr, "nbr_in": None, "pairs": None, "pair_num": None, "out_indices": indices,
                "num_out": n, "num_out_dev": n_dev, "out_shape": [int(s) for s in spatial_shape]}
    if site_table is not None and not want_pairs and n > 0:
        ws, cn, cks, cst, cdl, chint = site_table
        ws.record_stream(torch.cuda.current_stream())     # the strided build may have run (and allocated) on another stream
        nbr = torch.empty((n, k), dtype=torch.int32, device=dev)
        rc = rt.lib().sec_rulebook_subm3d_after_conv(rt.ptr(indices), n, rt.ptr(n_dev), int(batch_size), rt.i3(spatial_shape),
                                                     rt.i3(ksize), rt.i3(dilation), rt.ptr(nbr), rt.ptr(ws), ws.numel(), int(cn),
                                                     cks, cst, cdl, int(chint), rt.stream())
        rt.check(rc, "sec_rulebook_subm3d_after_conv")
        return {"nbr_out": nbr, "nbr_in": None, "pairs": None, "pair_num": None, "out_indices": indices,
                "num_out": n, "num_out_dev": n_dev, "out_shape": [int(s) for s in spatial_shape]}
    nbr = torch.empty((n, k), dtype=torch.int32, device=dev)
    pairs = torch.empty((k, 2, n), dtype=torch.int32, device=dev) if want_pairs else None
    pair_num = torch.zeros((k,), dtype=torch.int32, device=dev) if want_pairs else None
    l = rt.lib()
    ws = rt.workspace(l.sec_rulebook_workspace_bytes(n, k, 1), dev)
    rc = l.sec_rulebook_subm3d(rt.ptr(indices), n, rt.ptr(n_dev), int(batch_size), rt.i3(spatial_shape), rt.i3(ksize),
                               rt.i3(dilation), rt.ptr(nbr), rt.ptr(pairs), rt.ptr(pair_num), rt.ptr(ws),
                               ws.numel(), rt.stream())
    rt.check(rc, "sec_rulebook_subm3d")
    return {"nbr_out": nbr, "nbr_in": None, "pairs": pairs, "pair_num": pair_num, "out_indices": indices,
            "num_out": n, "num_out_dev": n_dev, "out_shape": [int(s) for s in spatial_shape]}


def _out_per_in(ks, st, dl):
    """Kernel offsets that can reach an output from one input (see sec_rulebook_workspace_bytes)."""
    tot = 1
    for d in range(3):
        tot *= max(sum(1 for k in range(ks[d]) if (k * dl[d]) % st[d] == r) for r in range(st[d]))
    return tot


_numbering = os.environ.get("SEC_RULEBOOK_NUMBERING", "first_touch")


def set_rulebook_numbering(mode):
    """Default output numbering of :func:`rulebook_conv`: "first_touch" (spconv's CPU path = the oracle's canonical order) or
    "sorted" (spconv's GPU path: ascending linear cell index; no hash table, about half the launches' time).  Returns the
    previous setting."""
    global _numbering
    assert mode in ("first_touch", "sorted")
    prev, _numbering = _numbering, mode
    return prev


def _rulebook_conv_sorted(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, want_pairs, n_dev, out_cap,
                          want_nbr_in, in_sites=None):
    n = indices.shape[0]
    ks, st, pd, dl = rt.i3(ksize), rt.i3(stride), rt.i3(padding), rt.i3(dilation)
    k = _kvol(ksize)
    dev = indices.device
    out_shape = conv_output_shape(spatial_shape, ksize, stride, padding, dilation)
    per_in = _out_per_in(ks, st, dl)
    static = out_cap is not None
    cap = int(out_cap) if static else max(1, min(n * per_in, int(batch_size) * int(np.prod(out_shape))))
    out_idx = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num_out = torch.empty((2,), dtype=torch.int32, device=dev)
    l = rt.lib()
    nbytes = l.sec_rulebook_sorted_workspace_bytes(n, k, int(batch_size), rt.i3(out_shape))
    if nbytes == 0:
        raise ValueError("sorted rulebook numbering: output grid too large")
    ws = rt.workspace(nbytes, dev)
    pre_out = torch.empty((cap, k), dtype=torch.int32, device=dev) if static else None
    pre_in = torch.empty((n, k), dtype=torch.int32, device=dev) if (static and (want_nbr_in or want_pairs)) else None
    geo = (rt.i3(spatial_shape), rt.i3(out_shape), ks, st, pd, dl)
    in_ws = in_sites[1] if (in_sites is not None and isinstance(in_sites[0], str)) else None
    if in_ws is not None:
        in_ws.record_stream(torch.cuda.current_stream())
    # out_indices are written by the tables call (every candidate knows its output's coordinates)
    # static inference pipelines: the 3x3x3 SubM layer that follows on these outputs gets its gather table filled by the same launch
    sub_nbr = torch.empty((cap, 27), dtype=torch.int32, device=dev) if (static and not want_pairs and not torch.is_grad_enabled()) else None
    rc = l.sec_rulebook_conv3d_build_sorted(rt.ptr(indices), n, rt.ptr(n_dev), int(batch_size), *geo, None, cap,
                                            rt.ptr(num_out), rt.ptr(pre_out), cap if static else 0, rt.ptr(pre_in),
                                            rt.ptr(sub_nbr), sub_nbr.numel() if sub_nbr is not None else 0,
                                            rt.ptr(in_ws), in_ws.numel() if in_ws is not None else 0, rt.ptr(ws),
                                            ws.numel(), rt.stream())
    rt.check(rc, "sec_rulebook_conv3d_build_sorted")
    m = cap if static else int(num_out[0].item())
    nbr_out = pre_out if static else torch.empty((m, k), dtype=torch.int32, device=dev)
    if static:
        nbr_in = pre_in
    else:
        nbr_in = torch.empty((n, k), dtype=torch.int32, device=dev) if (want_nbr_in or want_pairs) else None
    pairs = torch.empty((k, 2, n), dtype=torch.int32, device=dev) if want_pairs else None
    pair_num = torch.zeros((k,), dtype=torch.int32, device=dev) if want_pairs else None
    rc = l.sec_rulebook_conv3d_tables_sorted(rt.ptr(indices), n, rt.ptr(n_dev), int(batch_size), *geo, rt.ptr(nbr_out), m,
                                             rt.ptr(nbr_in), 1 if static else 0, rt.ptr(out_idx), cap, rt.ptr(pairs),
                                             rt.ptr(pair_num), rt.ptr(ws), ws.numel(), rt.stream())
    rt.check(rc, "sec_rulebook_conv3d_tables_sorted")
    return {"nbr_out": nbr_out, "nbr_in": nbr_in, "pairs": pairs, "pair_num": pair_num,
            "out_indices": out_idx[:m], "num_out": m, "num_out_dev": num_out if static else None,
            "out_shape": out_shape, "site_table": ("sorted", ws, sub_nbr) if n > 0 else None}


@_traced("rulebook_conv")
def rulebook_conv(indices, batch_size, spatial_shape, ksize, stride, padding, dilation=1, want_pairs=False,
                  n_dev=None, out_cap=None, out_per_in_hint=0, want_nbr_in=True, numbering=None, in_sites=None):
    """SparseConv3d rulebook (spconv.ops.get_indice_pairs(subm=False)); output numbering "first_touch" (default: spconv's CPU
    path, the oracle's canonical order) or "sorted" (spconv's GPU path, ascending linear cell index) -- see
    :func:`set_rulebook_numbering`.  ``in_sites`` (sorted numbering only): the ``site_table`` entry of the sorted build whose
    out_indices these ``indices`` are -- the output bitmap is then derived from that build's bitmap without atomics.

    Eager mode (default): one D2H sync for the active-output count, like the reference's numActOut.
    Static-capacity mode (``out_cap`` given, optional ``n_dev``): no sync; tables are sized ``out_cap`` rows,
    ``num_out_dev`` int32[2] = (live outputs clamped to out_cap, raw count for the overflow check)."""
    rt.require_gpu(indices)
    assert indices.dtype == torch.int32 and indices.is_contiguous()
    if (numbering or _numbering) == "sorted":
        return _rulebook_conv_sorted(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, want_pairs, n_dev,
                                     out_cap, want_nbr_in, in_sites)
    n = indices.shape[0]
    ks, st = rt.i3(ksize), rt.i3(stride)
    k = _kvol(ksize)
    dev = indices.device
    out_shape = conv_output_shape(spatial_shape, ksize, stride, padding, dilation)
    dl = rt.i3(dilation)
    per_in = _out_per_in(ks, st, dl)
    hint = int(out_per_in_hint) if 0 < int(out_per_in_hint) < per_in else 0
    static = out_cap is not None
    cap = int(out_cap) if static else max(1, min(n * per_in, int(batch_size) * int(np.prod(out_shape))))
    out_idx = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num_out = torch.empty((2,), dtype=torch.int32, device=dev)   # both words are written by the build
    l = rt.lib()
    ws = rt.workspace(l.sec_rulebook_workspace_bytes(n, k, hint or per_in), dev)
    # static capacity: the table sizes are known before the build, whose numbering launch then also writes their -1 fill
    pre_out = torch.empty((cap, k), dtype=torch.int32, device=dev) if static else None
    pre_in = torch.empty((n, k), dtype=torch.int32, device=dev) if (static and (want_nbr_in or want_pairs)) else None
    rc = l.sec_rulebook_conv3d_build(rt.ptr(indices), n, rt.ptr(n_dev), int(batch_size), rt.i3(spatial_shape),
                                     rt.i3(out_shape), ks, st, rt.i3(padding), rt.i3(dilation), rt.ptr(out_idx), cap,
                                     rt.ptr(num_out), hint, rt.ptr(pre_out), cap if static else 0, rt.ptr(pre_in), rt.ptr(ws),
                                     ws.numel(), rt.stream())
    rt.check(rc, "sec_rulebook_conv3d_build")
    m = cap if static else int(num_out[0].item())
    nbr_out = pre_out if static else torch.empty((m, k), dtype=torch.int32, device=dev)
    if static:
        nbr_in = pre_in
    else:
        nbr_in = torch.empty((n, k), dtype=torch.int32, device=dev) if (want_nbr_in or want_pairs) else None
    pairs = torch.empty((k, 2, n), dtype=torch.int32, device=dev) if want_pairs else None
    pair_num = torch.zeros((k,), dtype=torch.int32, device=dev) if want_pairs else None
    rc = l.sec_rulebook_conv3d_tables(n, ks, st, dl, hint, rt.ptr(nbr_out), m, rt.ptr(nbr_in), 1 if static else 0, rt.ptr(pairs),
                                      rt.ptr(pair_num), rt.ptr(ws), ws.numel(), rt.stream())
    rt.check(rc, "sec_rulebook_conv3d_tables")
    return {"nbr_out": nbr_out, "nbr_in": nbr_in, "pairs": pairs, "pair_num": pair_num,
            "out_indices": out_idx[:m], "num_out": m, "num_out_dev": num_out if static else None,
            "out_shape": out_shape, "site_table": (ws, n, ks, st, dl, hint) if n > 0 else None}


@_traced("rulebook_chain")
def rulebook_chain(indices0, batch_size, shape0, convs, n_dev=None, site_table=None, want_subm=None, want_site_map=False):
    """Every rulebook of a SubM / strided-conv stack in ONE call (sec_rulebook_chain_sorted; spconv GPU numbering), static
    capacity only.  ``indices0`` [n0, 4] = level-0 rows; ``convs`` = [(ksize, stride, padding, out_cap), ...] the strided layers in
    order; ``want_subm`` [levels + 1] bools: a 3x3x3 SubM table on that level's sites (level 0 needs ``site_table`` = the
    ``site_table`` entry of the :func:`voxelize` result the rows come from).  Returns None when the geometry is outside what the
    fused build covers (build layer by layer then), else a dict: ``levels`` = list (index 0 .. L) of dicts with ``shape``,
    ``indices``, ``num_dev`` (int32[2]: live clamped, raw), ``cap``, ``nbr_out`` (conv table, level >= 1), ``subm_nbr``; ``site_map``."""
    rt.require_gpu(indices0)
    assert indices0.dtype == torch.int32 and indices0.is_contiguous()
    levels = len(convs)
    want_subm = list(want_subm) if want_subm is not None else [False] * (levels + 1)
    assert len(want_subm) == levels + 1
    dev = indices0.device
    n0 = indices0.shape[0]
    if n0 == 0 or levels == 0:
        return None
    shapes = [[int(v) for v in shape0]]
    ks_all, st_all, pd_all, caps = [], [], [], []
    for ks, st, pd, cap in convs:
        ks3, st3, pd3 = list(rt.i3(ks)), list(rt.i3(st)), list(rt.i3(pd))
        shapes.append(conv_output_shape(shapes[-1], ks3, st3, pd3, 1))
        ks_all += ks3; st_all += st3; pd_all += pd3
        caps.append(int(cap))
    vox = site_table if (site_table is not None and isinstance(site_table[0], str) and site_table[0] == "vox") else None
    if want_subm[0] and vox is None:
        return None
    ia = lambda v: (ctypes.c_int * len(v))(*[int(x) for x in v])
    flat_shapes = ia([v for sh in shapes for v in sh])
    l = rt.lib()
    nbytes = l.sec_rulebook_chain_workspace_bytes(int(batch_size), levels, flat_shapes)
    if nbytes == 0:
        return None
    ws = rt.workspace(nbytes, dev)
    kvols = [int(np.prod(ks_all[3 * i:3 * i + 3])) for i in range(levels)]
    nbr = [torch.empty((caps[i], kvols[i]), dtype=torch.int32, device=dev) for i in range(levels)]
    oidx = [torch.empty((caps[i], 4), dtype=torch.int32, device=dev) for i in range(levels)]
    nout = [torch.empty((2,), dtype=torch.int32, device=dev) for _ in range(levels)]
    rows = [n0] + caps
    subm = [torch.empty((rows[i], 27), dtype=torch.int32, device=dev) if want_subm[i] else None for i in range(levels + 1)]
    smap = torch.empty((int(batch_size), *shapes[-1]), dtype=torch.int32, device=dev) if want_site_map else None
    pa = lambda ts: (ctypes.c_void_p * len(ts))(*[(t.data_ptr() if t is not None else None) for t in ts])
    if vox is not None:
        _, vws, vn, vmv, vmp, vgrid = vox
        vws.record_stream(torch.cuda.current_stream())
    else:
        vws, vn, vmv, vmp, vgrid = None, 0, 0, 0, shapes[0]
    rc = l.sec_rulebook_chain_sorted(rt.ptr(indices0), n0, rt.ptr(n_dev), int(batch_size), levels, flat_shapes, ia(ks_all), ia(st_all),
                                     ia(pd_all), ia(caps), pa(nbr), pa(oidx), pa(nout), pa(subm), rt.ptr(vws),
                                     vws.numel() if vws is not None else 0, int(vn), int(vmv), int(vmp), rt.i3(vgrid), rt.ptr(smap),
                                     rt.ptr(ws), ws.numel(), rt.stream())
    if rc == -3:
        return None
    rt.check(rc, "sec_rulebook_chain_sorted")
    out = [{"shape": shapes[0], "indices": indices0, "num_dev": n_dev, "cap": n0, "nbr_out": None, "subm_nbr": subm[0]}]
    for i in range(levels):
        out.append({"shape": shapes[i + 1], "indices": oidx[i], "num_dev": nout[i], "cap": caps[i], "nbr_out": nbr[i],
                    "subm_nbr": subm[i + 1]})
    return {"levels": out, "site_map": smap, "workspace": ws}


def simple_voxel(voxels, num_points, mean_features, out_dtype=None, num_dev=None):
    """SimpleVoxel.forward (voxel_encoder.py:220-225) of a voxel tensor the caller already holds ([N, T, F] fp32, num_points [N] int32):
    -> [N, mean_features] in ``out_dtype`` (default fp32), one launch, the operation order of the voxeliser's fused epilogue
    (sec_simple_voxel_f32).  ``num_dev``: device int32[1], rows at or past it come out as zeros."""
    rt.require_gpu(voxels, num_points)
    assert voxels.dtype == torch.float32 and voxels.dim() == 3 and voxels.is_contiguous()
    assert num_points.dtype == torch.int32 and num_points.is_contiguous() and num_points.numel() == voxels.shape[0]
    n, t, f = voxels.shape
    out_dtype = out_dtype or torch.float32
    mean = torch.empty((n, int(mean_features)), dtype=out_dtype, device=voxels.device)
    rt.check(rt.lib().sec_simple_voxel_f32(rt.ptr(voxels), rt.ptr(num_points), n, rt.ptr(num_dev), t, f, int(mean_features), rt.ptr(mean),
                                           rt.dtype_code(out_dtype), rt.stream()), "sec_simple_voxel_f32")
    return mean


def tensors_checksum(tensors):
    """int64 [len(tensors), 2] on the device: (sum of 32-bit words, sum of word * (index + 1)) mod 2^64 of every tensor's bytes, ONE
    launch for all of them (sec_tensors_checksum; contiguous CUDA tensors whose byte size is a multiple of 4).  Captured in a graph, the
    launch keeps reading the same storages."""
    import ctypes
    n = len(tensors)
    for t in tensors:
        rt.require_gpu(t)
        assert t.is_contiguous()
    sums = torch.empty((n, 2), dtype=torch.int64, device=tensors[0].device)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
    nbytes = (ctypes.c_longlong * n)(*[t.numel() * t.element_size() for t in tensors])
    rt.check(rt.lib().sec_tensors_checksum(ptrs, nbytes, n, rt.ptr(sums), rt.stream()), "sec_tensors_checksum")
    return sums


def rows_differ_(flag, a, b):
    """flag[0] |= 1 (int32 device tensor, zeroed by the caller) when any row of ``a`` [rows, ...] differs from ``b`` [...] -- contiguous
    fp32 tensors, bit compare, one launch (sec_rows_differ_f32)."""
    rt.require_gpu(flag, a, b)
    assert a.dtype == b.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous() and flag.dtype == torch.int32
    n = b.numel()
    assert n > 0 and a.numel() % n == 0
    rt.check(rt.lib().sec_rows_differ_f32(rt.ptr(a), a.numel() // n, rt.ptr(b), n, rt.ptr(flag), rt.stream()), "sec_rows_differ_f32")
    return flag


# ----------------------------------------------------------------------------- indice_conv
_conv_profiler = None


def set_conv_profiler(profiler):
    """Measurement hook (bench.py): ``profiler.begin(meta) -> token`` / ``profiler.end(token)`` bracket the
    sec_indice_conv_fwd launch on the current stream.  None disables it."""
    global _conv_profiler
    _conv_profiler = profiler


FP32_MODES = {"split16": 0, "exact": 1}
# How this package names the split-operand arithmetic wherever a dtype is reported (bench records, docs): fp32 STORAGE, products from
# three bf16 MFMA passes on (hi, lo) operand pairs -- 16 significant bits per operand, fp32 accumulation.  Not the same thing as fp32.
FP32_SPLIT_LABEL = "bf16x3"


def set_fp32_mode(mode):
    """Arithmetic of the fp32 sparse convolutions (include/second_hip.h sec_set_fp32_mode): "split16" (default; SEC_FP32_MODE in the
    environment overrides the default at import) or "exact" (IEEE fp32 products on v_mfma_f32_32x32x2_f32 / VALU -- the reference's
    arithmetic).  Returns the previous mode's name.  Process-wide; a captured graph keeps the mode it was captured under."""
    l = rt.lib()
    prev = "exact" if l.sec_get_fp32_mode() == 1 else "split16"
    rt.check(l.sec_set_fp32_mode(FP32_MODES[mode]), "sec_set_fp32_mode")
    return prev


def get_fp32_mode():
    return "exact" if rt.lib().sec_get_fp32_mode() == 1 else "split16"


class fp32_mode:
    """``with ops.fp32_mode("exact"): ...`` -- the launches issued inside use that arithmetic; None = leave the mode alone."""

    def __init__(self, mode):
        self.mode, self.prev = mode, None

    def __enter__(self):
        if self.mode is not None:
            self.prev = set_fp32_mode(self.mode)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            set_fp32_mode(self.prev)
        return False


def pack_weight(weight):
    """Fragment-order copy of a [kD,kH,kW,Cin,Cout] weight for the MFMA path; None when not applicable."""
    rt.require_gpu(weight)
    cin, cout = weight.shape[-2], weight.shape[-1]
    k = weight.numel() // (cin * cout)
    l = rt.lib()
    if weight.dtype == torch.float32 and l.sec_get_fp32_mode() == 1:
        return None                                     # exact fp32: the kernels read the weight itself
    if weight.dtype == torch.float32:
        # fp32 layers run on the bf16 matrix pipe with split operands (sec_indice_conv_fwd): per offset the fragment image of
        # bf16(W) followed by that of bf16(W - bf16(W)); None for shapes without an instantiation (VALU / on-the-fly paths then)
        if l.sec_packed_weight_x3_bytes(k, cin, cout) == 0:
            return None
        w = weight.detach().reshape(k, cin, cout).float().contiguous()
        hi = w.to(torch.bfloat16)
        lo = (w - hi.float()).to(torch.bfloat16)
        planes = []
        for plane in (hi, lo):
            img = torch.empty((l.sec_packed_weight_bytes(k, cin, cout, rt.dtype_code(torch.bfloat16)) // 2,), dtype=torch.bfloat16, device=w.device)
            rt.check(l.sec_pack_conv_weight(rt.ptr(plane.contiguous()), k, cin, cout, rt.dtype_code(torch.bfloat16), rt.ptr(img), rt.stream()),
                     "sec_pack_conv_weight")
            planes.append(img.view(k, -1))
        return torch.stack(planes, 1).contiguous()          # [k][hi | lo][fragment image of one offset]
    code = rt.dtype_code(weight.dtype)
    nbytes = l.sec_packed_weight_bytes(k, cin, cout, code)
    if nbytes == 0:
        return None
    packed = torch.empty((nbytes // 2,), dtype=weight.dtype, device=weight.device)
    rt.check(l.sec_pack_conv_weight(rt.ptr(weight.contiguous()), k, cin, cout, code, rt.ptr(packed), rt.stream()),
             "sec_pack_conv_weight")
    return packed


@_traced("indice_conv")
def indice_conv(features, weight, nbr_out, num_out, packed=None, scale=None, shift=None, relu=False,
                out_dtype=None, num_out_dev=None):
    """out[o] = sum_k features[nbr_out[o][k]] @ W[k], optional fused scale/shift/ReLU epilogue
    (spconv.ops.indice_conv / indice_subm_conv)."""
    rt.require_gpu(features, weight, nbr_out)
    assert features.is_contiguous() and weight.is_contiguous() and nbr_out.is_contiguous()
    assert features.dtype == weight.dtype, (features.dtype, weight.dtype)
    cin, cout = weight.shape[-2], weight.shape[-1]
    k = weight.numel() // (cin * cout)
    assert features.shape[1] == cin and nbr_out.shape[1] == k
    out_dtype = out_dtype or features.dtype
    out = torch.empty((num_out, cout), dtype=out_dtype, device=features.device)
    for t in (scale, shift):
        assert t is None or (t.dtype == torch.float32 and t.is_cuda and t.numel() == cout)
    token = None
    if _conv_profiler is not None:
        token = _conv_profiler.begin({"cin": cin, "cout": cout, "kvol": k, "n_in": features.shape[0],
                                      "n_out": int(num_out), "dtype": features.dtype, "nbr_out": nbr_out,
                                      "mfma": packed is not None, "num_out_dev": num_out_dev,
                                      "args": {"pos": (features, weight, nbr_out, num_out),
                                               "kw": dict(packed=packed, scale=scale, shift=shift, relu=relu,
                                                          out_dtype=out_dtype, num_out_dev=num_out_dev)}})
    rc = rt.lib().sec_indice_conv_fwd(rt.ptr(features), features.shape[0], cin, rt.ptr(weight), rt.ptr(packed), k, cout,
                                      rt.ptr(nbr_out), int(num_out), rt.ptr(num_out_dev), rt.ptr(scale), rt.ptr(shift),
                                      int(bool(relu)), rt.ptr(out), rt.dtype_code(features.dtype),
                                      rt.dtype_code(out_dtype), rt.stream())
    if token is not None:
        _conv_profiler.end(token)
    rt.check(rc, "sec_indice_conv_fwd")
    return out


def indice_conv_plan(cin, cout, kvol, num_out, dtype, out_dtype=None, packed=True):
    """Kernel id sec_indice_conv_fwd would dispatch for this shape (include/second_hip.h: 6 = the row-split SubM kernel)."""
    return int(rt.lib().sec_indice_conv_fwd_plan(int(cin), int(cout), int(kvol), int(num_out), rt.dtype_code(dtype),
                                                 rt.dtype_code(out_dtype or dtype), int(bool(packed))))


def indice_conv_set_variant(variant):
    """Force one kernel family of the 16-bit sparse conv (A/B runs, parity tests); -1 restores the automatic choice."""
    rt.check(rt.lib().sec_indice_conv_set_variant(int(variant)), "sec_indice_conv_set_variant")


_PREPACK = {}


def prepack_training_weights(sparse_items, dense_items, dtype):
    """Every weight image a mixed-precision training step needs, in TWO launches instead of one per layer
    (sec_pack_conv_weight_train_multi / sec_conv2d_pack_weight_train_multi): ``sparse_items`` = [(weight [kD,kH,kW,Cin,Cout] fp32,
    subm, want_dweight)] of the sparse convolutions, ``dense_items`` = [weight [Cout,Cin,k,k] fp32] of the RPN convolutions that run
    on the hand-written kernels.  The results wait in a table keyed by the weight's storage; :func:`pack_weight_train` /
    :func:`conv2d_pack_weight_train` hand them out (once) when the layer runs.  Call at the start of every step: the table is
    cleared first, so an image can never outlive the weights it was made from."""
    import ctypes
    _PREPACK.clear()
    l, code = rt.lib(), rt.dtype_code(dtype)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    arr_p = lambda ts: (vp * len(ts))(*[None if t is None else t.data_ptr() for t in ts])
    arr_i = lambda vs: (ci * len(vs))(*[int(v) for v in vs])
    sp = [(w, bool(subm), bool(want)) for (w, subm, want) in sparse_items
          if w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()]
    if sp:
        ws, ks, cins, couts, subs, w16s, pks, pkts, dws = [], [], [], [], [], [], [], [], []
        for w, subm, want in sp:
            cin, cout = w.shape[-2], w.shape[-1]
            k = w.numel() // (cin * cout)
            nf = l.sec_packed_weight_bytes(k, cin, cout, code)
            nt = l.sec_packed_weight_bytes(k, cout, cin, code) if cout % 16 == 0 else 0
            ws.append(w); ks.append(k); cins.append(cin); couts.append(cout); subs.append(int(subm))
            w16s.append(torch.empty(w.shape, dtype=dtype, device=w.device))
            pks.append(torch.empty((nf // 2,), dtype=dtype, device=w.device) if nf else None)
            pkts.append(torch.empty((nt // 2,), dtype=dtype, device=w.device) if nt else None)
            dws.append(torch.empty(w.shape, dtype=torch.float32, device=w.device) if want else None)
        rt.check(l.sec_pack_conv_weight_train_multi(len(sp), arr_p(ws), arr_i(ks), arr_i(cins), arr_i(couts), arr_i(subs), code, arr_p(w16s),
                                                    arr_p(pks), arr_p(pkts), arr_p(dws), rt.stream()), "sec_pack_conv_weight_train_multi")
        for (w, subm, _), w16, pk, pkt, dw in zip(sp, w16s, pks, pkts, dws):
            _PREPACK[("sp", w.data_ptr(), dtype, subm)] = (w16, pk, pkt, dw)
    de = [w for w in dense_items if w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 4 and w.shape[2] == w.shape[3]
          and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0]
    if de:
        both = []
        for w in de:
            cout, cin, k, _ = w.shape
            nb = l.sec_conv2d_packed_weight_bytes(cout, cin, k, code)
            both.append(torch.empty((2, nb // 2), dtype=dtype, device=w.device))
        rt.check(l.sec_conv2d_pack_weight_train_multi(len(de), arr_p(de), arr_i([w.shape[0] for w in de]), arr_i([w.shape[1] for w in de]),
                                                      arr_i([w.shape[2] for w in de]), code, arr_p([b[0] for b in both]),
                                                      arr_p([b[1] for b in both]), rt.stream()), "sec_conv2d_pack_weight_train_multi")
        for w, b in zip(de, both):
            _PREPACK[("2d", w.data_ptr(), dtype)] = (b[0], b[1])


def pack_weight_train(weight, dtype, subm, zero_grad=False):
    """fp32 master weight [kD,kH,kW,Cin,Cout] -> (16-bit copy, forward MFMA image or None, data-gradient MFMA image or None) in
    ONE launch (sec_pack_conv_weight_train): what a mixed-precision step otherwise spends to(dtype) + pack_weight + the transposed
    pack inside indice_conv_backward on.  ``subm``: the data-gradient image is offset-mirrored (SubM rulebooks have no input-major
    table).  ``zero_grad``: a fourth result, a zeroed fp32 tensor of the weight's shape written by the same launch -- hand it to
    :func:`indice_conv_backward` as ``dweight_out`` (its own zeroing is a memset node per layer and step)."""
    rt.require_gpu(weight)
    assert weight.dtype == torch.float32 and weight.is_contiguous() and dtype in (torch.bfloat16, torch.float16)
    hit = _PREPACK.pop(("sp", weight.data_ptr(), dtype, bool(subm)), None)     # packed by prepack_training_weights in this step's one launch
    if hit is not None:
        w16, pk, pkt, dw0 = hit
        if zero_grad and dw0 is None:
            dw0 = torch.zeros(weight.shape, dtype=torch.float32, device=weight.device)
        return (w16, pk, pkt, dw0) if zero_grad else (w16, pk, pkt)
    cin, cout = weight.shape[-2], weight.shape[-1]
    k = weight.numel() // (cin * cout)
    l, code = rt.lib(), rt.dtype_code(dtype)
    nf, nt = l.sec_packed_weight_bytes(k, cin, cout, code), (l.sec_packed_weight_bytes(k, cout, cin, code) if cout % 16 == 0 else 0)
    w16 = torch.empty(weight.shape, dtype=dtype, device=weight.device)
    pk = torch.empty((nf // 2,), dtype=dtype, device=weight.device) if nf else None
    pkt = torch.empty((nt // 2,), dtype=dtype, device=weight.device) if nt else None
    dw0 = torch.empty(weight.shape, dtype=torch.float32, device=weight.device) if zero_grad else None
    rt.check(l.sec_pack_conv_weight_train(rt.ptr(weight), k, cin, cout, int(bool(subm)), code, rt.ptr(w16), rt.ptr(pk), rt.ptr(pkt),
                                          rt.ptr(dw0), rt.stream()), "sec_pack_conv_weight_train")
    return (w16, pk, pkt, dw0) if zero_grad else (w16, pk, pkt)


def indice_conv_backward(features, weight, nbr_out, nbr_in, dout, need_dfeat=True, need_dweight=True, dweight_dtype=None,
                         packed_dgrad=None, dweight_out=None):
    """(dfeat, dweight) of indice_conv (spconv.ops.indice_conv_backward). nbr_in None => SubM mirror.  The kernels accumulate
    dweight in fp32; it is returned in ``dweight_dtype`` (default: the weight's dtype; torch.float32 hands a mixed-precision
    caller the unrounded gradient of its fp32 master weight without a cast launch).  ``packed_dgrad``: the data-gradient image of
    :func:`pack_weight_train` (mirrored iff nbr_in is None) -- the call then skips its own transposed pack.  ``dweight_out``: the
    ZEROED fp32 accumulator of pack_weight_train(zero_grad=True); the gradient is accumulated into it and it is what comes back."""
    rt.require_gpu(features, weight, nbr_out, dout)
    cin, cout = weight.shape[-2], weight.shape[-1]
    k = weight.numel() // (cin * cout)
    dout = dout.contiguous()
    dfeat = torch.empty_like(features) if need_dfeat else None
    prezeroed = need_dweight and dweight_out is not None
    if prezeroed:
        assert dweight_out.dtype == torch.float32 and dweight_out.is_contiguous() and dweight_out.numel() == weight.numel()
    dw = (dweight_out if prezeroed else torch.empty(weight.shape, dtype=torch.float32, device=weight.device)) if need_dweight else None
    l = rt.lib()
    ws = rt.workspace(l.sec_indice_conv_bwd_workspace_bytes(k, cin, cout, rt.dtype_code(features.dtype)), features.device)
    rc = l.sec_indice_conv_bwd(rt.ptr(features), features.shape[0], cin, rt.ptr(weight), k, cout, rt.ptr(nbr_out),
                               rt.ptr(nbr_in), dout.shape[0], rt.ptr(dout), rt.ptr(dfeat), rt.ptr(dw),
                               rt.dtype_code(features.dtype), rt.ptr(ws), ws.numel(), rt.ptr(packed_dgrad), int(prezeroed), rt.stream())
    rt.check(rc, "sec_indice_conv_bwd")
    return dfeat, (dw.to(dweight_dtype or weight.dtype) if dw is not None else None)


# ----------------------------------------------------------------------------- scatters
@_traced("sparse_to_dense")
def sparse_to_dense(features, indices, batch_size, spatial_shape, channels_last_2d=False, num_dev=None):
    """SparseConvTensor.dense().  Default: [B,C,D,H,W] contiguous.  ``channels_last_2d``: a
    [B, C*D, H, W] tensor in torch.channels_last memory format (what the RPN consumes) -- same values as
    dense().view(B, C*D, H, W) (second/pytorch/models/middle.py:206-210), no permute copy."""
    rt.require_gpu(features, indices)
    n, c = features.shape
    d, h, w = [int(s) for s in spatial_shape]
    b = int(batch_size)
    if channels_last_2d:
        out = torch.empty((b, c * d, h, w), dtype=features.dtype, device=features.device,
                          memory_format=torch.channels_last)
        sb, sc2, sy, sx = out.stride()
        strides = (sb, sc2 * d, sc2, sy, sx)  # channel index = c*D + z
    else:
        out = torch.empty((b, c, d, h, w), dtype=features.dtype, device=features.device)
        strides = out.stride()
    rc = rt.lib().sec_sparse_to_dense(rt.ptr(features.contiguous()), rt.ptr(indices), n, c, rt.ptr(num_dev), rt.ptr(out),
                                      out.numel(), *[int(s) for s in strides], rt.dtype_code(features.dtype), rt.stream())
    rt.check(rc, "sec_sparse_to_dense")
    return out


def dense_to_sparse(dense, indices, num_dev=None, depth=0):
    """rows[i,:] = dense[b_i, :, (z_i,) y_i, x_i]: the adjoint of :func:`sparse_to_dense` ([B,C,D,H,W]; with ``depth`` = D > 0 of its
    ``channels_last_2d`` form [B, C*D, H, W], channel = c * D + z, in any strides) and of :func:`pillar_scatter` ([B,C,H,W], z
    ignored) -- what autograd needs for their backward.  ``num_dev``: static capacity -- only the first num_dev[0] rows are gathered
    (the others stay unwritten)."""
    rt.require_gpu(dense, indices)
    n = indices.shape[0]
    c = dense.shape[1]
    st = [int(v) for v in dense.stride()]
    if dense.dim() == 5:
        sb, sc, sz, sy, sx = st
    elif depth:
        assert c % depth == 0
        c //= depth
        sb, sc, sz, sy, sx = st[0], st[1] * depth, st[1], st[2], st[3]
    else:
        (sb, sc, sy, sx), sz = st, 0
    rows = torch.empty((n, c), dtype=dense.dtype, device=dense.device)
    rc = rt.lib().sec_dense_to_sparse(rt.ptr(dense), rt.ptr(indices.contiguous()), n, c, rt.ptr(num_dev), rt.ptr(rows), sb, sc, sz, sy, sx,
                                      rt.dtype_code(dense.dtype), rt.stream())
    rt.check(rc, "sec_dense_to_sparse")
    return rows


@_traced("pillar_scatter")
def pillar_scatter(features, coords, batch_size, ny, nx, channels_last=False, num_dev=None):
    """PointPillarsScatter.forward (second/pytorch/models/pointpillars.py:444-476) as one launch."""
    rt.require_gpu(features, coords)
    p, c = features.shape
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    out = torch.empty((int(batch_size), c, int(ny), int(nx)), dtype=features.dtype, device=features.device,
                      memory_format=fmt)
    sb, sc, sy, sx = out.stride()
    rc = rt.lib().sec_pillar_scatter(rt.ptr(features.contiguous()), rt.ptr(coords), p, c, rt.ptr(num_dev), rt.ptr(out), out.numel(),
                                     int(sb), int(sc), int(sy), int(sx), rt.dtype_code(features.dtype), rt.stream())
    rt.check(rc, "sec_pillar_scatter")
    return out


@_traced("pfn_forward")
def pfn_forward(voxels, num_points, coords, weight_t, scale, shift, vx, vy, x_offset, y_offset, out_dtype=None,
                num_dev=None):
    """Fused PillarFeatureNet (one PFNLayer, eval): decorate -> Linear(9,C) -> folded BN -> ReLU -> max over points
    (second/pytorch/models/pointpillars.py:203-237,51-65).  voxels [P,T,4] fp32, coords [P,4] (b,z,y,x),
    weight_t [9,C] = linear.weight.T, scale/shift fp32 [C]."""
    rt.require_gpu(voxels, num_points, coords, weight_t, scale, shift)
    assert voxels.dtype == torch.float32 and voxels.is_contiguous() and coords.dtype == torch.int32
    p, t, f = voxels.shape
    c = weight_t.shape[1]
    assert weight_t.shape[0] == f + 5 and weight_t.dtype == torch.float32 and weight_t.is_contiguous()
    out_dtype = out_dtype or torch.float32
    out = torch.empty((p, c), dtype=out_dtype, device=voxels.device)
    rc = rt.lib().sec_pfn_fwd(rt.ptr(voxels), rt.ptr(num_points), rt.ptr(coords.contiguous()), p, rt.ptr(num_dev), t, f,
                              rt.ptr(weight_t), rt.ptr(scale), rt.ptr(shift), c, float(vx), float(vy), float(x_offset),
                              float(y_offset), rt.ptr(out), rt.dtype_code(out_dtype), rt.stream())
    rt.check(rc, "sec_pfn_fwd")
    return out


# ---- BatchNorm step counters of the fused training paths ----------------------------------------------------------------------
# `bn.num_batches_tracked += 1` is one tiny launch per BatchNorm layer (21 per step of the SECOND networks).  Inside
# deferred_bn_counters() the fused paths only note the counter; leaving the context adds 1 to all of them with one multi-tensor launch.
_bn_counter_stack = []      # frames: [pending counters, frame opened while the stream was being captured]


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


class deferred_bn_counters:
    def __enter__(self):
        _bn_counter_stack.append([[], _capturing()])
        return self

    def __exit__(self, *exc):
        pending, _ = _bn_counter_stack.pop()
        if pending:
            torch._foreach_add_(pending, 1)
        return False


def bump_bn_counter(bn):
    """num_batches_tracked += 1 now, or at the exit of the enclosing :class:`deferred_bn_counters`.  While a stream is being
    captured the increment is deferred only when the frame itself was opened inside the capture (DeviceTrainer.capture_step: the
    one multi-tensor add at its exit is then part of the same graph); a frame opened outside must not swallow increments a
    captured segment has to replay."""
    if _bn_counter_stack and (_bn_counter_stack[-1][1] or not (bn.num_batches_tracked.is_cuda and _capturing())):
        _bn_counter_stack[-1][0].append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1


def pfn_train_supported(voxels, channels):
    return bool(voxels.is_cuda and voxels.dtype == torch.float32 and voxels.dim() == 3 and voxels.shape[2] == 4
                and voxels.shape[1] <= 127 and channels <= 64)


class PFNTrainFunction(torch.autograd.Function):
    """PFNLayer in training mode (Linear(9, C) -> BatchNorm1d batch statistics -> ReLU -> max over the points; pointpillars.py:51-65)
    on sec_pfn_train_fwd / sec_pfn_train_bwd: the [P, T, C] activation tensor of the torch formulation never exists.  Gradients for
    linear.weight [C, 9], the BatchNorm weight and bias; the points get none (they are data).  running_mean / running_var are
    updated in place like torch.nn.BatchNorm1d does."""

    @staticmethod
    def forward(ctx, voxels, num_points, coords, weight, gamma, beta, running_mean, running_var, eps, momentum, geom):
        rt.require_gpu(voxels, num_points, coords, weight, gamma, beta)
        assert voxels.dtype == torch.float32 and voxels.is_contiguous() and coords.dtype == torch.int32
        p, t, f = voxels.shape
        c = weight.shape[0]
        wt = weight.detach().float().t().contiguous()
        ga, be = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        dev = voxels.device
        out = torch.empty((p, c), dtype=torch.float32, device=dev)
        arg = torch.empty((p, c), dtype=torch.int8, device=dev)
        stats = torch.empty((c * 11 + 9,), dtype=torch.float32, device=dev)
        l = rt.lib()
        ws = rt.workspace(l.sec_pfn_train_workspace_bytes(p, c), dev)
        num_points, coords = num_points.int().contiguous(), coords.contiguous()
        vx, vy, xo, yo = [float(v) for v in geom]
        rc = l.sec_pfn_train_fwd(rt.ptr(voxels), rt.ptr(num_points), rt.ptr(coords), p, t, f, rt.ptr(wt), rt.ptr(ga), rt.ptr(be),
                                 float(eps), float(momentum), rt.ptr(running_mean), rt.ptr(running_var), c, vx, vy, xo, yo,
                                 rt.ptr(out), rt.ptr(arg), rt.ptr(stats), rt.ptr(ws), ws.numel(), rt.stream())
        rt.check(rc, "sec_pfn_train_fwd")
        ctx.save_for_backward(voxels, num_points, coords, wt, ga, stats, out, arg)
        ctx.geom = (vx, vy, xo, yo)
        ctx.dtypes = (weight.dtype, gamma.dtype, beta.dtype)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        voxels, num_points, coords, wt, ga, stats, out, arg = ctx.saved_tensors
        p, t, f = voxels.shape
        c = wt.shape[1]
        dev = voxels.device
        dwt = torch.empty((9, c), dtype=torch.float32, device=dev)
        dga = torch.empty((c,), dtype=torch.float32, device=dev)
        dbe = torch.empty((c,), dtype=torch.float32, device=dev)
        l = rt.lib()
        ws = rt.workspace(l.sec_pfn_train_workspace_bytes(p, c), dev)
        g = grad_out.float().contiguous()
        vx, vy, xo, yo = ctx.geom
        rc = l.sec_pfn_train_bwd(rt.ptr(voxels), rt.ptr(num_points), rt.ptr(coords), p, t, f, rt.ptr(wt), rt.ptr(ga), rt.ptr(stats), c,
                                 vx, vy, xo, yo, rt.ptr(g), rt.ptr(out), rt.ptr(arg), rt.ptr(dwt), rt.ptr(dga), rt.ptr(dbe),
                                 rt.ptr(ws), ws.numel(), rt.stream())
        rt.check(rc, "sec_pfn_train_bwd")
        wd, gd, bd = ctx.dtypes
        return (None, None, None, dwt.t().contiguous().to(wd), dga.to(gd), dbe.to(bd), None, None, None, None, None)


@_traced("pfn_forward")
def pfn_forward_slots(points, vox, weight_t, scale, shift, vx, vy, x_offset, y_offset, out_dtype=None, num_dev=None):
    """:func:`pfn_forward` straight from the voxeliser's point lists: ``vox`` = the result of ``voxelize(points, ..., fill=False)``
    (or with fill), ``points`` the array it was called on.  Bit-identical to the tensor form; no [P, T, 4] tensor is read."""
    rt.require_gpu(points, weight_t, scale, shift)
    kind, ws, n, max_voxels, max_points, _ = vox["site_table"]
    assert kind == "vox" and points.shape[0] == n and points.dtype == torch.float32 and points.is_contiguous()
    npv, coords = vox["num_points_per_voxel"], vox["coordinates"].contiguous()
    p, c = coords.shape[0], weight_t.shape[1]
    batch = vox["voxel_offsets"].numel() - 1
    out_dtype = out_dtype or torch.float32
    out = torch.empty((p, c), dtype=out_dtype, device=points.device)
    rc = rt.lib().sec_pfn_fwd_slots(rt.ptr(points), rt.ptr(ws), ws.numel(), n, batch, max_voxels, max_points, points.shape[1],
                                    rt.ptr(npv), rt.ptr(coords), p, rt.ptr(num_dev), rt.ptr(weight_t), rt.ptr(scale), rt.ptr(shift), c,
                                    float(vx), float(vy), float(x_offset), float(y_offset), rt.ptr(out), rt.dtype_code(out_dtype),
                                    rt.stream())
    rt.check(rc, "sec_pfn_fwd_slots")
    return out


@_traced("voxel_block_filter")
def voxel_block_filter(vox, grid_size_xy, block_factor, block_size, height_threshold, height_high_threshold=3.0,
                       sync=True):
    """Block filtering of points_to_voxel_3d_with_filtering on the output dict of :func:`voxelize`
    (which must have been produced with sync=False or sync=True, any); returns a dict of the same layout."""
    voxels, coors, npv, voff = vox["voxels"], vox["coordinates"], vox["num_points_per_voxel"], vox["voxel_offsets"]
    rt.require_gpu(voxels, coors, npv, voff)
    rows, t, f = voxels.shape
    batch = voff.numel() - 1
    dev = voxels.device
    ov, oc, on = torch.empty_like(voxels), torch.empty_like(coors), torch.empty_like(npv)
    ooff = torch.empty_like(voff)
    l = rt.lib()
    ws = rt.workspace(l.sec_block_filter_workspace_bytes(rows, batch, int(grid_size_xy[0]), int(grid_size_xy[1]),
                                                         int(block_factor)), dev)
    rc = l.sec_voxel_block_filter_f32(rt.ptr(voxels), rt.ptr(coors), rt.ptr(npv), rt.ptr(voff), rows, batch, t, f,
                                      int(grid_size_xy[0]), int(grid_size_xy[1]), int(block_factor), int(block_size),
                                      float(height_threshold), float(height_high_threshold), rt.ptr(ov), rt.ptr(oc),
                                      rt.ptr(on), rt.ptr(ooff), rt.ptr(ws), ws.numel(), rt.stream())
    rt.check(rc, "sec_voxel_block_filter_f32")
    out = {"voxels": ov, "coordinates": oc, "num_points_per_voxel": on, "voxel_offsets": ooff}
    if sync:
        total = int(ooff[-1].item())
        for k in ("voxels", "coordinates", "num_points_per_voxel"):
            out[k] = out[k][:total]
        out["voxel_num"] = total
    return out


def bias_act_(x, bias, relu=True):
    """In place y = relu?(x + bias[c]) on a channels-last activation [B,C,H,W] (or [P,C]); bias float32 [C].
    The folded-BatchNorm bias + ReLU that follows every RPN conv (rpn.py:486-497), one pass."""
    rt.require_gpu(x, bias)
    assert bias.dtype == torch.float32 and bias.is_contiguous()
    if x.dim() == 4:
        assert x.is_contiguous(memory_format=torch.channels_last), "bias_act_ expects channels_last"
        c = x.shape[1]
    else:
        assert x.is_contiguous()
        c = x.shape[-1]
    assert bias.numel() == c
    rc = rt.lib().sec_bias_act_nhwc(rt.ptr(x), rt.ptr(bias), x.numel() // c, c, int(bool(relu)), rt.dtype_code(x.dtype),
                                    rt.stream())
    rt.check(rc, "sec_bias_act_nhwc")
    return x


def conv2d_pack_weight(weight):
    """[Cout,Cin,k,k] (torch layout, bf16/f16) -> MFMA slab order for :func:`conv2d_nhwc`; None if unsupported."""
    rt.require_gpu(weight)
    cout, cin, kh, kw = weight.shape
    if weight.dtype == torch.float32 or kh != kw:
        return None
    l = rt.lib()
    nbytes = l.sec_conv2d_packed_weight_bytes(cout, cin, kh, rt.dtype_code(weight.dtype))
    if nbytes == 0:
        return None
    packed = torch.empty((nbytes // 2,), dtype=weight.dtype, device=weight.device)
    rt.check(l.sec_conv2d_pack_weight(rt.ptr(weight.contiguous()), cout, cin, kh, rt.dtype_code(weight.dtype), rt.ptr(packed),
                                      rt.stream()), "sec_conv2d_pack_weight")
    return packed


@_traced("conv2d_nhwc")
def conv2d_nhwc(x, packed, bias, cout, ksize, stride=1, pad=0, relu=False, sparse_input=False):
    """Dense conv2d + bias + ReLU in one launch (hand-written MFMA implicit GEMM).  x: [B,Cin,H,W] in
    torch.channels_last memory format (bf16/f16); returns [B,Cout,Ho,Wo] channels_last.
    ``sparse_input``: x is a scattered sparse tensor (mostly zero tiles), which the 3x3/s1 kernel may skip."""
    rt.require_gpu(x, packed)
    assert x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
    b, cin, h, w = x.shape
    ho, wo = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    y = torch.empty((b, cout, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    rc = rt.lib().sec_conv2d_nhwc(rt.ptr(x), b, h, w, cin, rt.ptr(packed), rt.ptr(bias), cout, ksize, stride, pad,
                                  int(bool(relu)) | (2 if sparse_input else 0), rt.ptr(y), rt.dtype_code(x.dtype),
                                  rt.stream())
    rt.check(rc, "sec_conv2d_nhwc")
    return y


def conv2d_pack_weight_x3(weight):
    """fp32 [Cout, 128, 3, 3] -> the packed (hi | lo) bf16 weight pair of :func:`conv2d_nhwc_x3`: W = bf16(W) + bf16(W - bf16(W))."""
    rt.require_gpu(weight)
    w = weight.detach().float().contiguous()
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    n = w.numel()
    ph, pl = conv2d_pack_weight(hi), conv2d_pack_weight(lo)
    if ph is None or pl is None:
        return None
    return torch.cat([ph[:n], pl[:n], torch.zeros(8, dtype=torch.bfloat16, device=w.device)])


@_traced("split_bf16x2")
def split_bf16x2(x):
    """fp32 tensor (any layout, element count a multiple of 4) -> (hi, lo) bf16 tensors of the same shape / strides with
    hi = bf16(x), lo = bf16(x - hi): the operand form of :func:`conv2d_nhwc_x3` (16 significant bits)."""
    rt.require_gpu(x)
    assert x.dtype == torch.float32
    hi, lo = torch.empty_like(x, dtype=torch.bfloat16), torch.empty_like(x, dtype=torch.bfloat16)
    assert hi.stride() == x.stride()
    rt.check(rt.lib().sec_split_f32_bf16x2(rt.ptr(x), x.numel(), rt.ptr(hi), rt.ptr(lo), rt.stream()), "sec_split_f32_bf16x2")
    return hi, lo


@_traced("merge_bf16x2")
def merge_bf16x2(hi, lo):
    """(hi, lo) bf16 -> fp32 hi + lo (same shape / strides)."""
    rt.require_gpu(hi, lo)
    assert hi.dtype == lo.dtype == torch.bfloat16 and hi.stride() == lo.stride() and hi.shape == lo.shape
    y = torch.empty_like(hi, dtype=torch.float32)
    assert y.stride() == hi.stride()
    rt.check(rt.lib().sec_merge_bf16x2_f32(rt.ptr(hi), rt.ptr(lo), hi.numel(), rt.ptr(y), rt.stream()), "sec_merge_bf16x2_f32")
    return y


@_traced("conv2d_nhwc_x3")
def conv2d_nhwc_x3(x_hi, x_lo, packed_hi_lo, bias, cout, relu=True, sparse_input=False):
    """3x3 / stride 1 / pad 1 conv on 128 input channels for fp32 networks, operands as bf16 (hi, lo) plane pairs, three bf16
    MFMA passes accumulated in fp32 (sec_conv2d_nhwc_x3).  x_hi / x_lo [B,128,H,W] channels_last bf16 -> (y_hi, y_lo)."""
    rt.require_gpu(x_hi, x_lo, packed_hi_lo)
    assert x_hi.dim() == 4 and x_hi.shape[1] == 128 and x_hi.dtype == x_lo.dtype == torch.bfloat16 and x_hi.shape == x_lo.shape
    assert x_hi.is_contiguous(memory_format=torch.channels_last) and x_lo.is_contiguous(memory_format=torch.channels_last)
    b, _, h, w = x_hi.shape
    y_hi = torch.empty((b, cout, h, w), dtype=torch.bfloat16, device=x_hi.device, memory_format=torch.channels_last)
    y_lo = torch.empty_like(y_hi)
    rc = rt.lib().sec_conv2d_nhwc_x3(rt.ptr(x_hi), rt.ptr(x_lo), b, h, w, rt.ptr(packed_hi_lo), rt.ptr(bias), cout,
                                     int(bool(relu)) | (2 if sparse_input else 0), rt.ptr(y_hi), rt.ptr(y_lo), rt.stream())
    rt.check(rc, "sec_conv2d_nhwc_x3")
    return y_hi, y_lo


@_traced("conv1x1_chain_x3")
def conv1x1_chain_x3(x_hi, x_lo, packed_w1, bias1, packed_w2, bias2, cout2, relu1=True):
    """y = W2 * act(W1 * x + bias1) + bias2 on fp32 values carried as (hi, lo) bf16 planes (sec_conv1x1_chain_x3): x_hi / x_lo
    [B,128,H,W] channels_last bf16 -> fp32 [B, cout2, H, W] channels_last.  Weights: :func:`conv2d_pack_weight_x3` of the [C,128,1,1]
    tensors (cout2 padded to 64 / 128)."""
    rt.require_gpu(x_hi, x_lo, packed_w1, packed_w2)
    assert x_hi.dim() == 4 and x_hi.shape[1] == 128 and x_hi.dtype == x_lo.dtype == torch.bfloat16 and x_hi.shape == x_lo.shape
    assert x_hi.is_contiguous(memory_format=torch.channels_last) and x_lo.is_contiguous(memory_format=torch.channels_last)
    b, _, h, w = x_hi.shape
    y = torch.empty((b, int(cout2), h, w), dtype=torch.float32, device=x_hi.device, memory_format=torch.channels_last)
    rc = rt.lib().sec_conv1x1_chain_x3(rt.ptr(x_hi), rt.ptr(x_lo), b * h * w, rt.ptr(packed_w1), rt.ptr(bias1), int(bool(relu1)),
                                       rt.ptr(packed_w2), rt.ptr(bias2), int(cout2), rt.ptr(y), rt.stream())
    rt.check(rc, "sec_conv1x1_chain_x3")
    return y


@_traced("conv2d_nhwc_x3_tiles")
def conv2d_nhwc_x3_tiles(x_hi, x_lo, packed_hi_lo, bias, cout, tile_order, live_counts, background=None, relu=True, nbr_masks=None,
                         background_in=None):
    """:func:`conv2d_nhwc_x3` on the live tiles of one layer of :func:`rpn_tile_live` only (sec_conv2d_nhwc_x3_tiles; the fp32
    counterpart of :func:`conv2d_nhwc_tiles`).  ``background`` = (hi, lo) planes of this layer's output for an EMPTY frame, copied into
    the other tiles (None: they stay unwritten, for a lazy consumer); ``nbr_masks`` + ``background_in`` = (hi, lo) of the producing
    layer's empty-frame output: halo pixels of tiles the producer did not write are read from it."""
    rt.require_gpu(x_hi, x_lo, packed_hi_lo, tile_order, live_counts)
    assert x_hi.dim() == 4 and x_hi.shape[1] == 128 and x_hi.dtype == x_lo.dtype == torch.bfloat16 and x_hi.shape == x_lo.shape
    assert x_hi.is_contiguous(memory_format=torch.channels_last) and x_lo.is_contiguous(memory_format=torch.channels_last)
    b, _, h, w = x_hi.shape
    tiles = ((h + 7) // 8) * ((w + 15) // 16)
    assert tile_order.dtype == torch.int16 and tile_order.is_contiguous() and tuple(tile_order.shape) == (b, tiles)
    assert live_counts.dtype == torch.int32 and live_counts.is_contiguous() and live_counts.numel() == b
    for pair, ch in ((background, int(cout)), (background_in, 128)):
        if pair is not None:
            for t in pair:
                rt.require_gpu(t)
                assert t.dtype == torch.bfloat16 and t.numel() == h * w * ch and t.is_contiguous(memory_format=torch.channels_last)
    if nbr_masks is not None:
        assert background_in is not None and nbr_masks.dtype == torch.int16 and nbr_masks.is_contiguous() and tuple(nbr_masks.shape) == (2, b, tiles)
    y_hi = torch.empty((b, int(cout), h, w), dtype=torch.bfloat16, device=x_hi.device, memory_format=torch.channels_last)
    y_lo = torch.empty_like(y_hi)
    if POISON_LAZY_OUTPUTS and background is None:
        y_hi.fill_(float("nan"))
        y_lo.fill_(float("nan"))
    bg, bgi = background or (None, None), background_in or (None, None)
    rc = rt.lib().sec_conv2d_nhwc_x3_tiles(rt.ptr(x_hi), rt.ptr(x_lo), b, h, w, rt.ptr(packed_hi_lo), rt.ptr(bias), int(cout), int(bool(relu)),
                                           rt.ptr(tile_order), rt.ptr(live_counts), rt.ptr(bg[0]), rt.ptr(bg[1]),
                                           rt.ptr(nbr_masks) if nbr_masks is not None else None, rt.ptr(bgi[0]), rt.ptr(bgi[1]),
                                           rt.ptr(y_hi), rt.ptr(y_lo), rt.stream())
    rt.check(rc, "sec_conv2d_nhwc_x3_tiles")
    return y_hi, y_lo


@_traced("sparse_site_map")
def sparse_site_map(indices, batch_size, spatial_shape, num_dev=None):
    """[B, D, H, W] int32 map of a sparse tensor's sites: row + 1, 0 = no active site (input of :func:`conv2d_nhwc_gather`)."""
    rt.require_gpu(indices)
    d, h, w = [int(v) for v in spatial_shape]
    m = torch.empty((int(batch_size), d, h, w), dtype=torch.int32, device=indices.device)
    rc = rt.lib().sec_sparse_site_map(rt.ptr(indices.contiguous()), indices.shape[0], rt.ptr(num_dev), int(batch_size), d, h, w,
                                      rt.ptr(m), rt.stream())
    rt.check(rc, "sec_sparse_site_map")
    return m


@_traced("sparse_site_map")
def sparse_site_map_sorted(conv_workspace, rows_cap, batch_size, spatial_shape, num_dev=None):
    """:func:`sparse_site_map` for the outputs of a sorted-numbering strided build, read off that build's bitmap
    (``conv_workspace`` = the ``site_table`` workspace of its :func:`rulebook_conv` result): one launch instead of fill + scatter."""
    rt.require_gpu(conv_workspace)
    d, h, w = [int(v) for v in spatial_shape]
    m = torch.empty((int(batch_size), d, h, w), dtype=torch.int32, device=conv_workspace.device)
    rc = rt.lib().sec_sparse_site_map_sorted(rt.ptr(conv_workspace), conv_workspace.numel(), rt.ptr(num_dev), int(rows_cap),
                                             int(batch_size), d, h, w, rt.ptr(m), rt.stream())
    rt.check(rc, "sec_sparse_site_map_sorted")
    return m


def gather_channel_perm(c, d):
    """perm[z * c + ch] = ch * d + z: the reference's channel ch * D + z of ``dense().view(N, C * D, H, W)`` in the order the
    gather convolution reads it (plane-major).  ``w[:, perm]`` are the weights to pack for :func:`conv2d_nhwc_gather`."""
    j = torch.arange(c * d)
    return (j % c) * d + j // c


POISON_LAZY_OUTPUTS = False    # tests: the tiles a lazy conv leaves unwritten are filled with NaN first (a wrongful read reaches the output)


@_traced("conv2d_nhwc_gather")
def conv2d_nhwc_gather(features, site_map, packed, bias, cout, relu=True, tile_order=None, live_counts=None, background=None):
    """3x3 / stride 1 / pad 1 conv + bias + ReLU of ``SparseConvTensor.dense().view(B, 128, H, W)`` read straight from the
    sparse tensor's rows ``features`` [rows, 64] through ``site_map`` [B, 2, H, W] (:func:`sparse_site_map`): no dense image,
    tiles without sites cost a map lookup.  ``packed`` = conv2d_pack_weight(w[:, gather_channel_perm(64, 2)])."""
    rt.require_gpu(features, site_map, packed)
    assert features.dim() == 2 and features.shape[1] == 64 and features.is_contiguous() and site_map.dtype == torch.int32
    b, d, h, w = site_map.shape
    assert d == 2 and site_map.is_contiguous()
    y = torch.empty((b, int(cout), h, w), dtype=features.dtype, device=features.device, memory_format=torch.channels_last)
    if POISON_LAZY_OUTPUTS and tile_order is not None and background is None:
        y.fill_(float("nan"))
    if tile_order is not None:      # live tiles from rpn_tile_live (layer 0): spread evenly over the XCDs, the others copied from `background` (the empty frame's output)
        rt.require_gpu(tile_order, live_counts)
        assert tile_order.dtype == torch.int16 and tile_order.is_contiguous() and tile_order.shape[0] == b
        assert live_counts.dtype == torch.int32 and live_counts.numel() == b
        if background is not None:   # None: the other tiles stay unwritten -- every consumer must be lazy (conv2d_nhwc_tiles(nbr_masks=...))
            rt.require_gpu(background)
            assert background.dtype == features.dtype
            assert background.numel() == h * w * int(cout) and background.is_contiguous(memory_format=torch.channels_last)
    rc = rt.lib().sec_conv2d_nhwc_gather(rt.ptr(features), features.shape[0], rt.ptr(site_map), b, h, w, rt.ptr(packed), rt.ptr(bias),
                                         int(cout), int(bool(relu)), rt.ptr(tile_order) if tile_order is not None else None,
                                         rt.ptr(live_counts) if tile_order is not None else None,
                                         rt.ptr(background) if (tile_order is not None and background is not None) else None, rt.ptr(y),
                                         rt.dtype_code(features.dtype), rt.stream())
    rt.check(rc, "sec_conv2d_nhwc_gather")
    return y


@_traced("pillar_site_map")
def pillar_site_map(coords, batch_size, ny, nx, num_dev=None):
    """[B, ny, nx] int32, row + 1 of the pillar in cell (y, x) of frame b, 0 = none (sec_sparse_site_map on the one-plane grid: a
    pillar's z is 0, pointpillars.py:462) -- 2.5 MB for config 4 instead of the 82 MB feature canvas."""
    m = sparse_site_map.__wrapped_op__(coords, batch_size, (1, int(ny), int(nx)), num_dev=num_dev)
    return m.view(int(batch_size), int(ny), int(nx))


@_traced("conv2d_nhwc_rows")
def conv2d_nhwc_rows(rows, site_map, packed, bias, cout, ksize, stride, pad, relu=True):
    """``conv2d_nhwc(pillar_scatter(rows, coords), ...)`` without the image (sec_conv2d_nhwc_rows): ``rows`` [P, cin] 16-bit pillar features,
    ``site_map`` [B, ny, nx] from :func:`pillar_site_map`.  Bit-identical to the scattered form."""
    rt.require_gpu(rows, site_map, packed)
    assert rows.dim() == 2 and rows.is_contiguous() and site_map.dtype == torch.int32 and site_map.is_contiguous() and site_map.dim() == 3
    b, h, w = site_map.shape
    ho, wo = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    y = torch.empty((b, int(cout), ho, wo), dtype=rows.dtype, device=rows.device, memory_format=torch.channels_last)
    rc = rt.lib().sec_conv2d_nhwc_rows(rt.ptr(rows), rows.shape[0], rt.ptr(site_map), b, h, w, rows.shape[1], rt.ptr(packed), rt.ptr(bias),
                                       int(cout), int(ksize), int(stride), int(pad), int(bool(relu)), rt.ptr(y), rt.dtype_code(rows.dtype),
                                       rt.stream())
    rt.check(rc, "sec_conv2d_nhwc_rows")
    return y


def conv2d_into_supported(cin, cout, ksize, stride, pad, dtype):
    """Shapes :func:`conv2d_nhwc_into` serves (the strided / patch conv kernel, csrc/dense_patch.hpp patch::dispatch)."""
    if dtype not in (torch.bfloat16, torch.float16):
        return False
    c128 = cout % 128 == 0
    return ((ksize, stride, pad) == (3, 2, 1) and ((cin == 64 and (cout == 64 or c128)) or (cin == 128 and c128))
            or (ksize, stride, pad, cin) == (4, 4, 0, 64) and c128 or (ksize, stride, pad, cin) == (2, 2, 0, 128) and c128
            or (ksize, stride, pad) == (1, 1, 0) and cin in (256, 384) and c128)


@_traced("conv2d_nhwc")
def conv2d_nhwc_into(x, packed, bias, cout, ksize, stride, pad, relu, out, channel_offset):
    """:func:`conv2d_nhwc` writing channels [channel_offset, channel_offset + cout) of the channels_last map ``out`` [B, C, Ho, Wo]
    (sec_conv2d_nhwc_into): the deblocks of a multi-block RPN fill the concatenated map the heads read without a ``torch.cat``.
    Returns the channel slice (a view)."""
    rt.require_gpu(x, packed, out)
    assert x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and out.is_contiguous(memory_format=torch.channels_last)
    b, cin, h, w = x.shape
    ho, wo = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    assert out.dtype == x.dtype and out.shape[0] == b and tuple(out.shape[2:]) == (ho, wo) and channel_offset + cout <= out.shape[1]
    rc = rt.lib().sec_conv2d_nhwc_into(rt.ptr(x), b, h, w, cin, rt.ptr(packed), rt.ptr(bias), int(cout), int(ksize), int(stride), int(pad),
                                       int(bool(relu)), rt.ptr(out), int(out.shape[1]), int(channel_offset), rt.dtype_code(x.dtype), rt.stream())
    rt.check(rc, "sec_conv2d_nhwc_into")
    return out[:, channel_offset:channel_offset + cout]


def conv2d_rows_supported(cin, cout, ksize, stride, pad, dtype):
    return (dtype in (torch.bfloat16, torch.float16) and cin == 64 and ksize == 3 and stride == 2 and pad == 1
            and (cout == 64 or cout % 128 == 0))


@_traced("rpn_tile_live")
def rpn_tile_live(site_map, layers, masks=False):
    """order [layers, B, tiles] int16 + counts [layers, B] int32 for the first ``layers`` 3x3 convs of the RPN (layer 0 = the gathered
    one): per frame the 8 x 16 tiles (row-major indices) that can differ from the layer's background first, the others from the
    end backwards; counts = how many can differ (sec_rpn_tile_live).  ``masks``: also nbr_masks [layers, 2, B, tiles] int16 for the
    lazy consumers (sec_rpn_tile_live_masks: which of a tile's 3 x 3 neighbours the previous conv really wrote)."""
    rt.require_gpu(site_map)
    assert site_map.dtype == torch.int32 and site_map.dim() == 4 and site_map.shape[1] == 2 and site_map.is_contiguous()
    b, _, h, w = site_map.shape
    tiles = ((h + 7) // 8) * ((w + 15) // 16)
    assert tiles < 32768
    order = torch.empty((int(layers), b, tiles), dtype=torch.int16, device=site_map.device)
    counts = torch.empty((int(layers), b), dtype=torch.int32, device=site_map.device)
    ws_bytes = rt.lib().sec_rpn_tile_live_workspace_bytes(b, h, w)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=site_map.device)
    if masks:
        nbr = torch.empty((int(layers), 2, b, tiles), dtype=torch.int16, device=site_map.device)
        rt.check(rt.lib().sec_rpn_tile_live_masks(rt.ptr(site_map), b, h, w, int(layers), rt.ptr(order), rt.ptr(counts), rt.ptr(nbr), rt.ptr(ws),
                                                  ws_bytes, rt.stream()), "sec_rpn_tile_live_masks")
        return order, counts, nbr
    rt.check(rt.lib().sec_rpn_tile_live(rt.ptr(site_map), b, h, w, int(layers), rt.ptr(order), rt.ptr(counts), rt.ptr(ws), ws_bytes,
                                        rt.stream()), "sec_rpn_tile_live")
    return order, counts


@_traced("conv2d_nhwc_tiles")
def conv2d_nhwc_tiles(x, packed, bias, cout, tile_order, live_counts, background, relu=True, nbr_masks=None, background_in=None):
    """3x3 / stride 1 / pad 1 conv + bias + ReLU on a channels_last [B,128,H,W] tensor; only the live tiles of ``tile_order`` [B, tiles] /
    ``live_counts`` [B] (one layer of :func:`rpn_tile_live`) are convolved, the others are copied from ``background`` = this layer's
    output for an EMPTY frame, channels_last [1, cout, H, W] (sec_conv2d_nhwc_tiles).
    LAZY form (``nbr_masks`` [2, B, tiles] = this layer's slice of rpn_tile_live(masks=True), ``background_in`` = the PRODUCING layer's
    empty-frame output): ``x`` holds only the tiles its producer found live, the halo pixels of the others are read from
    ``background_in``; ``background`` may then be None -- this layer writes its live tiles only (sec_conv2d_nhwc_tiles_lazy)."""
    rt.require_gpu(x, packed, tile_order, live_counts)
    assert x.dim() == 4 and x.shape[1] == 128 and x.is_contiguous(memory_format=torch.channels_last)
    b, _, h, w = x.shape
    tiles = ((h + 7) // 8) * ((w + 15) // 16)
    assert tile_order.dtype == torch.int16 and tile_order.is_contiguous() and tuple(tile_order.shape) == (b, tiles)
    assert live_counts.dtype == torch.int32 and live_counts.is_contiguous() and live_counts.numel() == b
    if background is not None:
        rt.require_gpu(background)
        assert background.dtype == x.dtype and background.numel() == h * w * int(cout) and background.is_contiguous(memory_format=torch.channels_last)
    y = torch.empty((b, int(cout), h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if POISON_LAZY_OUTPUTS and background is None:
        y.fill_(float("nan"))
    if nbr_masks is not None:
        rt.require_gpu(nbr_masks, background_in)
        assert nbr_masks.dtype == torch.int16 and nbr_masks.is_contiguous() and tuple(nbr_masks.shape) == (2, b, tiles)
        assert background_in.dtype == x.dtype and background_in.numel() == h * w * 128 and background_in.is_contiguous(memory_format=torch.channels_last)
        rc = rt.lib().sec_conv2d_nhwc_tiles_lazy(rt.ptr(x), b, h, w, rt.ptr(packed), rt.ptr(bias), int(cout), int(bool(relu)),
                                                 rt.ptr(tile_order), rt.ptr(live_counts), rt.ptr(background) if background is not None else None,
                                                 rt.ptr(nbr_masks), rt.ptr(background_in), rt.ptr(y), rt.dtype_code(x.dtype), rt.stream())
        rt.check(rc, "sec_conv2d_nhwc_tiles_lazy")
        return y
    assert background is not None, "conv2d_nhwc_tiles: without nbr_masks the background tiles must be copied"
    rc = rt.lib().sec_conv2d_nhwc_tiles(rt.ptr(x), b, h, w, rt.ptr(packed), rt.ptr(bias), int(cout), int(bool(relu)),
                                        rt.ptr(tile_order), rt.ptr(live_counts), rt.ptr(background), rt.ptr(y),
                                        rt.dtype_code(x.dtype), rt.stream())
    rt.check(rc, "sec_conv2d_nhwc_tiles")
    return y


@_traced("conv2d_nhwc_tiles_tail")
def conv2d_nhwc_tiles_tail(x, packed, bias, tile_order, live_counts, nbr_masks, background_in, packed_w1, bias1, packed_w2, bias2, cout2,
                           relu=True, relu1=True):
    """The lazy form of :func:`conv2d_nhwc_tiles` (128 -> 128) with the fused 1x1 tail of :func:`conv1x1_chain` (``x_live_only``, no
    background) in its epilogue -- one launch, the conv's output never reaches memory (sec_conv2d_nhwc_tiles_tail).  Returns the head
    tensor [B, cout2 = 64, H, W] channels_last; tiles outside the list are unwritten unless the conv took the plain tile order."""
    rt.require_gpu(x, packed, tile_order, live_counts, nbr_masks, background_in, packed_w1, packed_w2, bias1)
    assert x.dim() == 4 and x.shape[1] == 128 and x.is_contiguous(memory_format=torch.channels_last) and int(cout2) == 64
    b, _, h, w = x.shape
    tiles = ((h + 7) // 8) * ((w + 15) // 16)
    assert tile_order.dtype == torch.int16 and tile_order.is_contiguous() and tuple(tile_order.shape) == (b, tiles)
    assert live_counts.dtype == torch.int32 and live_counts.is_contiguous() and live_counts.numel() == b
    assert nbr_masks.dtype == torch.int16 and nbr_masks.is_contiguous() and tuple(nbr_masks.shape) == (2, b, tiles)
    assert background_in.dtype == x.dtype and background_in.numel() == h * w * 128 and background_in.is_contiguous(memory_format=torch.channels_last)
    y = torch.empty((b, 64, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if POISON_LAZY_OUTPUTS:
        y.fill_(float("nan"))
    rc = rt.lib().sec_conv2d_nhwc_tiles_tail(rt.ptr(x), b, h, w, rt.ptr(packed), rt.ptr(bias), int(bool(relu)), rt.ptr(tile_order), rt.ptr(live_counts),
                                             rt.ptr(nbr_masks), rt.ptr(background_in), rt.ptr(packed_w1), rt.ptr(bias1), int(bool(relu1)),
                                             rt.ptr(packed_w2), rt.ptr(bias2) if bias2 is not None else None, 64, rt.ptr(y),
                                             rt.dtype_code(x.dtype), rt.stream())
    rt.check(rc, "sec_conv2d_nhwc_tiles_tail")
    return y


# ----------------------------------------------------------------------------- IoU / NMS
@_traced("conv1x1_chain")
def conv1x1_chain(x, packed_w1, bias1, packed_w2, bias2, cout2, relu1=True, tile_order=None, live_counts=None, background=None, x_live_only=False):
    """y = W2 * act(W1 * x + bias1) + bias2 for two back-to-back 1x1 convs on a channels_last [B,128,H,W] tensor
    (the RPN deblock + merged heads, rpn.py:275-285,386-391); the 128-channel intermediate stays in LDS.
    ``tile_order`` / ``live_counts`` (the last 3x3 conv's lists of :func:`rpn_tile_live`) + ``background`` (this op's output for an
    empty frame, [1, cout2, H, W]): only the live tiles are computed, the others copied (sec_conv1x1_chain_nhwc_tiles);
    ``background=None`` (with ``x_live_only``): the others are not written at all -- 22 MB of copies per batch of 8 less -- and the
    consumers read them from the empty frame's map (``lazy=`` of :func:`predict_select` / :func:`predict_decode`)."""
    rt.require_gpu(x, packed_w1, packed_w2, bias1)
    assert x.dim() == 4 and x.shape[1] == 128 and x.is_contiguous(memory_format=torch.channels_last)
    b, _, h, w = x.shape
    y = torch.empty((b, int(cout2), h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if tile_order is not None:
        rt.require_gpu(tile_order, live_counts)
        assert tile_order.dtype == torch.int16 and tile_order.is_contiguous() and tuple(tile_order.shape) == (b, ((h + 7) // 8) * ((w + 15) // 16))
        assert live_counts.dtype == torch.int32 and live_counts.numel() == b
        if background is None:
            # lazy consumers (predict_select / predict_decode with ``lazy=``): the tiles outside the list stay UNWRITTEN
            assert x_live_only, "conv1x1_chain without a background: the lists must be binding (x_live_only)"
            if POISON_LAZY_OUTPUTS:
                y.fill_(float("nan"))
        else:
            rt.require_gpu(background)
            assert background.dtype == x.dtype and background.numel() == h * w * int(cout2) and background.is_contiguous(memory_format=torch.channels_last)
        # x_live_only (SEC_CHAIN_X_LIVE_ONLY): the producer of x wrote the live tiles of these lists only -- the lists are used whatever the live share
        rc = rt.lib().sec_conv1x1_chain_nhwc_tiles(rt.ptr(x), b, h, w, rt.ptr(packed_w1), rt.ptr(bias1), int(bool(relu1)) | (2 if x_live_only else 0), rt.ptr(packed_w2),
                                                   rt.ptr(bias2), int(cout2), rt.ptr(tile_order), rt.ptr(live_counts), rt.ptr(background),
                                                   rt.ptr(y), rt.dtype_code(x.dtype), rt.stream())
        rt.check(rc, "sec_conv1x1_chain_nhwc_tiles")
        return y
    rc = rt.lib().sec_conv1x1_chain_nhwc(rt.ptr(x), b * h * w, rt.ptr(packed_w1), rt.ptr(bias1), int(bool(relu1)),
                                         rt.ptr(packed_w2), rt.ptr(bias2), int(cout2), rt.ptr(y), rt.dtype_code(x.dtype),
                                         rt.stream())
    rt.check(rc, "sec_conv1x1_chain_nhwc")
    return y


def rotate_iou(boxes, qboxes, criterion=-1):
    """[N,K] rotated IoU (nms_gpu.py rotate_iou_gpu_eval)."""
    rt.require_gpu(boxes, qboxes)
    boxes, qboxes = boxes.float().contiguous(), qboxes.float().contiguous()
    n, k = boxes.shape[0], qboxes.shape[0]
    out = torch.zeros((n, k), dtype=torch.float32, device=boxes.device)
    rt.check(rt.lib().sec_rotate_iou_f32(rt.ptr(boxes), n, rt.ptr(qboxes), k, int(criterion), rt.ptr(out), rt.stream()),
             "sec_rotate_iou_f32")
    return out


@_traced("nms_sorted")
def nms_sorted(dets, counts, thresh, kind="rotate", semantics="numba", eps=1.0, post_max=0, exact_clip=False):
    """Greedy NMS of score-sorted boxes. dets [B,max_n,stride] float32, counts [B] int32 (device).
    Returns (keep [B,max_n] int32 positions, num_keep [B] int32), all on device, no host sync.
    ``exact_clip``: SEC_NMS_EXACT_CLIP -- no inscribed-circle shortcut, every overlapping pair runs the reference's clipper."""
    rt.require_gpu(dets, counts)
    assert dets.dtype == torch.float32 and dets.dim() == 3 and dets.is_contiguous() and counts.dtype == torch.int32
    b, max_n, stride = dets.shape
    keep = torch.empty((b, max_n), dtype=torch.int32, device=dets.device)
    num_keep = torch.empty((b,), dtype=torch.int32, device=dets.device)
    l = rt.lib()
    ws = rt.workspace(l.sec_nms_workspace_bytes(b, max_n), dets.device)
    rc = l.sec_nms_sorted_f32(rt.ptr(dets), rt.ptr(counts), b, max_n, stride, float(thresh),
                              {"rotate": 0, "axis_aligned": 1}[kind], {"numba": 0, "cpu": 1}[semantics] | (256 if exact_clip else 0), float(eps),
                              int(post_max or 0), rt.ptr(keep), rt.ptr(num_keep), rt.ptr(ws), ws.numel(), rt.stream())
    rt.check(rc, "sec_nms_sorted_f32")
    return keep, num_keep


# ----------------------------------------------------------------------------- fused predict (voxelnet.py:377-645)
def _strides5(t):
    import ctypes
    assert t.dim() == 5
    return (ctypes.c_int64 * 5)(*[int(x) for x in t.stride()])


def _lazy_args(view, lazy, pick):
    """(tile_live pointer, background pointer of the same view) of a ``lazy=(tile_live [B, tiles] int16, pick-able background)``."""
    tile_live, bg = lazy
    b, a, h, w, _ = view.shape
    assert tile_live.dtype == torch.int16 and tile_live.is_contiguous() and tuple(tile_live.shape) == (b, ((h + 7) // 8) * ((w + 15) // 16))
    bgv = pick(bg)
    assert bgv.dtype == view.dtype and tuple(bgv.shape[1:]) == tuple(view.shape[1:]) and tuple(bgv.stride()[1:]) == tuple(view.stride()[1:]), \
        "the background view must have the strides of the head view (same channels-last layout)"
    return tile_live, bgv


@_traced("predict_select")
def predict_select(cls, k, score_thr, lazy=None):
    """cls: [B, A, H, W, num_class] view (any strides).  -> (top_idx [B,k] int32 anchor ids sorted by descending
    score, top_score [B,k] sigmoid scores, top_label [B,k], counts [B] = entries with score >= score_thr).  The selection is
    rows [0, counts[b]) of frame b (the reference masks by the threshold before its topk); rows behind them are unspecified.
    ``lazy`` = (tile_live [B, tiles] int16 with bit 4 = "tile written", bg = the empty frame's view [1, A, H, W, num_class] with the
    strides of ``cls``): elements of unwritten 8 x 16 tiles are read from ``bg`` (sec_predict_select_lazy)."""
    rt.require_gpu(cls)
    b, a, h, w, nc = cls.shape
    k = min(int(k), a * h * w, 1024)
    dev = cls.device
    top_idx = torch.empty((b, k), dtype=torch.int32, device=dev)
    top_score = torch.empty((b, k), dtype=torch.float32, device=dev)
    top_label = torch.empty((b, k), dtype=torch.int32, device=dev)
    counts = torch.empty((b,), dtype=torch.int32, device=dev)
    keys = torch.empty((b * a * h * w,), dtype=torch.int32, device=dev)
    if lazy is not None:
        tile_live, bgv = _lazy_args(cls, lazy, lambda t: t)
        rc = rt.lib().sec_predict_select_lazy(rt.ptr(cls), _strides5(cls), b, a, h, w, nc, k, float(score_thr), rt.ptr(keys), rt.ptr(top_idx),
                                              rt.ptr(top_score), rt.ptr(top_label), rt.ptr(counts), rt.dtype_code(cls.dtype),
                                              rt.ptr(tile_live), rt.ptr(bgv), rt.stream())
        rt.check(rc, "sec_predict_select_lazy")
        return top_idx, top_score, top_label, counts
    rc = rt.lib().sec_predict_select(rt.ptr(cls), _strides5(cls), b, a, h, w, nc, k, float(score_thr), rt.ptr(keys), rt.ptr(top_idx),
                                     rt.ptr(top_score), rt.ptr(top_label), rt.ptr(counts), rt.dtype_code(cls.dtype), rt.stream())
    rt.check(rc, "sec_predict_select")
    return top_idx, top_score, top_label, counts


@_traced("predict_decode")
def predict_decode(box, dir_cls, anchors, top_idx, top_score, rotate=True, lazy=None):
    """box: [B,A,H,W,7] view, dir_cls: [B,A,H,W,bins] view or None, anchors [A*H*W,7] fp32.
    -> (decoded [B,k,7] fp32, dets [B,k,6] fp32 NMS rows, dir_label [B,k] int32).
    ``lazy`` = (tile_live, (bg_box, bg_dir)): as in :func:`predict_select` (sec_predict_decode_lazy)."""
    rt.require_gpu(box, anchors, top_idx, top_score)
    b, a, h, w, code = box.shape
    assert code == 7 and anchors.dtype == torch.float32 and anchors.is_contiguous() and anchors.shape == (a * h * w, 7)
    k = top_idx.shape[1]
    dev = box.device
    dec = torch.empty((b, k, 7), dtype=torch.float32, device=dev)
    dets = torch.empty((b, k, 6), dtype=torch.float32, device=dev)
    dlab = torch.empty((b, k), dtype=torch.int32, device=dev)
    if dir_cls is not None:
        assert dir_cls.dtype == box.dtype
    if lazy is not None:
        tile_live, bgb = _lazy_args(box, lazy, lambda t: t[0])
        bgd = _lazy_args(dir_cls, lazy, lambda t: t[1])[1] if dir_cls is not None else None
        rc = rt.lib().sec_predict_decode_lazy(rt.ptr(box), _strides5(box), rt.ptr(dir_cls), _strides5(dir_cls) if dir_cls is not None else None,
                                              dir_cls.shape[-1] if dir_cls is not None else 0, b, a, h, w, k, rt.ptr(anchors),
                                              rt.ptr(top_idx), rt.ptr(top_score), int(bool(rotate)), rt.ptr(dec), rt.ptr(dets), rt.ptr(dlab),
                                              rt.dtype_code(box.dtype), rt.ptr(tile_live), rt.ptr(bgb), rt.ptr(bgd), rt.stream())
        rt.check(rc, "sec_predict_decode_lazy")
        return dec, dets, dlab
    rc = rt.lib().sec_predict_decode(rt.ptr(box), _strides5(box), rt.ptr(dir_cls),
                                     _strides5(dir_cls) if dir_cls is not None else None,
                                     dir_cls.shape[-1] if dir_cls is not None else 0, b, a, h, w, k, rt.ptr(anchors),
                                     rt.ptr(top_idx), rt.ptr(top_score), int(bool(rotate)), rt.ptr(dec), rt.ptr(dets),
                                     rt.ptr(dlab), rt.dtype_code(box.dtype), rt.stream())
    rt.check(rc, "sec_predict_decode")
    return dec, dets, dlab


@_traced("predict_finalize")
def predict_finalize(dec, top_score, top_label, dir_label, keep, num_keep, post_max, use_direction, dir_offset,
                     dir_limit_offset, num_dir_bins, range6):
    """-> dict(boxes [B,P,7], scores [B,P], labels [B,P] int32, valid [B,P] bool)."""
    rt.require_gpu(dec, keep, num_keep)
    b, k, _ = dec.shape
    p = min(int(post_max), k)
    dev = dec.device
    boxes = torch.empty((b, p, 7), dtype=torch.float32, device=dev)
    scores = torch.empty((b, p), dtype=torch.float32, device=dev)
    labels = torch.empty((b, p), dtype=torch.int32, device=dev)
    valid = torch.empty((b, p), dtype=torch.uint8, device=dev)
    rc = rt.lib().sec_predict_finalize(rt.ptr(dec), rt.ptr(top_score), rt.ptr(top_label), rt.ptr(dir_label), rt.ptr(keep),
                                       rt.ptr(num_keep), b, k, p, int(bool(use_direction)), float(dir_offset),
                                       float(dir_limit_offset), int(num_dir_bins), rt.ptr(range6), rt.ptr(boxes),
                                       rt.ptr(scores), rt.ptr(labels), rt.ptr(valid), rt.stream())
    rt.check(rc, "sec_predict_finalize")
    # (the kernel writes 0 / 1 bytes: reinterpret, do not convert -- `.bool()` was the last torch kernel inside the captured step, 7.8 us)
    return {"boxes": boxes, "scores": scores, "labels": labels, "valid": valid.view(torch.bool)}


# ----------------------------------------------------------------------------- training: targets + loss
@_traced("assign_targets")
def assign_targets(anchors, gt_boxes, gt_offsets, matched_threshold, unmatched_threshold, gt_classes=None,
                   gt_importance=None):
    """TargetAssigner.assign -> create_target_np (second/core/target_ops.py:29-229) with NearestIouSimilarity and
    GroundBox3dCoder.encode, for a whole batch on the device.  anchors [A,7] fp32; gt_boxes [G,7] fp32 (frames concatenated),
    gt_offsets [B+1] int32.  -> labels [B,A] int32, bbox_targets [B,A,7] fp32, importance [B,A] fp32."""
    rt.require_gpu(anchors, gt_boxes, gt_offsets)
    assert anchors.dtype == torch.float32 and anchors.is_contiguous() and anchors.shape[1] == 7
    assert gt_offsets.dtype == torch.int32
    gt_boxes = gt_boxes.float().contiguous()
    a, b, g = anchors.shape[0], gt_offsets.numel() - 1, gt_boxes.shape[0]
    dev = anchors.device
    # the kernel reads raw int32 / float32 arrays: an int64 label tensor (torch.from_numpy's default) would be read as garbage
    if gt_classes is not None:
        rt.require_gpu(gt_classes)
        assert gt_classes.numel() == g and gt_classes.device == dev, "gt_classes: one class id per ground-truth box, on the anchors' device"
        gt_classes = gt_classes.to(torch.int32).contiguous()
    if gt_importance is not None:
        rt.require_gpu(gt_importance)
        assert gt_importance.numel() == g and gt_importance.device == dev, "gt_importance: one weight per ground-truth box, on the anchors' device"
        gt_importance = gt_importance.to(torch.float32).contiguous()
    labels = torch.empty((b, a), dtype=torch.int32, device=dev)
    targets = torch.empty((b, a, 7), dtype=torch.float32, device=dev)
    importance = torch.empty((b, a), dtype=torch.float32, device=dev)
    l = rt.lib()
    ws = rt.workspace(l.sec_assign_targets_workspace_bytes(b, a, g), dev)
    rc = l.sec_assign_targets_f32(rt.ptr(anchors), a, rt.ptr(gt_boxes) if g else None, rt.ptr(gt_classes), rt.ptr(gt_importance),
                                  rt.ptr(gt_offsets), g, b, float(matched_threshold), float(unmatched_threshold), rt.ptr(labels),
                                  rt.ptr(targets), rt.ptr(importance), rt.ptr(ws), ws.numel(), rt.stream())
    rt.check(rc, "sec_assign_targets_f32")
    return labels, targets, importance


@_traced("assign_targets_per_class")
def assign_targets_per_class(anchors, gt_boxes, gt_offsets, gt_classes, class_anchor_begin, class_ids, matched_thresholds,
                             unmatched_thresholds, gt_importance=None):
    """TargetAssigner.assign for multi-class configs, a whole batch on the device.  Range c of the class-major anchor array =
    anchors[class_anchor_begin[c]:class_anchor_begin[c+1]] with its own thresholds; class_ids[c] = k > 0: assign_per_class
    (target_assigner.py:90-160, only ground truth of class k); class_ids[c] = 0: assign_all with per-anchor thresholds
    (target_assigner.py:53-88).  Same outputs as assign_targets."""
    rt.require_gpu(anchors, gt_boxes, gt_offsets)
    assert anchors.dtype == torch.float32 and anchors.is_contiguous() and anchors.shape[1] == 7
    assert gt_offsets.dtype == torch.int32
    gt_boxes = gt_boxes.float().contiguous()
    a, b, g = anchors.shape[0], gt_offsets.numel() - 1, gt_boxes.shape[0]
    n = len(class_ids)
    assert len(class_anchor_begin) == n + 1 and len(matched_thresholds) == n and len(unmatched_thresholds) == n
    assert gt_classes is not None and gt_classes.dtype == torch.int32 and gt_classes.numel() == g
    dev = anchors.device
    labels = torch.empty((b, a), dtype=torch.int32, device=dev)
    targets = torch.empty((b, a, 7), dtype=torch.float32, device=dev)
    importance = torch.empty((b, a), dtype=torch.float32, device=dev)
    l = rt.lib()
    ws = rt.workspace(l.sec_assign_targets_workspace_bytes(b, a, g), dev)
    begin = (ctypes.c_int * (n + 1))(*[int(v) for v in class_anchor_begin])
    ids = (ctypes.c_int * n)(*[int(v) for v in class_ids])
    mt = (ctypes.c_float * n)(*[float(v) for v in matched_thresholds])
    ut = (ctypes.c_float * n)(*[float(v) for v in unmatched_thresholds])
    rc = l.sec_assign_targets_per_class_f32(rt.ptr(anchors), a, rt.ptr(gt_boxes) if g else None, rt.ptr(gt_classes),
                                            rt.ptr(gt_importance), rt.ptr(gt_offsets), g, b, n, begin, ids, mt, ut, rt.ptr(labels),
                                            rt.ptr(targets), rt.ptr(importance), rt.ptr(ws), ws.numel(), rt.stream())
    rt.check(rc, "sec_assign_targets_per_class_f32")
    return labels, targets, importance


LOSS_DEFAULTS = dict(alpha=0.25, gamma=2.0, sigma=3.0, pos_cls_weight=1.0, neg_cls_weight=1.0, classification_weight=1.0,
                     localization_weight=2.0, direction_loss_weight=0.2, direction_offset=0.0, sin_error_factor=1.0,
                     code_weights=(1.0,) * 7)   # second/configs/car.fhd.config:35-68


@_traced("second_loss")
def second_loss_raw(cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, importance, **cfg):
    """VoxelNet.loss (second/pytorch/models/voxelnet.py:239-312) values and head gradients in one fused pass.
    cls_preds [B,N,C], box_preds [B,N,7], dir_preds [B,N,bins] or None: contiguous fp32.  -> (out6, d_cls, d_box, d_dir) with
    out6 = (loss, cls_loss_reduced, loc_loss_reduced, dir_loss_reduced, cls_pos_loss, cls_neg_loss)."""
    rt.require_gpu(cls_preds, box_preds, labels, reg_targets, anchors, importance)
    p = dict(LOSS_DEFAULTS, **cfg)
    for t in (cls_preds, box_preds, reg_targets, anchors, importance) + ((dir_preds,) if dir_preds is not None else ()):
        assert t.dtype == torch.float32 and t.is_contiguous(), "second_loss takes contiguous fp32 tensors"
    assert labels.dtype == torch.int32 and labels.is_contiguous()
    b, n, nc = cls_preds.shape
    bins = dir_preds.shape[-1] if dir_preds is not None else 0
    dev = cls_preds.device
    d_cls, d_box = torch.empty_like(cls_preds), torch.empty_like(box_preds)
    d_dir = torch.empty_like(dir_preds) if dir_preds is not None else None
    out6 = torch.empty((6,), dtype=torch.float32, device=dev)
    params = rt.f_arr([p["alpha"], p["gamma"], p["sigma"], p["pos_cls_weight"], p["neg_cls_weight"], p["classification_weight"],
                       p["localization_weight"], p["direction_loss_weight"], p["direction_offset"], p["sin_error_factor"],
                       *p["code_weights"]])
    l = rt.lib()
    ws = rt.workspace(l.sec_second_loss_workspace_bytes(b, n), dev)
    rc = l.sec_second_loss_f32(rt.ptr(cls_preds), rt.ptr(box_preds), rt.ptr(dir_preds), rt.ptr(labels), rt.ptr(reg_targets),
                               rt.ptr(anchors), rt.ptr(importance), b, n, nc, bins, params, rt.ptr(d_cls), rt.ptr(d_box),
                               rt.ptr(d_dir), rt.ptr(out6), rt.ptr(ws), ws.numel(), rt.stream())
    rt.check(rc, "sec_second_loss_f32")
    return out6, d_cls, d_box, d_dir


class SecondLossFunction(torch.autograd.Function):
    """loss = SecondLossFunction.apply(cls, box, dir, labels, reg_targets, anchors, importance, cfg_dict): the scalar training
    loss of VoxelNet.loss; backward hands the gradients the forward pass already computed to autograd (scaled by grad_output)."""

    @staticmethod
    def forward(ctx, cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, importance, cfg):
        shapes = (cls_preds.shape, box_preds.shape, None if dir_preds is None else dir_preds.shape)
        b = cls_preds.shape[0]
        dts = (cls_preds.dtype, box_preds.dtype, None if dir_preds is None else dir_preds.dtype)
        f = lambda t, k: t.reshape(b, -1, k).float().contiguous()
        nc = cfg.get("num_class", 1)
        out6, d_cls, d_box, d_dir = second_loss_raw(f(cls_preds, nc), f(box_preds, 7),
                                                    None if dir_preds is None else f(dir_preds, cfg.get("num_direction_bins", 2)),
                                                    labels, reg_targets, anchors, importance,
                                                    **{k: v for k, v in cfg.items() if k in LOSS_DEFAULTS})
        ctx.save_for_backward(d_cls, d_box, d_dir if d_dir is not None else d_cls.new_zeros(0))
        ctx.meta = (shapes, dts)
        ctx.mark_non_differentiable(out6)
        return out6[0], out6

    @staticmethod
    def backward(ctx, g_loss, _g_all):
        d_cls, d_box, d_dir = ctx.saved_tensors
        shapes, dts = ctx.meta
        gd = None if shapes[2] is None else (d_dir * g_loss).reshape(shapes[2]).to(dts[2])
        return ((d_cls * g_loss).reshape(shapes[0]).to(dts[0]), (d_box * g_loss).reshape(shapes[1]).to(dts[1]), gd,
                None, None, None, None, None)


# ----------------------------------------------------------------------------- dense RPN training (rpn.py:486-497 under train.py:316-322)
def conv2d_dgrad_weight(weight):
    """[Cout,Cin,3,3] -> the weights of the convolution that maps dY to dX for stride 1 / pad 1: W'[ci][co][ky][kx] =
    W[co][ci][2-ky][2-kx] (the data gradient of a correlation is the correlation with the flipped, transposed kernel)."""
    return weight.flip(2, 3).transpose(0, 1).contiguous()


def conv2d_wgrad(x, dy, ksize=3):
    """dW [Cout,Cin,k,k] fp32 of y = conv2d(x, W, stride 1, padding k // 2), k = 3 or 1: x [B,Cin,H,W], dy [B,Cout,H,W], both
    channels_last 16-bit, Cin = Cout = 128 (sec_conv2d_wgrad_nhwc).  Deterministic."""
    rt.require_gpu(x, dy)
    assert x.dtype == dy.dtype and x.dim() == 4 and x.shape[0] == dy.shape[0] and x.shape[2:] == dy.shape[2:]
    assert x.is_contiguous(memory_format=torch.channels_last) and dy.is_contiguous(memory_format=torch.channels_last)
    b, cin, h, w = x.shape
    cout = dy.shape[1]
    l = rt.lib()
    k = int(ksize)
    nb = l.sec_conv2d_wgrad_workspace_bytes(b, h, w, cin, cout, k)
    if nb == 0:
        raise rt.SecondHipError(f"conv2d_wgrad: unsupported shape {cin}->{cout} k{k} (3x3 / s1 / p1 or 1x1 with 128 channels only)")
    ws = rt.workspace(nb, x.device)
    dw = torch.empty((cout, cin, k, k), dtype=torch.float32, device=x.device)
    rt.check(l.sec_conv2d_wgrad_nhwc(rt.ptr(x), rt.ptr(dy), b, h, w, cin, cout, k, 1, k // 2, rt.ptr(dw), rt.ptr(ws), ws.numel(),
                                     rt.dtype_code(x.dtype), rt.stream()), "sec_conv2d_wgrad_nhwc")
    return dw


def conv2d_wgrad_supported(cin, cout, ksize, stride, pad, dtype):
    return (cin, cout, ksize, stride, pad) == (128, 128, 3, 1, 1) and dtype in (torch.bfloat16, torch.float16)


def _bn_dims(y):
    if y.dim() == 2:
        assert y.is_contiguous()
        return y.shape[0], y.shape[1], 1, 1
    assert y.dim() == 4 and y.is_contiguous(memory_format=torch.channels_last)
    return tuple(y.shape)


def bn_train_supported(channels, dtype):
    return dtype in (torch.bfloat16, torch.float16) and channels % 8 == 0 and channels <= 256 and 256 % (channels // 8) == 0


def bn_relu_forward(y, gamma, beta, eps, momentum, running_mean=None, running_var=None, relu=True, rows_dev=None):
    """Training-mode BatchNorm + ReLU of a 16-bit activation whose channels are innermost in memory: a channels_last [B,C,H,W]
    tensor (BatchNorm2d of the RPN) or the [N,C] feature rows of a sparse tensor (BatchNorm1d of the sparse middle:
    middle.py:146-189) (sec_bn_relu_fwd_nhwc).  gamma / beta fp32 [C]; running_mean / running_var (fp32, updated in place) may be
    None.  -> (z, save_mean, save_invstd)."""
    rt.require_gpu(y, gamma, beta)
    b, c, h, w = _bn_dims(y)
    for t in (gamma, beta, running_mean, running_var):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == c and t.is_cuda)
    l = rt.lib()
    nb = l.sec_bn_train_workspace_bytes(c)
    if nb == 0 or 256 % (c // 8):
        raise rt.SecondHipError(f"bn_relu_forward: unsupported channel count {c}")
    ws = rt.workspace(nb, y.device)
    z = torch.empty_like(y)
    mean = torch.empty((c,), dtype=torch.float32, device=y.device)
    invstd = torch.empty((c,), dtype=torch.float32, device=y.device)
    rt.check(l.sec_bn_relu_fwd_nhwc(rt.ptr(y), b * h * w, c, rt.ptr(gamma), rt.ptr(beta), float(eps), float(momentum),
                                    rt.ptr(running_mean), rt.ptr(running_var), int(bool(relu)), rt.ptr(z), rt.ptr(mean), rt.ptr(invstd),
                                    rt.ptr(ws), ws.numel(), rt.dtype_code(y.dtype), rt.ptr(rows_dev), rt.stream()), "sec_bn_relu_fwd_nhwc")
    return z, mean, invstd


def bn_relu_backward(dz, y, gamma, beta, save_mean, save_invstd, relu=True, rows_dev=None):
    """-> (dy, dgamma, dbeta) of :func:`bn_relu_forward` (sec_bn_relu_bwd_nhwc); dz, y channels_last 16-bit."""
    rt.require_gpu(dz, y, gamma, beta, save_mean, save_invstd)
    assert dz.shape == y.shape and dz.dtype == y.dtype
    b, c, h, w = _bn_dims(y)
    _bn_dims(dz)
    l = rt.lib()
    ws = rt.workspace(l.sec_bn_train_workspace_bytes(c), y.device)
    dy = torch.empty_like(y)
    dgamma = torch.empty((c,), dtype=torch.float32, device=y.device)
    dbeta = torch.empty((c,), dtype=torch.float32, device=y.device)
    rt.check(l.sec_bn_relu_bwd_nhwc(rt.ptr(dz), rt.ptr(y), b * h * w, c, rt.ptr(gamma), rt.ptr(beta), rt.ptr(save_mean),
                                    rt.ptr(save_invstd), int(bool(relu)), rt.ptr(dy), rt.ptr(dgamma), rt.ptr(dbeta), rt.ptr(ws),
                                    ws.numel(), rt.dtype_code(y.dtype), rt.ptr(rows_dev), rt.stream()), "sec_bn_relu_bwd_nhwc")
    return dy, dgamma, dbeta


def conv2d_pack_weight_train(weight, dtype):
    """fp32 master weight [Cout,Cin,k,k] -> (packed forward image, packed data-gradient image) in `dtype`, one launch
    (sec_conv2d_pack_weight_train).  The second is conv2d_pack_weight(conv2d_dgrad_weight(weight.to(dtype)))."""
    rt.require_gpu(weight)
    assert weight.dtype == torch.float32 and weight.is_contiguous() and weight.shape[2] == weight.shape[3]
    hit = _PREPACK.pop(("2d", weight.data_ptr(), dtype), None)     # packed by prepack_training_weights in this step's one launch
    if hit is not None:
        return hit
    cout, cin, k, _ = weight.shape
    l = rt.lib()
    nbytes = l.sec_conv2d_packed_weight_bytes(cout, cin, k, rt.dtype_code(dtype))
    if nbytes == 0:
        raise rt.SecondHipError(f"conv2d_pack_weight_train: unsupported shape {tuple(weight.shape)}")
    both = torch.empty((2, nbytes // 2), dtype=dtype, device=weight.device)
    rt.check(l.sec_conv2d_pack_weight_train(rt.ptr(weight), cout, cin, k, rt.dtype_code(dtype), rt.ptr(both[0]), rt.ptr(both[1]),
                                            rt.stream()), "sec_conv2d_pack_weight_train")
    return both[0], both[1]


class Conv3x3Function(torch.autograd.Function):
    """nn.Conv2d(128, 128, 3, padding=1, bias=False) on channels_last 16-bit activations over an fp32 master weight: forward and
    data gradient on k_conv2d_halo_reg, weight gradient on k_conv2d_wgrad3x3; both packed weight images come from one launch over
    the master weight.  (ZeroPad2d(1) + Conv2d(padding=0) of the first RPN layer is the same operator.)"""

    @staticmethod
    def forward(ctx, x, weight):
        pk_f, pk_d = conv2d_pack_weight_train(weight.detach().contiguous(), x.dtype)
        y = conv2d_nhwc(x, pk_f, None, weight.shape[0], 3, 1, 1, relu=False)
        ctx.save_for_backward(x, pk_d)
        ctx.wshape, ctx.wdtype = tuple(weight.shape), weight.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pk_d = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_nhwc(dy, pk_d, None, ctx.wshape[1], 3, 1, 1, relu=False)
        if ctx.needs_input_grad[1]:
            dw = conv2d_wgrad(x, dy).to(ctx.wdtype)
        return dx, dw


class ConvTranspose1x1Function(torch.autograd.Function):
    """nn.ConvTranspose2d(128, 128, 1, stride=1, bias=False) -- the deblock of a single-block RPNV2 (rpn.py:275-285) -- on
    channels_last 16-bit activations over its fp32 master weight [Cin, Cout, 1, 1]: y[co] = sum_ci x[ci] W[ci][co] is a 1x1
    convolution with the transposed matrix, its data gradient the 1x1 convolution with W itself, its weight gradient the
    pixel-contraction of k_conv2d_wgrad3x3 with one tap.  Read as a CONV weight [Cin as cout, Cout as cin], the "dgrad image" of
    sec_conv2d_pack_weight_train is the forward's matrix and its "forward image" the backward's: one pack launch."""

    @staticmethod
    def forward(ctx, x, weight):
        pk_b, pk_f = conv2d_pack_weight_train(weight.detach().contiguous(), x.dtype)
        y = conv2d_nhwc(x, pk_f, None, weight.shape[1], 1, 1, 0, relu=False)
        ctx.save_for_backward(x, pk_b)
        ctx.wshape, ctx.wdtype = tuple(weight.shape), weight.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pk_b = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_nhwc(dy, pk_b, None, ctx.wshape[0], 1, 1, 0, relu=False)
        if ctx.needs_input_grad[1]:
            # conv2d_wgrad gives [co][ci] = sum_px x[ci] dy[co]; the transposed conv's weight is [ci][co]
            dw = conv2d_wgrad(x, dy, 1).reshape(ctx.wshape[1], ctx.wshape[0]).t().reshape(ctx.wshape).to(ctx.wdtype)
        return dx, dw


class Heads1x1Function(torch.autograd.Function):
    """The RPN's 1x1 heads (conv_box / conv_cls / conv_dir_cls with bias, rpn.py:386-391) as ONE 1x1 convolution 128 -> 64 (the
    heads' output channels stacked and zero padded) on channels_last 16-bit activations over an fp32 weight [64, 128, 1, 1] and
    bias [64]: forward and data gradient on sec_conv2d_nhwc, weight gradient on the one-tap form of k_conv2d_wgrad3x3, bias gradient
    a pixel sum."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        pk_f, pk_d = conv2d_pack_weight_train(weight.detach().contiguous(), x.dtype)
        y = conv2d_nhwc(x, pk_f, bias.detach().float().contiguous(), weight.shape[0], 1, 1, 0, relu=False)
        ctx.save_for_backward(x, pk_d)
        ctx.wshape, ctx.wdtype, ctx.bdtype = tuple(weight.shape), weight.dtype, bias.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pk_d = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_nhwc(dy, pk_d, None, ctx.wshape[1], 1, 1, 0, relu=False)
        if ctx.needs_input_grad[1]:
            dw = conv2d_wgrad(x, dy, 1).to(ctx.wdtype)         # the one-tap kernel reads the 64-channel gradient as it is
        if ctx.needs_input_grad[2]:
            db = dy.float().sum(dim=(0, 2, 3)).to(ctx.bdtype)
        return dx, dw, db


def heads_loss_supported(head_channels, anchors_per_loc, num_class, num_dir_bins, dtype):
    return dtype in (torch.bfloat16, torch.float16) and bool(rt.lib().sec_heads_loss_supported(
        int(head_channels), int(anchors_per_loc), int(num_class), int(num_dir_bins), rt.dtype_code(dtype)))


def _loss_params(cfg):
    p = dict(LOSS_DEFAULTS, **{k: v for k, v in cfg.items() if k in LOSS_DEFAULTS})
    return rt.f_arr([p["alpha"], p["gamma"], p["sigma"], p["pos_cls_weight"], p["neg_cls_weight"], p["classification_weight"],
                     p["localization_weight"], p["direction_loss_weight"], p["direction_offset"], p["sin_error_factor"],
                     *p["code_weights"]])


class HeadsLossFunction(torch.autograd.Function):
    """loss, out6 = HeadsLossFunction.apply(x, weight, bias, labels, reg_targets, anchors, importance, anchors_per_loc, num_class,
    num_dir_bins, cfg): the stacked 1x1 heads (Heads1x1Function: one 128 -> 64 convolution, box | cls | dir | padding) AND
    VoxelNet.loss (voxelnet.py:239-312) on their output without ever splitting it: sec_heads_loss_fwd reads the channels-last 16-bit
    head tensor, backward's sec_heads_loss_bwd writes the gradient of the loss -- times the gradient arriving at `loss`, a device
    scalar (the loss scale of fp16 training) -- in the same layout, together with the bias gradient; the data and weight gradients of
    the convolution follow on sec_conv2d_nhwc / the one-tap weight-gradient kernel.  Replaces the three permute + float copies the
    reference's formulation needs forward and the ~30 small kernels autograd runs to stitch three gradients back (0.25 ms of a
    3.5 ms car.fhd step).  Same values as Heads1x1Function + SecondLossFunction up to the order of the loss sums."""

    @staticmethod
    def forward(ctx, x, weight, bias, labels, reg_targets, anchors, importance, anchors_per_loc, num_class, num_dir_bins, cfg, want_terms=False):
        """``want_terms``: also return (cls_preds [B, N, C] fp32, cls_loss [B, N, C], loc_loss [B, N, 7]) -- the per-anchor tensors of the
        reference's loss dict (voxelnet.py:299-309), written by the same launch (sec_heads_loss_fwd_terms); not differentiable."""
        rt.require_gpu(x, labels, reg_targets, anchors, importance)
        pk_f, pk_d = conv2d_pack_weight_train(weight.detach().contiguous(), x.dtype)
        y = conv2d_nhwc(x, pk_f, bias.detach().float().contiguous(), weight.shape[0], 1, 1, 0, relu=False)
        b, hc, h, w = y.shape
        a, nc, bins = int(anchors_per_loc), int(num_class), int(num_dir_bins)
        assert labels.dtype == torch.int32 and labels.is_contiguous() and tuple(labels.shape) == (b, a * h * w)
        for t in (reg_targets, anchors, importance):
            assert t.dtype == torch.float32 and t.is_contiguous()
        params = _loss_params(cfg)
        out6 = torch.empty((6,), dtype=torch.float32, device=x.device)
        l = rt.lib()
        ws = rt.workspace(l.sec_heads_loss_workspace_bytes(b, h, w, a), x.device)
        terms = ()
        if want_terms:
            n = a * h * w
            terms = (torch.empty((b, n, nc), dtype=torch.float32, device=x.device), torch.empty((b, n, nc), dtype=torch.float32, device=x.device),
                     torch.empty((b, n, 7), dtype=torch.float32, device=x.device))
            rt.check(l.sec_heads_loss_fwd_terms(rt.ptr(y), rt.dtype_code(y.dtype), b, h, w, hc, a, nc, bins, rt.ptr(labels), rt.ptr(reg_targets),
                                                rt.ptr(anchors), rt.ptr(importance), params, rt.ptr(out6), rt.ptr(terms[0]), rt.ptr(terms[1]),
                                                rt.ptr(terms[2]), rt.ptr(ws), ws.numel(), rt.stream()), "sec_heads_loss_fwd_terms")
        else:
            rt.check(l.sec_heads_loss_fwd(rt.ptr(y), rt.dtype_code(y.dtype), b, h, w, hc, a, nc, bins, rt.ptr(labels), rt.ptr(reg_targets),
                                          rt.ptr(anchors), rt.ptr(importance), params, rt.ptr(out6), rt.ptr(ws), ws.numel(), rt.stream()),
                     "sec_heads_loss_fwd")
        ctx.save_for_backward(x, pk_d, y, labels, reg_targets, anchors, importance, ws)     # ws: the frames' positive counts stay in it
        ctx.meta = (a, nc, bins, params, tuple(weight.shape), weight.dtype, bias.dtype)
        ctx.mark_non_differentiable(out6, *terms)
        return (out6[0], out6) + terms

    @staticmethod
    def backward(ctx, g_loss, _g_all, *_g_terms):
        x, pk_d, y, labels, reg_targets, anchors, importance, ws = ctx.saved_tensors
        a, nc, bins, params, wshape, wdtype, bdtype = ctx.meta
        b, hc, h, w = y.shape
        g = g_loss.detach().reshape(1).float().contiguous()
        dy = torch.empty_like(y)                                  # channels_last like y
        db = torch.empty((hc,), dtype=torch.float32, device=y.device)
        l = rt.lib()
        rt.check(l.sec_heads_loss_bwd(rt.ptr(y), rt.dtype_code(y.dtype), b, h, w, hc, a, nc, bins, rt.ptr(labels), rt.ptr(reg_targets),
                                      rt.ptr(anchors), rt.ptr(importance), params, rt.ptr(g), rt.ptr(dy), rt.ptr(db), rt.ptr(ws),
                                      ws.numel(), 1, rt.stream()), "sec_heads_loss_bwd")
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_nhwc(dy, pk_d, None, wshape[1], 1, 1, 0, relu=False)
        if ctx.needs_input_grad[1]:
            dw = conv2d_wgrad(x, dy, 1).to(wdtype)
        return dx, dw, (db.to(bdtype) if ctx.needs_input_grad[2] else None), None, None, None, None, None, None, None, None, None


class BatchNormReluFunction(torch.autograd.Function):
    """nn.BatchNorm2d (training mode: batch statistics, running statistics updated) + nn.ReLU in two launches + a finalize, on
    channels_last 16-bit activations with fp32 affine parameters."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, eps, momentum, relu, rows_dev=None):
        """``rows_dev`` (device int32[>=1]): static-capacity rows -- statistics and both passes over the first rows_dev[0] rows only."""
        z, mean, invstd = bn_relu_forward(y, gamma.detach().float().contiguous(), beta.detach().float().contiguous(), eps, momentum,
                                          running_mean, running_var, relu, rows_dev=rows_dev)
        ctx.save_for_backward(y, gamma, beta, mean, invstd)
        ctx.relu, ctx.rows_dev = relu, rows_dev
        return z

    @staticmethod
    def backward(ctx, dz):
        y, gamma, beta, mean, invstd = ctx.saved_tensors
        dz = dz.contiguous() if y.dim() == 2 else dz.contiguous(memory_format=torch.channels_last)
        dy, dgamma, dbeta = bn_relu_backward(dz, y, gamma.detach().float().contiguous(),
                                             beta.detach().float().contiguous(), mean, invstd, ctx.relu, rows_dev=ctx.rows_dev)
        return dy, dgamma.to(gamma.dtype), dbeta.to(beta.dtype), None, None, None, None, None, None
