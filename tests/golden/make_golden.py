#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by EXECUTING THE REFERENCE'S OWN PYTHON.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference (traveller59/second.pytorch) cannot be imported as-is here: numba and spconv
are absent.  This script therefore installs *execution shims* --
they contain no algorithm, they only let the reference's unmodified source run:

  * ``numba.jit/njit``         -> identity decorators (the jitted helpers run as plain Python);
  * ``numba.cuda``             -> a tiny SIMT emulator: ``kernel[grid, block, stream](*args)``
                                  runs one Python thread per CUDA thread of a block, with
                                  ``cuda.syncthreads()`` = ``threading.Barrier``,
                                  ``cuda.shared.array`` = one numpy array per block,
                                  ``cuda.local.array`` = fresh numpy array,
                                  ``cuda.blockIdx/threadIdx`` = thread-local proxies.
                                  So ``rotate_nms_gpu``, ``nms_gpu``, ``rotate_iou_gpu_eval``
                                  (second/core/non_max_suppression/nms_gpu.py) execute verbatim.
  * ``spconv``, ``cv2``, ``torchvision.models.resnet`` -> empty stand-ins (import only).

Arithmetic caveat (documented in DESIGN.md): under the shim, fp32 scalars follow numpy
(NEP 50) promotion, numba would promote a few intermediates to fp64; both are within
1e-6 of each other, tests use 2e-5 absolute on IoU values and exact match on keep lists.

Outputs (all small .npz):
  rotate_iou.npz        boxes/qboxes -> iou for criterion -1,0,1,2   (rotate_iou_gpu_eval)
  rotate_nms.npz        dets, thresholds -> keep lists                (rotate_nms_gpu)
  nms_axis_aligned.npz  dets -> keep (nms_gpu, '+1' convention) and nms_jit ('>=' eps)
  standup.npz           center_to_corner_box2d / corner_to_standup_nd / iou_jit(eps=0)
  voxel_coords.npz      simplevis._points_to_bevmap_reverse_kernel: per-point voxel ids
  torch_modules.npz     SimpleVoxel, second_box_decode, limit_period, PointPillarsScatter,
                        PillarFeatureNet, RPNV2 (small) forward on seeded inputs + weights
  train_targets_losses.npz  create_target_np (nearest-IoU matching + box encoding) and VoxelNet.loss
                        (focal / smooth-L1 with sin difference / direction CE) with autograd gradients
"""
import collections
import collections.abc
import contextlib
import os
import sys
import threading
import types

import numpy as np

REF = os.environ.get("SECOND_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- shims
class _Idx:
    def __init__(self, tls, name):
        self._tls, self._name = tls, name

    def _get(self, i):
        return getattr(self._tls, self._name)[i]

    x = property(lambda s: s._get(0))
    y = property(lambda s: s._get(1))
    z = property(lambda s: s._get(2))


class _DevArray(np.ndarray):
    def copy_to_host(self, ary=None, stream=None):
        if ary is None:
            return np.array(self)
        ary[...] = np.asarray(self).reshape(ary.shape)
        return ary


class _Stream:
    @contextlib.contextmanager
    def auto_synchronize(self):
        yield self

    def synchronize(self):
        pass


class _Kernel:
    def __init__(self, fn, tls):
        self.fn, self.tls = fn, tls
        self.__name__ = getattr(fn, "__name__", "kernel")

    def __call__(self, *a, **k):  # device function call
        return self.fn(*a, **k)

    def __getitem__(self, cfg):
        grid, block = cfg[0], cfg[1]
        grid = tuple(grid) if isinstance(grid, (tuple, list)) else (grid,)
        grid = grid + (1,) * (3 - len(grid))
        nthr = int(block) if not isinstance(block, (tuple, list)) else int(np.prod(block))

        def launch(*args):
            for bz in range(grid[2]):
                for by in range(grid[1]):
                    for bx in range(grid[0]):
                        self._run_block((bx, by, bz), nthr, args)
        return launch

    def _run_block(self, bidx, nthr, args):
        barrier = threading.Barrier(nthr)
        shared = {}
        lock = threading.Lock()
        errors = []

        def body(t):
            tls = self.tls
            tls.blockIdx, tls.threadIdx = bidx, (t, 0, 0)
            tls.barrier, tls.shared, tls.shared_ctr, tls.lock = barrier, shared, 0, lock
            try:
                self.fn(*args)
            except BaseException as e:  # pragma: no cover
                errors.append(e)
                barrier.abort()
        ths = [threading.Thread(target=body, args=(t,)) for t in range(nthr)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        if errors:
            raise errors[0]


def install_shims():
    collections.Iterable = collections.abc.Iterable  # torchplus/train/optim.py:1 on py>=3.10
    tls = threading.local()

    def passthrough(*dargs, **dkw):
        if len(dargs) == 1 and callable(dargs[0]) and not dkw:
            return dargs[0]
        return lambda f: f

    numba = types.ModuleType("numba")
    numba.jit = numba.njit = passthrough
    for n in ("float32", "float64", "int32", "int64"):
        setattr(numba, n, getattr(np, n))
    numba.prange = range
    cuda = types.ModuleType("numba.cuda")

    def cuda_jit(*dargs, **dkw):
        if len(dargs) == 1 and callable(dargs[0]) and not dkw:
            return _Kernel(dargs[0], tls)
        return lambda f: _Kernel(f, tls)

    cuda.jit = cuda_jit
    cuda.blockIdx = _Idx(tls, "blockIdx")
    cuda.threadIdx = _Idx(tls, "threadIdx")
    cuda.syncthreads = lambda: tls.barrier.wait()
    cuda.local = types.SimpleNamespace(array=lambda shape, dtype=np.float32: np.zeros(shape, dtype))

    def shared_array(shape, dtype=np.float32):
        key = tls.shared_ctr
        tls.shared_ctr += 1
        with tls.lock:
            if key not in tls.shared:
                tls.shared[key] = np.zeros(shape, dtype)
            return tls.shared[key]

    cuda.shared = types.SimpleNamespace(array=shared_array)
    cuda.to_device = lambda a, stream=None: np.array(a).view(_DevArray)
    cuda.stream = lambda: _Stream()
    cuda.select_device = lambda i: None
    numba.cuda = cuda
    sys.modules["numba"], sys.modules["numba.cuda"] = numba, cuda

    def absent(name):
        def f(*a, **k):
            raise RuntimeError(f"{name} belongs to the absent spconv dependency")
        return f

    spconv = types.ModuleType("spconv")
    utils = types.ModuleType("spconv.utils")
    for n in ("non_max_suppression", "non_max_suppression_cpu", "rotate_non_max_suppression_cpu",
              "rbbox_iou", "rbbox_intersection", "VoxelGeneratorV2", "points_to_voxel"):
        setattr(utils, n, absent(n))
    import torch

    class _SparseModule(torch.nn.Module):
        pass

    spconv.SparseModule = _SparseModule
    spconv.utils = utils
    sys.modules["spconv"], sys.modules["spconv.utils"] = spconv, utils
    cv2 = types.ModuleType("cv2")
    cv2.__getattr__ = lambda name: 0  # constants used as default args at import time
    sys.modules["cv2"] = cv2
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvr = types.ModuleType("torchvision.models.resnet")
    tv.models, tvm.resnet = tvm, tvr
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.resnet": tvr})
    if REF not in sys.path:
        sys.path.insert(0, REF)


# --------------------------------------------------------------------------- inputs
def random_rboxes(rng, n, extent=12.0):
    """(x, y, w, l, r) boxes that overlap often."""
    xy = rng.uniform(0, extent, (n, 2))
    wl = rng.uniform(0.8, 4.5, (n, 2))
    r = rng.uniform(-np.pi, np.pi, (n, 1))
    return np.concatenate([xy, wl, r], 1).astype(np.float32)


def special_rbox_pairs():
    a = [[0, 0, 1, 1, 0], [0, 0, 1, 1, 0], [0, 0, 1, 1, 0], [0, 0, 2, 4, 0.3], [5, 5, 2, 3, 1.0],
         [0, 0, 1, 1, 0], [1, 1, 3.9, 1.6, 1.57], [0, 0, 2, 2, 0]]
    b = [[0, 0, 1, 1, 0], [0.5, 0, 1, 1, 0], [0, 0, 1, 1, np.pi / 4], [0.5, 0.2, 2, 4, -0.4],
         [50, 50, 2, 3, 1.0], [1.0, 0, 1, 1, 0], [1.2, 0.9, 3.9, 1.6, 0.0], [0, 0, 1, 1, 0.5]]
    return np.array(a, np.float32), np.array(b, np.float32)


# --------------------------------------------------------------------------- generators
def gen_rotate_iou():
    import importlib
    nms_gpu = importlib.import_module('second.core.non_max_suppression.nms_gpu')
    rng = np.random.default_rng(1)
    boxes = random_rboxes(rng, 40)
    qboxes = random_rboxes(rng, 30)
    sa, sb = special_rbox_pairs()
    out = {"boxes": boxes, "qboxes": qboxes, "special_a": sa, "special_b": sb}
    for crit in (-1, 0, 1, 2):
        out[f"iou_c{crit}"] = nms_gpu.rotate_iou_gpu_eval(boxes, qboxes, crit)
    out["iou_plain"] = nms_gpu.rotate_iou_gpu(boxes, qboxes)
    out["special_iou"] = np.array(
        [nms_gpu.rotate_iou_gpu_eval(sa[i:i + 1], sb[i:i + 1], -1)[0, 0] for i in range(len(sa))], np.float32)
    np.savez_compressed(os.path.join(OUT, "rotate_iou.npz"), **out)
    print("rotate_iou: special", out["special_iou"])


def gen_rotate_nms():
    import importlib
    nms_gpu = importlib.import_module('second.core.non_max_suppression.nms_gpu')
    rng = np.random.default_rng(2)
    out = {}
    for tag, n, ext in (("a", 100, 14.0), ("b", 70, 40.0), ("c", 1, 5.0), ("d", 65, 6.0)):
        b = random_rboxes(rng, n, ext)
        scores = rng.permutation(n).astype(np.float32) / n + 0.001  # distinct scores
        dets = np.concatenate([b, scores[:, None]], 1).astype(np.float32)
        out[f"dets_{tag}"] = dets
        for thr in (0.01, 0.3):
            keep = nms_gpu.rotate_nms_gpu(dets, thr)
            out[f"keep_{tag}_{thr}"] = np.array(keep, np.int64)
            print(f"rotate_nms {tag} thr={thr}: keep {len(keep)}/{n}")
    np.savez_compressed(os.path.join(OUT, "rotate_nms.npz"), **out)


def gen_nms_axis_aligned():
    import importlib
    nms_gpu = importlib.import_module('second.core.non_max_suppression.nms_gpu')
    nms_cpu = importlib.import_module('second.core.non_max_suppression.nms_cpu')
    rng = np.random.default_rng(3)
    n = 150
    xy = rng.uniform(0, 60, (n, 2))
    wh = rng.uniform(2, 25, (n, 2))
    scores = rng.permutation(n).astype(np.float32) / n + 0.001
    dets = np.concatenate([xy, xy + wh, scores[:, None]], 1).astype(np.float32)
    out = {"dets": dets}
    for thr in (0.1, 0.5):
        out[f"keep_gpu_{thr}"] = np.array(nms_gpu.nms_gpu(dets, thr), np.int64)
        out[f"keep_jit_eps0_{thr}"] = np.array(nms_cpu.nms_jit(dets, thr, 0.0), np.int64)
        out[f"keep_jit_eps1_{thr}"] = np.array(nms_cpu.nms_jit(dets, thr, 1.0), np.int64)
        print("nms aa", thr, len(out[f"keep_gpu_{thr}"]), len(out[f"keep_jit_eps0_{thr}"]))
    np.savez_compressed(os.path.join(OUT, "nms_axis_aligned.npz"), **out)


def gen_standup():
    from second.core import box_np_ops
    rng = np.random.default_rng(4)
    dets = random_rboxes(rng, 50)
    corners = box_np_ops.center_to_corner_box2d(dets[:, :2], dets[:, 2:4], dets[:, 4])
    standup = box_np_ops.corner_to_standup_nd(corners)
    iou = box_np_ops.iou_jit(standup, standup, eps=0.0)
    np.savez_compressed(os.path.join(OUT, "standup.npz"), dets=dets, corners=corners.astype(np.float32),
                        standup=standup.astype(np.float32), standup_iou=iou.astype(np.float32))
    print("standup: corners", corners.shape, corners.dtype)


def gen_voxel_coords():
    """second/utils/simplevis.py:8-60 is the in-repo copy of spconv's points_to_voxel loop."""
    from second.utils import simplevis
    rng = np.random.default_rng(5)
    out = {}
    cases = {
        "kitti": dict(voxel_size=[0.05, 0.05, 0.1], rng_=[0, -40, -3, 70.4, 40, 1], n=3000, cap=40000),
        "coarse_cap": dict(voxel_size=[0.4, 0.4, 0.5], rng_=[0, -8, -3, 16, 8, 1], n=2500, cap=300),
    }
    for tag, c in cases.items():
        vs = np.array(c["voxel_size"], np.float32)
        pr = np.array(c["rng_"], np.float32)
        lo, hi = pr[:3] - 1.0, pr[3:] + 1.0  # some points out of range
        pts = rng.uniform(lo, hi, (c["n"], 3)).astype(np.float32)
        # put points exactly on voxel borders / range bounds (fp32 division edge cases)
        k = c["n"] // 10
        cells = rng.integers(0, 40, (k, 3)).astype(np.float32)
        pts[:k] = pr[:3] + cells * vs
        pts[k] = pr[3:]            # upper bound: must be dropped
        pts[k + 1] = pr[:3]        # lower bound: kept
        pts = np.concatenate([pts, rng.uniform(0, 1, (c["n"], 1)).astype(np.float32)], 1)
        # duplicates so that some voxels hold several points
        pts[-200:] = pts[100:300]
        grid = np.round((pr[3:] - pr[:3]) / vs).astype(np.int32)
        lookup = -np.ones(tuple(grid[::-1]), np.int32)
        bev = np.zeros((int(grid[2]) + 2, int(grid[1]), int(grid[0])), np.float32)
        lowers = np.linspace(pr[2], pr[5], int(grid[2]), endpoint=False).astype(np.float32)
        simplevis._points_to_bevmap_reverse_kernel(pts, vs, pr, lookup, bev, lowers, False, c["cap"])
        zyx = np.argwhere(lookup >= 0)
        vid = lookup[lookup >= 0]
        order = np.argsort(vid)
        out[f"{tag}_points"] = pts
        out[f"{tag}_voxel_size"] = vs
        out[f"{tag}_range"] = pr
        out[f"{tag}_cap"] = np.int64(c["cap"])
        out[f"{tag}_coors"] = zyx[order].astype(np.int32)   # voxel id order, (z,y,x)
        out[f"{tag}_density"] = bev[-1].copy()               # points counted per (y,x) column
        print("voxel", tag, "grid", grid, "voxels", len(vid))
    np.savez_compressed(os.path.join(OUT, "voxel_coords.npz"), **out)


def gen_torch_modules():
    import torch
    from second.pytorch.models import voxel_encoder, pointpillars, rpn
    from second.pytorch.core import box_torch_ops
    torch.manual_seed(0)
    out = {}
    # SimpleVoxel (voxel_encoder.py:207-225)
    vox = torch.rand(64, 5, 4)
    npts = torch.randint(1, 6, (64,), dtype=torch.int32)
    for i in range(64):
        vox[i, npts[i]:] = 0
    sv = voxel_encoder.SimpleVoxel(num_input_features=4)
    out["sv_voxels"], out["sv_num_points"] = vox.numpy(), npts.numpy()
    out["sv_out"] = sv(vox, npts, None).numpy()
    # second_box_decode / limit_period (box_torch_ops.py:56-101,370-371)
    enc = torch.randn(200, 7) * 0.3
    anchors = torch.cat([torch.rand(200, 3) * 40, torch.rand(200, 3) * 3 + 0.5, torch.rand(200, 1) * 3], 1)
    out["dec_enc"], out["dec_anchors"] = enc.numpy(), anchors.numpy()
    out["dec_out"] = box_torch_ops.second_box_decode(enc, anchors).numpy()
    val = torch.randn(100) * 6
    out["lp_val"] = val.numpy()
    out["lp_out"] = box_torch_ops.limit_period(val, 1.0, np.pi).numpy()
    # PointPillarsScatter (pointpillars.py:444-476)
    ny, nx, c, bsz = 12, 10, 8, 2
    coords = []
    for b in range(bsz):
        cells = torch.randperm(ny * nx)[:30]
        coords.append(torch.stack([torch.full_like(cells, b), torch.zeros_like(cells), cells // nx, cells % nx], 1))
    coords = torch.cat(coords).int()
    feats = torch.randn(coords.shape[0], c)
    sc = pointpillars.PointPillarsScatter(output_shape=[bsz, 1, ny, nx, c], num_input_features=c)
    out["ps_feats"], out["ps_coords"] = feats.numpy(), coords.numpy()
    out["ps_out"] = sc(feats, coords, bsz).numpy()
    # PillarFeatureNet (pointpillars.py:150-237), eval-mode BN with random stats
    pfn = pointpillars.PillarFeatureNet(num_input_features=4, use_norm=True, num_filters=(16,),
                                        voxel_size=(0.25, 0.25, 8), pc_range=(-50, -50, -5, 50, 50, 3))
    bn = pfn.pfn_layers[0].norm
    bn.running_mean.uniform_(-0.1, 0.1)
    bn.running_var.uniform_(0.5, 1.5)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.uniform_(-0.2, 0.2)
    pfn.eval()
    P, T = 40, 10
    pv = torch.rand(P, T, 4) * 4 - 2
    pn = torch.randint(1, T + 1, (P,), dtype=torch.int32)
    for i in range(P):
        pv[i, pn[i]:] = 0
    pc = torch.stack([torch.zeros(P), torch.zeros(P), torch.randint(0, 400, (P,)).float(),
                      torch.randint(0, 400, (P,)).float()], 1).int()
    with torch.no_grad():
        out["pfn_out"] = pfn(pv, pn, pc).numpy()
    out["pfn_voxels"], out["pfn_num_points"], out["pfn_coords"] = pv.numpy(), pn.numpy(), pc.numpy()
    for k, v in pfn.state_dict().items():
        out["pfn_sd." + k] = v.numpy()
    # RPNV2 (rpn.py:468-497 + RPNBase :334-420), small width so the fixture stays small
    net = rpn.RPNV2(use_norm=True, num_class=1, layer_nums=(2,), layer_strides=(1,), num_filters=(16,),
                    upsample_strides=(1,), num_upsample_filters=(16,), num_input_features=16,
                    num_anchor_per_loc=2, encode_background_as_zeros=True, use_direction_classifier=True,
                    box_code_size=7, num_direction_bins=2)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.1, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.2, 0.2)
    net.eval()
    x = torch.randn(2, 16, 12, 10)
    with torch.no_grad():
        r = net(x)
    out["rpn_in"] = x.numpy()
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        out["rpn_" + k] = r[k].numpy()
    for k, v in net.state_dict().items():
        out["rpn_sd." + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "torch_modules.npz"), **out)
    print("torch modules: rpn box_preds", out["rpn_box_preds"].shape)


def gen_train_targets_losses():
    """Training-side fixtures: the reference's own target assignment (second/core/target_ops.py:29 create_target_np with
    NearestIouSimilarity, region_similarity.py:73-93, and GroundBox3dCoder.encode = box_np_ops.second_box_encode) and its own
    loss (second/pytorch/models/voxelnet.py:239-312 VoxelNet.loss -> create_loss / prepare_loss_weights /
    get_direction_target over losses.py:135-296), executed on seeded inputs; gradients by autograd."""
    import torch
    if not getattr(np.meshgrid, "_as_list", False):            # numpy >= 2 returns a tuple; box_np_ops.py:626-629 assigns into it
        _mg = np.meshgrid
        np.meshgrid = lambda *a, **k: list(_mg(*a, **k))
        np.meshgrid._as_list = True
    from second.core import box_np_ops, region_similarity, target_ops
    from second.pytorch.core import losses
    from second.pytorch.models import voxelnet as V
    rng = np.random.default_rng(11)
    fm = [1, 50, 44]
    anchors = box_np_ops.create_anchors_3d_range(fm, [0, -40.0, -1.0, 70.4, 40.0, -1.0], sizes=[1.6, 3.9, 1.56],
                                                 rotations=[0, 1.57], dtype=np.float32).reshape(-1, 7)
    sim = region_similarity.NearestIouSimilarity()

    def similarity_fn(a, g):
        return sim.compare(a[:, [0, 1, 3, 4, 6]], g[:, [0, 1, 3, 4, 6]])

    def encode(boxes, anc):
        return box_np_ops.second_box_encode(boxes, anc)
    frames = []
    pick = rng.choice(len(anchors), 14, replace=False)
    g0 = anchors[pick].copy()
    g0[:, :2] += rng.normal(0, 0.2, (14, 2)).astype(np.float32)
    g0[:, 2] += rng.normal(0, 0.1, 14).astype(np.float32)
    g0[:, 3:6] *= rng.uniform(0.85, 1.2, (14, 3)).astype(np.float32)
    g0[:7, 6] += rng.normal(0, 0.15, 7).astype(np.float32)     # half of them roughly aligned with their anchor ...
    g0[7:, 6] = rng.uniform(-np.pi, np.pi, 7).astype(np.float32)   # ... the others at any heading
    g0[0] = anchors[pick[0]]                                   # one ground truth sits exactly on an anchor (IoU 1)
    frames.append(g0.astype(np.float32))
    frames.append(np.zeros((0, 7), np.float32))                # a frame without ground truth
    g2 = anchors[rng.choice(len(anchors), 5, replace=False)].copy()
    g2[:, :2] += rng.normal(0, 0.6, (5, 2)).astype(np.float32)
    g2[:, 6] = rng.uniform(-np.pi, np.pi, 5).astype(np.float32)
    g2[4, :2] = [500.0, 500.0]                                 # a box no anchor overlaps (the empty_gt_mask path)
    frames.append(g2.astype(np.float32))
    out = {"anchors": anchors, "feature_map_size": np.array(fm), "matched_threshold": np.float32(0.6),
           "unmatched_threshold": np.float32(0.45)}
    labels, targets, importance = [], [], []
    for f, g in enumerate(frames):
        r = target_ops.create_target_np(anchors, g, similarity_fn, encode, matched_threshold=0.6, unmatched_threshold=0.45,
                                        gt_classes=np.ones(len(g), np.int32), positive_fraction=None, rpn_batch_size=512,
                                        norm_by_num_examples=False, box_code_size=7)
        out[f"gt_{f}"] = g
        labels.append(r["labels"]); targets.append(r["bbox_targets"]); importance.append(r["importance"])
    out["labels"], out["bbox_targets"], out["importance"] = np.stack(labels), np.stack(targets).astype(np.float32), np.stack(importance)
    print("targets: positives per frame", [(l > 0).sum() for l in labels], "negatives", [(l == 0).sum() for l in labels])
    # ---- loss (car.fhd settings: focal gamma 2 alpha 0.25, smooth-L1 sigma 3, sin-difference, direction classifier)
    b, n = len(frames), len(anchors)
    tg = torch.Generator().manual_seed(5)
    cls = (torch.randn(b, n, 1, generator=tg) * 2 - 2).requires_grad_()
    box = (torch.randn(b, n, 7, generator=tg) * 0.3).requires_grad_()
    dirp = torch.randn(b, n, 2, generator=tg).requires_grad_()
    lab = torch.from_numpy(out["labels"]).int()
    reg = torch.from_numpy(out["bbox_targets"])
    imp = torch.from_numpy(out["importance"])
    cls_w, reg_w, cared = V.prepare_loss_weights(lab, pos_cls_weight=1.0, neg_cls_weight=1.0,
                                                 loss_norm_type=V.LossNormType.NormByNumPositives, dtype=torch.float32)
    cls_t = (lab * cared.type_as(lab)).unsqueeze(-1)
    loc_ftor = losses.WeightedSmoothL1LocalizationLoss(sigma=3.0, code_weights=[1.0] * 7, codewise=True)
    cls_ftor = losses.SigmoidFocalClassificationLoss(gamma=2.0, alpha=0.25)
    dir_ftor = losses.WeightedSoftmaxClassificationLoss()
    loc_loss, cls_loss = V.create_loss(loc_ftor, cls_ftor, box_preds=box, cls_preds=cls, cls_targets=cls_t, cls_weights=cls_w * imp,
                                       reg_targets=reg, reg_weights=reg_w * imp, num_class=1, encode_rad_error_by_sin=True,
                                       encode_background_as_zeros=True, box_code_size=7, sin_error_factor=1.0, num_direction_bins=2)
    loc_red = loc_loss.sum() / b * 2.0
    cls_red = cls_loss.sum() / b * 1.0
    pos_l, neg_l = V._get_pos_neg_loss(cls_loss, lab)
    anc_t = torch.from_numpy(anchors).unsqueeze(0).repeat(b, 1, 1)
    dir_t = V.get_direction_target(anc_t, reg, dir_offset=0.0, num_bins=2)
    w = (lab > 0).type_as(dirp) * imp
    w = w / torch.clamp(w.sum(-1, keepdim=True), min=1.0)
    dir_loss = dir_ftor(dirp, dir_t, weights=w).sum() / b
    loss = loc_red + cls_red + dir_loss * 0.2
    loss.backward()
    out.update(cls_preds=cls.detach().numpy(), box_preds=box.detach().numpy(), dir_preds=dirp.detach().numpy(),
               loss=np.float32(loss.item()), loc_loss_reduced=np.float32(loc_red.item()), cls_loss_reduced=np.float32(cls_red.item()),
               dir_loss_reduced=np.float32(dir_loss.item()), cls_pos_loss=np.float32(pos_l.item()), cls_neg_loss=np.float32(neg_l.item()),
               d_cls=cls.grad.numpy(), d_box=box.grad.numpy(), d_dir=dirp.grad.numpy(), dir_targets=dir_t.argmax(-1).numpy().astype(np.int32),
               cls_loss=cls_loss.detach().numpy(), loc_loss=loc_loss.detach().numpy())
    np.savez_compressed(os.path.join(OUT, "train_targets_losses.npz"), **out)
    print("loss", loss.item(), "loc", loc_red.item(), "cls", cls_red.item(), "dir", dir_loss.item())


def gen_train_targets_multiclass():
    """Multi-class fixtures (BASELINE configs 4 / 5): the reference's TargetAssigner (second/core/target_assigner.py) over three
    AnchorGeneratorRange generators -- two rotations, one rotation, two sizes -- in both of its modes: assign_per_class
    (all.fhd.config:295) and assign_all with per-anchor threshold arrays (all.pp.largea.config:269); non-uniform gt importance
    (pins how assign_per_class indexes it); then VoxelNet.loss pieces for num_class = 3."""
    import torch
    if not getattr(np.meshgrid, "_as_list", False):
        _mg = np.meshgrid
        np.meshgrid = lambda *a, **k: list(_mg(*a, **k))
        np.meshgrid._as_list = True
    from second.core import region_similarity
    from second.core.anchor_generator import AnchorGeneratorRange
    from second.core.box_coders import GroundBox3dCoder
    from second.core.target_assigner import TargetAssigner
    from second.pytorch.core import losses
    from second.pytorch.models import voxelnet as V
    rng = np.random.default_rng(23)
    fm = [1, 40, 36]
    classes = ["car", "pedestrian", "trailer"]
    spec = [dict(sizes=[1.95, 4.6, 1.72], rotations=[0, 1.57], z=-0.94, mt=0.4, ut=0.3),
            dict(sizes=[0.66, 0.73, 1.76], rotations=[0], z=-0.74, mt=0.5, ut=0.35),
            dict(sizes=[3.0, 15.0, 3.8, 2.0, 3.0, 3.8], rotations=[0, 1.57], z=0.22, mt=0.5, ut=0.35)]
    gens = [AnchorGeneratorRange([-20, -20, sp["z"], 20, 20, sp["z"]], sizes=sp["sizes"], rotations=sp["rotations"], class_name=c,
                                 match_threshold=sp["mt"], unmatch_threshold=sp["ut"]) for c, sp in zip(classes, spec)]
    sims = [region_similarity.NearestIouSimilarity() for _ in classes]
    out = {"feature_map_size": np.array(fm), "matched": np.array([sp["mt"] for sp in spec], np.float32),
           "unmatched": np.array([sp["ut"] for sp in spec], np.float32)}
    res = {}
    for per_class in (True, False):
        ta = TargetAssigner(GroundBox3dCoder(), gens, classes, feature_map_sizes=[fm] * 3, positive_fraction=None,
                            region_similarity_calculators=sims, sample_size=512, assign_per_class=per_class)
        ret = ta.generate_anchors(fm)
        anchors = ret["anchors"].reshape(-1, 7)
        adict = ta.generate_anchors_dict(fm)
        begin = np.cumsum([0] + [adict[c]["anchors"].shape[0] for c in classes])
        if per_class:
            out["anchors"], out["class_anchor_begin"] = anchors, begin.astype(np.int32)
            # ground truth: jittered copies of anchors of each class
            frames = []
            g, names = [], []
            for ci, (c, k) in enumerate(zip(classes, (5, 4, 3))):
                a = adict[c]["anchors"]
                b = a[rng.choice(len(a), k, replace=False)].copy()
                b[:, :2] += rng.normal(0, 0.15 if ci != 1 else 0.05, (k, 2)).astype(np.float32)
                b[:, 3:6] *= rng.uniform(0.9, 1.15, (k, 3)).astype(np.float32)
                b[:, 6] += rng.normal(0, 0.2, k).astype(np.float32)
                g.append(b); names += [c] * k
            order = rng.permutation(len(names))             # classes interleaved: index-within-class != index-in-frame
            frames.append((np.concatenate(g)[order].astype(np.float32), [names[i] for i in order]))
            a = adict["car"]["anchors"]
            b = a[rng.choice(len(a), 4, replace=False)].copy()
            b[:, :2] += rng.normal(0, 0.3, (4, 2)).astype(np.float32)
            frames.append((b.astype(np.float32), ["car"] * 4))                     # no pedestrian / trailer boxes
            frames.append((np.zeros((0, 7), np.float32), []))                      # no ground truth at all
            imps = [rng.uniform(0.5, 1.5, len(f[0])).astype(np.float32) for f in frames]
        labels, targets, importance = [], [], []
        for f, ((gt, names), imp) in enumerate(zip(frames, imps)):
            gt_classes = np.array([classes.index(n) + 1 for n in names], np.int32)
            r = ta.assign(anchors, adict, gt, None, gt_classes=gt_classes, gt_names=np.array(names),
                          matched_thresholds=ret["matched_thresholds"], unmatched_thresholds=ret["unmatched_thresholds"],
                          importance=imp)
            labels.append(r["labels"]); targets.append(r["bbox_targets"]); importance.append(r["importance"])
            out[f"gt_{f}"], out[f"gt_classes_{f}"], out[f"gt_importance_{f}"] = gt, gt_classes, imp
        tag = "per_class" if per_class else "all"
        out[f"labels_{tag}"] = np.stack(labels).astype(np.int32)
        out[f"bbox_targets_{tag}"] = np.stack(targets).astype(np.float32)
        out[f"importance_{tag}"] = np.stack(importance).astype(np.float32)
        print(tag, "positives per frame / class", [[int((l == k).sum()) for k in (1, 2, 3)] for l in labels],
              "dont-care", [int((l == -1).sum()) for l in labels])
    # ---- loss for three classes (all.fhd settings: focal, smooth-L1 sigma 3, sin difference, direction offset 0.78)
    b, n = len(frames), len(out["anchors"])
    tg = torch.Generator().manual_seed(9)
    cls = (torch.randn(b, n, 3, generator=tg) * 2 - 2).requires_grad_()
    box = (torch.randn(b, n, 7, generator=tg) * 0.3).requires_grad_()
    dirp = torch.randn(b, n, 2, generator=tg).requires_grad_()
    lab = torch.from_numpy(out["labels_per_class"]).int()
    reg = torch.from_numpy(out["bbox_targets_per_class"])
    imp = torch.from_numpy(out["importance_per_class"])
    cls_w, reg_w, cared = V.prepare_loss_weights(lab, pos_cls_weight=1.0, neg_cls_weight=1.0,
                                                 loss_norm_type=V.LossNormType.NormByNumPositives, dtype=torch.float32)
    cls_t = (lab * cared.type_as(lab)).unsqueeze(-1)
    loc_ftor = losses.WeightedSmoothL1LocalizationLoss(sigma=3.0, code_weights=[1.0] * 7, codewise=True)
    cls_ftor = losses.SigmoidFocalClassificationLoss(gamma=2.0, alpha=0.25)
    dir_ftor = losses.WeightedSoftmaxClassificationLoss()
    loc_loss, cls_loss = V.create_loss(loc_ftor, cls_ftor, box_preds=box, cls_preds=cls, cls_targets=cls_t, cls_weights=cls_w * imp,
                                       reg_targets=reg, reg_weights=reg_w * imp, num_class=3, encode_rad_error_by_sin=True,
                                       encode_background_as_zeros=True, box_code_size=7, sin_error_factor=1.0, num_direction_bins=2)
    loc_red = loc_loss.sum() / b * 2.0
    cls_red = cls_loss.sum() / b * 1.0
    pos_l, neg_l = V._get_pos_neg_loss(cls_loss, lab)
    anc_t = torch.from_numpy(out["anchors"]).unsqueeze(0).repeat(b, 1, 1)
    dir_t = V.get_direction_target(anc_t, reg, dir_offset=0.78, num_bins=2)
    w = (lab > 0).type_as(dirp) * imp
    w = w / torch.clamp(w.sum(-1, keepdim=True), min=1.0)
    dir_loss = dir_ftor(dirp, dir_t, weights=w).sum() / b
    loss = loc_red + cls_red + dir_loss * 0.2
    loss.backward()
    out.update(cls_preds=cls.detach().numpy(), box_preds=box.detach().numpy(), dir_preds=dirp.detach().numpy(),
               loss=np.float32(loss.item()), loc_loss_reduced=np.float32(loc_red.item()), cls_loss_reduced=np.float32(cls_red.item()),
               dir_loss_reduced=np.float32(dir_loss.item()), cls_pos_loss=np.float32(pos_l.item()), cls_neg_loss=np.float32(neg_l.item()),
               d_cls=cls.grad.numpy(), d_box=box.grad.numpy(), d_dir=dirp.grad.numpy())
    np.savez_compressed(os.path.join(OUT, "train_targets_multiclass.npz"), **out)
    print("loss", loss.item(), "loc", loc_red.item(), "cls", cls_red.item(), "dir", dir_loss.item(), "pos", pos_l.item(), "neg", neg_l.item())


if __name__ == "__main__":
    install_shims()
    which = sys.argv[1:] or ["rotate_iou", "rotate_nms", "nms_axis_aligned", "standup", "voxel_coords",
                             "torch_modules", "train_targets_losses", "train_targets_multiclass"]
    for w in which:
        globals()["gen_" + w]()
