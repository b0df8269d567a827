"""Import shims that let the UNMODIFIED reference (traveller59/second.pytorch, v1.6.0-alpha, a 2019 code
base) import on python 3.10 / torch 2.10 / numpy 2 / protobuf 7 with this repo's drop-in ``spconv``.

    from second_amd import compat
    compat.install("/path/to/second.pytorch")      # then: import second.pytorch.train

Nothing here contains hot-path arithmetic: only stand-ins for optional third-party imports the reference
performs at module import time and a loader for its protoc-3.x era ``*_pb2.py`` files.
"""
import collections
import collections.abc
import glob
import importlib
import os
import re
import sys
import types


def _missing(name):
    try:
        importlib.import_module(name)
        return False
    except Exception:
        return True


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    def _getattr(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Placeholder(f"{name}.{attr}")
    m.__getattr__ = _getattr
    sys.modules[name] = m
    return m


class _Placeholder:
    """Import-time placeholder: usable as a default argument / attribute, fails loudly when called."""

    def __init__(self, name):
        self._name = name

    def __call__(self, *a, **k):
        raise ImportError(f"{self._name} is not available in this environment (second_amd.compat placeholder)")

    def __getattr__(self, attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Placeholder(f"{self._name}.{attr}")


def _install_numba():
    """numba.jit/njit as identity decorators: the reference's CPU helpers (anchors, target assignment, box math)
    then run as plain Python/numpy; its numba.cuda kernels are replaced by libsecond_hip (never called)."""
    import numpy as np

    def passthrough(*dargs, **dkw):
        if len(dargs) == 1 and callable(dargs[0]) and not dkw:
            return dargs[0]
        return lambda f: f

    numba = _stub("numba", jit=passthrough, njit=passthrough, prange=range, float32=np.float32, float64=np.float64,
                  int32=np.int32, int64=np.int64)
    cuda = _stub("numba.cuda", jit=passthrough)
    numba.cuda = cuda


def _install_fire():
    def Fire(component=None, *a, **k):
        import inspect
        argv = sys.argv[1:]
        if not argv:
            return component
        fn = getattr(sys.modules["__main__"], argv[0]) if not callable(component) else component
        kwargs = {}
        for tok in argv[1:]:
            if tok.startswith("--") and "=" in tok:
                key, val = tok[2:].split("=", 1)
                try:
                    import ast
                    val = ast.literal_eval(val)
                except Exception:
                    pass
                kwargs[key] = val
        return fn(**kwargs)
    _stub("fire", Fire=Fire)


def _install_tensorboardx():
    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None
    _stub("tensorboardX", SummaryWriter=SummaryWriter)


def _install_torchvision():
    import torch

    class _Block(torch.nn.Module):
        expansion = 1
    tv = _stub("torchvision")
    tvm = _stub("torchvision.models")
    tvr = _stub("torchvision.models.resnet", BasicBlock=_Block, Bottleneck=_Block)
    tv.models, tvm.resnet = tvm, tvr


def load_protos(reference_root):
    """Rebuild second/protos/*_pb2 message classes from the serialized descriptors embedded in the generated
    files (they cannot be imported under protobuf >= 4) and install them as ``second.protos.<x>_pb2``."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    proto_dir = os.path.join(reference_root, "second", "protos")
    files = {}
    for path in sorted(glob.glob(os.path.join(proto_dir, "*_pb2.py"))):
        src = open(path).read()
        m = re.search(r"serialized_pb=_b\('((?:[^'\\]|\\.)*)'\)", src, re.S)
        if not m:
            continue
        blob = eval("b'" + m.group(1) + "'")  # the literal the generated file itself would evaluate
        fdp = descriptor_pb2.FileDescriptorProto()
        fdp.ParseFromString(blob)
        files[fdp.name] = (fdp, os.path.basename(path)[:-3])
    pool = descriptor_pool.DescriptorPool()
    done = set()

    def add(name):
        if name in done:
            return
        fdp, _ = files[name]
        for dep in fdp.dependency:
            add(dep)
        pool.Add(fdp)
        done.add(name)
    for name in files:
        add(name)
    import second  # the reference package (namespace for second.protos)
    pkg = importlib.import_module("second.protos")
    for name, (fdp, modname) in files.items():
        mod = types.ModuleType(f"second.protos.{modname}")
        fd = pool.FindFileByName(name)
        mod.DESCRIPTOR = fd
        for msg_name, desc in fd.message_types_by_name.items():
            setattr(mod, msg_name, message_factory.GetMessageClass(desc))
        for enum_name, enum in fd.enum_types_by_name.items():
            setattr(mod, enum_name, enum)
            for v in enum.values:
                setattr(mod, v.name, v.number)
        sys.modules[mod.__name__] = mod
        setattr(pkg, modname, mod)
    return sorted(m for _, m in files.values())


_SPAWN_SCOPE = None          # None: DataLoader untouched; "reference": the reference's own loaders only; "all": every loader with workers


def _from_reference(obj):
    mod = getattr(obj, "__module__", None) or getattr(type(obj), "__module__", "") or ""
    return mod == "second" or mod.startswith("second.") or mod.startswith("torchplus")


def spawn_dataloader_workers(scope="all"):
    """``torch.utils.data.DataLoader(num_workers > 0)`` defaults to the ``spawn`` start method.

    The reference forks its loader workers (second/pytorch/train.py:262-277, ``num_workers = 3``) and the workers call
    ``spconv.utils.VoxelGeneratorV2.generate`` (second/data/preprocess.py:301-316).  Upstream that is a CPU loop; here it
    runs on the MI355X, and a HIP context does not survive ``fork()`` ("Cannot re-initialize CUDA in forked subprocess").
    Spawned workers import this package afresh and create their own context.  An explicit ``multiprocessing_context``
    argument is left alone.  Idempotent.

    ``scope``: "reference" (what a bare ``import spconv`` installs, the zero-edit path) touches only loaders whose dataset or
    collate function comes from the reference's packages (``second.*``: DatasetWrapper, merge_second_batch) -- any other
    DataLoader of the host program keeps its start method; "all" (``compat.install()``, an explicit call) switches every loader
    with workers.  The first loader that is switched is reported once on stderr."""
    global _SPAWN_SCOPE
    import torch.utils.data as tud
    if _SPAWN_SCOPE == "all" or (_SPAWN_SCOPE == "reference" and scope == "reference"):
        return
    _SPAWN_SCOPE = scope
    if getattr(tud.DataLoader, "_second_amd_spawn", False):
        return
    base = tud.DataLoader
    told = []

    class DataLoader(base):
        _second_amd_spawn = True

        def __init__(self, *a, **k):
            if k.get("num_workers", 0) > 0 and k.get("multiprocessing_context") is None:
                dataset = a[0] if a else k.get("dataset")
                if _SPAWN_SCOPE == "all" or _from_reference(dataset) or _from_reference(k.get("collate_fn")):
                    k["multiprocessing_context"] = "spawn"
                    if not told:
                        told.append(True)
                        import sys as _sys
                        print("[second_amd] DataLoader workers use the 'spawn' start method (they voxelise on the GPU; a HIP context "
                              "does not survive fork()).  multiprocessing_context=... or SEC_KEEP_FORK=1 overrides.", file=_sys.stderr)
            super().__init__(*a, **k)
    DataLoader.__name__, DataLoader.__qualname__ = base.__name__, base.__qualname__
    tud.DataLoader = DataLoader


def install(reference_root=None, spconv_path=None, spawn_workers=True):
    """Make ``import second...`` / ``import torchplus`` work.  Idempotent.  ``spawn_workers``: see
    :func:`spawn_dataloader_workers` (needed whenever the reference's DataLoader has worker processes)."""
    collections.Iterable = collections.abc.Iterable      # torchplus/train/optim.py:1, fastai_optim.py:1
    import torch  # noqa: F401  (before any stand-in module exists: torch introspects sys.modules at import)
    import numpy as np
    if not getattr(np.meshgrid, "_second_amd_list", False):
        _meshgrid = np.meshgrid

        def meshgrid(*a, **k):   # numpy >= 2 returns a tuple; box_np_ops.py:626-629 assigns into the result
            return list(_meshgrid(*a, **k))
        meshgrid._second_amd_list = True
        np.meshgrid = meshgrid
    for alias, typ in (("float", float), ("int", int), ("bool", bool)):  # removed numpy aliases used by 2019 code
        if not hasattr(np, alias):
            setattr(np, alias, typ)
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    spconv_path = spconv_path or here
    if spconv_path not in sys.path:
        sys.path.insert(0, spconv_path)
    if _missing("numba"):
        _install_numba()
    if _missing("fire"):
        _install_fire()
    if _missing("tensorboardX"):
        _install_tensorboardx()
    if _missing("torchvision"):
        _install_torchvision()
    for name in ("cv2", "skimage", "skimage.io", "seaborn", "shapely", "shapely.geometry", "pyquaternion"):
        if _missing(name):
            _stub(name)
    if spawn_workers:
        spawn_dataloader_workers("all")
    if reference_root:
        if reference_root not in sys.path:
            sys.path.append(reference_root)
        return load_protos(reference_root)
    return []


def accelerate_nms():
    """Optional: replace ``second.pytorch.core.box_torch_ops.rotate_nms`` / ``nms`` (box_torch_ops.py:454-515) by
    device-resident equivalents.  The originals copy the candidates to the host, run numba / spconv CPU code there and
    copy the kept indices back -- once per sample and class (voxelnet.py:452-455,578-584).  Same signature, same
    semantics (top-k pre-selection, `rotate_nms_cc` = standup pre-filter and IoU >= thr, `nms_gpu_cc` = +1 convention and
    IoU > thr), same return value (LongTensor of kept indices into the inputs); the only host synchronisation left is the
    size of the result.  CPU tensors keep the original path.  Call after :func:`install`; returns the patched module."""
    import importlib
    import torch
    from .. import ops
    bto = importlib.import_module("second.pytorch.core.box_torch_ops")
    if getattr(bto, "_second_amd_accelerated", False):
        return bto
    orig = {"rotate_nms": bto.rotate_nms, "nms": bto.nms}

    def _run(kind, semantics, boxes, scores, pre_max_size, post_max_size, iou_threshold):
        n = scores.shape[0]
        n_cand = n if pre_max_size is None else min(n, pre_max_size)
        if not (boxes.is_cuda or bto._second_amd_force) or n_cand > 4096:   # sec_nms_sorted_f32 handles <= 4096 boxes
            return orig[kind](boxes, scores, pre_max_size, post_max_size, iou_threshold)
        if n == 0:
            return torch.zeros([0]).long().to(boxes.device)
        if pre_max_size is not None:
            scores, indices = torch.topk(scores, k=min(n, pre_max_size))
        else:
            scores, indices = torch.sort(scores, descending=True)
        dets = torch.cat([boxes[indices], scores.unsqueeze(-1)], dim=1).float().contiguous().unsqueeze(0)
        counts = torch.full((1,), dets.shape[1], dtype=torch.int32, device=boxes.device)
        keep, num = ops.nms_sorted(dets, counts, float(iou_threshold), "rotate" if kind == "rotate_nms" else "axis_aligned",
                                   semantics, post_max=int(post_max_size or 0))
        keep = keep[0, :int(num[0].item())].long()
        return indices[keep]

    def rotate_nms(rbboxes, scores, pre_max_size=None, post_max_size=None, iou_threshold=0.5):
        return _run("rotate_nms", "cpu", rbboxes, scores, pre_max_size, post_max_size, iou_threshold)

    def nms(bboxes, scores, pre_max_size=None, post_max_size=None, iou_threshold=0.5):
        return _run("nms", "numba", bboxes, scores, pre_max_size, post_max_size, iou_threshold)

    bto.rotate_nms, bto.nms = rotate_nms, nms
    bto._second_amd_force = False      # tests: take the device formulation for CPU tensors too (oracle backend)
    bto._second_amd_accelerated = True
    bto._second_amd_original_nms = orig
    return bto


def accelerate_model(net, dtype=None, graph=True, static=None, strict=True, train_dtype=None, fp32_exact=False, deferred=None):
    """Serve a reference-built ``VoxelNet``'s ``net(example)`` in eval mode (voxelnet.py:339-375, called by train.py:524) from the
    fused static-capacity, graph-captured pipeline: parameters adopted by state-dict key, same return value
    (voxelnet.py:616-643), fp32 by default and 16-bit after ``net.half()``.  See :mod:`second_amd.dropin`.  Call after the
    network is built (any time before the first eval batch; later ``load_state_dict`` / ``.half()`` / ``.to()`` are followed).
    ``train_dtype=torch.bfloat16`` (or float16; SEC_ACCELERATE_TRAIN=bf16 in the environment): TRAINING-mode calls
    (train.py:306-325) are served too -- the loss dict of voxelnet.py:299-312 from one hipGraph replay, ``loss.backward()`` from a
    second one that leaves the gradients on the network's own parameters, 16-bit features over the fp32 weights
    (:mod:`second_amd.dropin_train`).  ``fp32_exact=True``: an fp32 network is served with IEEE fp32 products (the reference's
    arithmetic) instead of the default split-operand bf16 passes ("bf16x3", 16 significant bits per operand).  ``deferred=True``
    (SEC_ACCELERATE_DEFERRED=1; what ``second_amd.launch evaluate`` uses): eval-mode calls return at once with dicts that fill
    themselves when read -- evaluate() only collects them while it loops (train.py:519-524) -- and consecutive calls overlap on two
    lanes; anything that reads a result gets exactly what the synchronous call returns."""
    from ..dropin import accelerate_model as _acc
    return _acc(net, dtype=dtype, graph=graph, static=static, strict=strict, train_dtype=train_dtype, fp32_exact=fp32_exact,
                deferred=deferred)


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """Same signature and numpy-in / numpy-out contract as second/core/non_max_suppression/nms_gpu.py:604-640
    (rotate_iou_gpu_eval, a numba.cuda kernel), on ``sec_rotate_iou_f32``: [N,5] x [K,5] (x, y, w, l, r) -> [N,K];
    criterion -1 = IoU, 0 = intersection / area(box), 1 = intersection / area(query), 2 = intersection area."""
    import numpy as np
    import torch
    from .. import ops
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    query_boxes = np.ascontiguousarray(query_boxes, dtype=np.float32)
    n, k = boxes.shape[0], query_boxes.shape[0]
    if n == 0 or k == 0:
        return np.zeros((n, k), dtype=np.float32)
    dev = torch.device("cuda", device_id)
    out = ops.rotate_iou(torch.from_numpy(boxes).to(dev), torch.from_numpy(query_boxes).to(dev), int(criterion))
    return out.cpu().numpy().astype(boxes.dtype)


def accelerate_eval():
    """Route the KITTI evaluation's rotated-IoU calls to the MI355X: ``second/utils/eval.py:124`` (bev_box_overlap) and
    ``:175`` (box3d_overlap) call ``rotate_iou_gpu_eval(boxes, qboxes, criterion)``, a numba.cuda kernel that cannot run
    here; :func:`rotate_iou_gpu_eval` above replaces it (all four criteria; pinned on the original by
    tests/golden/rotate_iou.npz).  Call after :func:`install`; returns the patched module."""
    import importlib
    ev = importlib.import_module("second.utils.eval")
    if getattr(ev, "_second_amd_accelerated", False):
        return ev
    ev._second_amd_original_rotate_iou = ev.rotate_iou_gpu_eval
    ev.rotate_iou_gpu_eval = rotate_iou_gpu_eval
    ev._second_amd_accelerated = True
    return ev
