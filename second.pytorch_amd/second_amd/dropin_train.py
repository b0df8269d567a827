"""Training-mode ``net(example)`` of a reference-built VoxelNet, served from the captured device step.

What the reference's loop does (second/pytorch/train.py:306-330)::

    ret_dict = net(example_torch)                    # VoxelNet.forward in training mode -> VoxelNet.loss (voxelnet.py:239-312, 339-375)
    loss = ret_dict["loss"].mean()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)
    optimizer.step(); optimizer.zero_grad(); net.update_global_step()
    net.update_metrics(cls_loss_reduced, loc_loss_reduced, cls_preds, labels, cared)

Eagerly that is the module graph: three modules per sparse layer, a host round trip per strided layer, ``.dense()``, MIOpen
convolutions, ~60 torch launches of loss glue, and autograd walking all of it backwards.  :class:`FusedTrainStep` keeps the
loop and the reference's objects -- its parameters, its optimizer, its metrics -- and replaces what happens BETWEEN them:

  * a mirror of the network (:class:`second_amd.models.SecondDetector`, training mode) whose parameters and BatchNorm buffers ARE
    the reference network's tensors (shared objects, not copies): running statistics are updated in place, ``state_dict()`` /
    checkpoints / the optimizer see every change, nothing has to be copied back;
  * forward = copies of the example into static-capacity buffers + ONE hipGraph replay: SimpleVoxel mean -> rulebook chain -> 14
    sparse convs with batch-statistics BatchNorm -> dense scatter (channels last) -> RPN on the hand-written conv / BatchNorm
    kernels -> stacked 1x1 heads -> ``sec_heads_loss_fwd_terms`` -- the six loss scalars AND the per-anchor tensors the reference
    returns (``cls_preds``, ``cls_loss``, ``loc_loss``); 16-bit features over the fp32 parameters (the arithmetic of the
    reference's ``enable_mixed_precision`` mode: 16-bit activations, fp32 master weights, BatchNorm and loss in fp32);
  * ``ret["loss"]`` carries an autograd node whose backward is a SECOND hipGraph replay (loss gradient -> RPN -> sparse stack, every
    weight gradient) that leaves the gradients in ONE flat fp32 bucket; ``p.grad`` of the reference's parameters become views of
    that bucket (no per-parameter launches), accumulate semantics kept (a ``.grad`` that is already there is added to);
  * the bucket is a :class:`second_amd.distributed.GradBucket` registered as ``net._sec_grad_bucket``: ``second_amd.launch``'s
    gradient seam (``clip_grad_norm_`` -> one RCCL all-reduce) finds the gradients already packed.

The reference's ``clip_grad_norm_``, optimizer, one-cycle schedule, ``update_metrics`` and checkpointing run unmodified on the
reference's own tensors.  Anything outside the captured step's reach -- other loss types, several classes per anchor set,
DataParallel-padded examples, non-fp32 parameters, CPU tensors -- keeps the original forward (the engine says why in
``stats["train_fallback_reason"]``).
"""
import warnings

import numpy as np
import torch

from . import ops
from .distributed import GradBucket
from .models import SecondDetector, rpn_forward_mixed

SCALARS = ("loss", "cls_loss_reduced", "loc_loss_reduced", "dir_loss_reduced", "cls_pos_loss", "cls_neg_loss")   # out6 order of the loss kernels


class NotTrainable(NotImplementedError):
    """The network's training forward is outside what the captured step reproduces; the message says which property."""


def train_config(net):
    """Loss settings off the network object (attribute names of voxelnet.py:121-139, losses.py:143-151, 246-256) -> the ``cfg``
    dict of ops.HeadsLossFunction.  Raises :class:`NotTrainable` for settings the fused loss kernel does not implement."""
    why = []
    norm = getattr(net, "_loss_norm_type", None)
    if getattr(norm, "name", str(norm)) != "NormByNumPositives":
        why.append(f"loss_norm_type {getattr(norm, 'name', norm)}")
    cls_f, loc_f = getattr(net, "_cls_loss_ftor", None), getattr(net, "_loc_loss_ftor", None)
    if type(cls_f).__name__ != "SigmoidFocalClassificationLoss":
        why.append(f"classification loss {type(cls_f).__name__}")
    if type(loc_f).__name__ != "WeightedSmoothL1LocalizationLoss" or not getattr(loc_f, "_codewise", True):
        why.append(f"localization loss {type(loc_f).__name__}")
    if not getattr(net, "_encode_rad_error_by_sin", False):
        why.append("encode_rad_error_by_sin off")
    if not getattr(net, "_encode_background_as_zeros", True):
        why.append("background class column")
    if type(getattr(net, "_dir_loss_ftor", None)).__name__ != "WeightedSoftmaxClassificationLoss" and getattr(net, "_use_direction_classifier", False):
        why.append("direction loss other than softmax")
    if why:
        raise NotTrainable("accelerate_model(training): " + "; ".join(why))
    cw = getattr(loc_f, "_code_weights", None)
    cw = [1.0] * 7 if cw is None else [float(v) for v in np.asarray(cw.detach().cpu() if isinstance(cw, torch.Tensor) else cw).reshape(-1)]
    if len(cw) != 7:
        raise NotTrainable(f"accelerate_model(training): {len(cw)} code weights (the 7-value box code only)")
    alpha = cls_f._alpha
    if alpha is None:
        raise NotTrainable("accelerate_model(training): focal loss without alpha")
    return dict(ops.LOSS_DEFAULTS, alpha=float(alpha), gamma=float(cls_f._gamma or 0.0), sigma=float(loc_f._sigma),
                pos_cls_weight=float(net._pos_cls_weight), neg_cls_weight=float(net._neg_cls_weight),
                classification_weight=float(net._cls_loss_weight), localization_weight=float(net._loc_loss_weight),
                direction_loss_weight=float(net._direction_loss_weight), direction_offset=float(net._dir_offset),
                sin_error_factor=float(net._sin_error_factor), code_weights=tuple(cw), num_class=int(net._num_class),
                num_direction_bins=int(net._num_direction_bins) if net._use_direction_classifier else 0)


def _owner(root, dotted):
    mod = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        mod = getattr(mod, p)
    return mod, parts[-1]


def share_state(det, net):
    """Make every parameter and buffer of ``det`` that ``net`` has under the same state-dict key THE SAME tensor object.  Returns
    the keys of ``det`` the network does not have (the caller decides whether that is fatal)."""
    theirs_p, theirs_b = dict(net.named_parameters()), dict(net.named_buffers())
    missing = []
    for name, p in list(det.named_parameters()):
        src = theirs_p.get(name)
        if src is None or tuple(src.shape) != tuple(p.shape):
            missing.append(name)
            continue
        mod, leaf = _owner(det, name)
        mod._parameters[leaf] = src
    for name, b in list(det.named_buffers()):
        src = theirs_b.get(name)
        if src is None or tuple(src.shape) != tuple(b.shape):
            if name.split(".")[0] in ("voxel_feature_extractor", "middle_feature_extractor", "rpn"):
                missing.append(name)
            continue
        mod, leaf = _owner(det, name)
        mod._buffers[leaf] = src
    return missing


class _Replay(torch.autograd.Function):
    """forward: replay the session's forward graph; backward: replay its backward graph and hand the bucket views to ``.grad``."""

    @staticmethod
    def forward(ctx, sess, *params):
        sess.g_fwd.replay()
        ctx.sess = sess
        ctx.serial = sess.serial
        six = sess.scalars.clone()           # the six loss scalars (out6); the session's own tensor is overwritten by the next call
        ctx.mark_non_differentiable(six)
        return six[0].clone(), six

    @staticmethod
    def backward(ctx, g, _g6):
        sess = ctx.sess
        if ctx.serial != sess.serial:
            raise RuntimeError("accelerate_model(training): backward() of an older net(example) call -- the captured step keeps the "
                               "activations of the LATEST forward only (call backward before the next training forward)")
        sess.deposit(g)
        return (None,) * (1 + len(sess.bucket.params))


class _TrainSession:
    """Static buffers + the two captured graphs for one (batch size, row capacity, voxel layout, anchor count)."""

    def __init__(self, eng, batch, cap, vox_shape, vox_dtype, anchors0, strided_caps):
        self.eng, self.batch, self.cap = eng, batch, cap
        dev = anchors0.device
        n_anchor = anchors0.shape[0]
        self.voxels = torch.zeros((cap,) + tuple(vox_shape), dtype=vox_dtype, device=dev)
        self.num_points = torch.ones((cap,), dtype=torch.int32, device=dev)
        self.coors = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
        self.n_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.anchors = anchors0.detach().float().contiguous().clone()
        self.labels = torch.full((batch, n_anchor), -1, dtype=torch.int32, device=dev)
        self.reg_targets = torch.zeros((batch, n_anchor, 7), dtype=torch.float32, device=dev)
        self.importance = torch.ones((batch, n_anchor), dtype=torch.float32, device=dev)
        self.g_static = torch.ones((), dtype=torch.float32, device=dev)
        self.anchor_flag = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.caps = list(strided_caps)
        self.bucket = eng.bucket
        self.serial = 0
        self.g_fwd = self.g_bwd = None
        self.event = torch.cuda.Event()

    # ---- the step body (what both graphs are captured from)
    def _strided(self):
        import spconv
        return [m for m in self.eng.det.middle_feature_extractor.modules() if isinstance(m, spconv.SparseConvolution) and not m.subm]

    def body(self):
        eng, det, dt = self.eng, self.eng.det, self.eng.dtype
        for m, c in zip(self._strided(), self.caps):
            m.static_out_rows = c
        nf = det.cfg["num_point_features"]           # SimpleVoxel (voxel_encoder.py:207-225), summed in fp32 whatever the storage type
        feats = (self.voxels[:, :, :nf].float().sum(1) / self.num_points.float().unsqueeze(1)).to(dt)
        prev = ops.set_rulebook_numbering(det.rulebook_numbering)
        try:
            with ops.deferred_bn_counters():
                if eng.prepack:
                    ops.prepack_training_weights(*eng.prepack_items, dt)
                spatial = det.middle_feature_extractor(feats, self.coors, self.batch, channels_last=True, num_active_dev=self.n_dev)
                x = spatial.to(dt).contiguous(memory_format=torch.channels_last)
                out = rpn_forward_mixed(det.rpn, x, dt, loss_args=(self.labels, self.reg_targets, self.anchors, self.importance, eng.loss_cfg),
                                        loss_terms=True)
        finally:
            ops.set_rulebook_numbering(prev)
            ops._PREPACK.clear()
        if "out6" not in out:
            raise NotTrainable("accelerate_model(training): head shape outside the fused loss kernel (sec_heads_loss_supported)")
        checks = list(getattr(det.middle_feature_extractor, "last_overflow_checks", []))
        out["counters"] = torch.stack([num[1] for num, _ in checks]).int() if checks else torch.zeros((1,), dtype=torch.int32, device=x.device)
        out["limits"] = [int(c) for _, c in checks]
        return out

    def _grads(self, out):
        grads = torch.autograd.grad(out["loss"], self.bucket.params, grad_outputs=self.g_static, allow_unused=True)
        dsts, srcs = [], []
        for v, g in zip(self.bucket.views, grads):
            if g is None:
                v.zero_()
            else:
                dsts.append(v)
                srcs.append(g if g.dtype == v.dtype else g.to(v.dtype))
        if dsts:
            torch._foreach_copy_(dsts, srcs)

    def build(self):
        eng = self.eng
        det = eng.det
        # BatchNorm running statistics / counters must not see the warm-up steps (they run on whatever the buffers hold)
        saved = {k: v.clone() for k, v in det.state_dict().items() if "running_" in k or "num_batches" in k}
        grads_before = self.bucket.flat.clone()      # the warm-up backward passes write the bucket: the caller may hold gradients in it
        pool = torch.cuda.graph_pool_handle()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._grads(self.body())
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.g_fwd, self.g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with ops.rt.capture_guard(), torch.cuda.graph(self.g_fwd, pool=pool, capture_error_mode="thread_local"):
            out = self.body()
        with ops.rt.capture_guard(), torch.cuda.graph(self.g_bwd, pool=pool, capture_error_mode="thread_local"):
            self._grads(out)
        with torch.no_grad():
            sd = det.state_dict()
            for k, v in saved.items():
                sd[k].copy_(v)
            self.bucket.flat.copy_(grads_before)
        self.scalars = out["out6"]
        self.terms = (out["cls_preds"], out["cls_loss"], out["loc_loss"])
        self.counters, self.limits = out["counters"], out["limits"]
        nc = self.counters.numel()
        self.dev_flags = torch.zeros((nc + 1,), dtype=torch.int32, device=self.anchors.device)
        self.host_flags = torch.empty((nc + 1,), dtype=torch.int32, pin_memory=True)
        eng.stats["train_captures"] += 1

    # ---- the backward half
    def deposit(self, g):
        bucket = self.bucket
        aliased = [p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views)]
        keep = None
        if any(aliased):                      # gradients the caller left in the bucket (no zero_grad): accumulate, as autograd would
            keep = bucket.flat.clone()
            if not all(aliased):
                off, stale = 0, []
                for p, a in zip(bucket.params, aliased):
                    if not a:
                        stale.append(keep[off:off + p.numel()])
                    off += p.numel()
                torch._foreach_zero_(stale)
        self.g_static.copy_(g.detach().reshape(()).float())
        self.g_bwd.replay()
        if keep is not None:
            bucket.flat.add_(keep)
        extra_d, extra_s = [], []
        for p, v in zip(bucket.params, bucket.views):
            if p.grad is None:
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                extra_d.append(p.grad)
                extra_s.append(v if p.grad.dtype == v.dtype else v.to(p.grad.dtype))
        if extra_d:
            torch._foreach_add_(extra_d, extra_s)


class FusedTrainStep:
    def __init__(self, net, cfg, dtype=torch.bfloat16, margin=1.5, row_bucket=16384):
        """``cfg``: dropin.model_config(net).  ``dtype``: the feature type of the captured step (torch.bfloat16 / torch.float16) --
        the parameters stay the reference's fp32 tensors."""
        if dtype not in (torch.bfloat16, torch.float16):
            raise NotTrainable("accelerate_model(training): 16-bit features only (train_dtype=torch.bfloat16 / torch.float16)")
        if cfg.get("vfe") == "PillarFeatureNet":
            raise NotTrainable("accelerate_model(training): PointPillars networks train through second_amd.training.DeviceTrainer")
        self.net, self.cfg, self.dtype = net, cfg, dtype
        self.loss_cfg = train_config(net)
        self.margin, self.row_bucket = float(margin), int(row_bucket)
        self.det = self.bucket = None
        self.prepack = True
        self._sessions = {}
        self._pkey = None
        self.stats = {"train_calls": 0, "train_captures": 0, "train_overflow_recaptures": 0, "train_adoptions": 0, "train_fallback_reason": None}

    # ------------------------------------------------------------------ adoption: shared tensors, nothing to copy
    def _params_key(self):
        ps = list(self.net.parameters())
        # object identity + storage address + dtype: an optimizer that swaps `p.data` for another tensor, `.half()`, `.to()` all show
        return tuple((id(p), p.data_ptr(), p.dtype, p.requires_grad) for p in ps)

    def refresh(self):
        key = self._params_key()
        if self.det is not None and key == self._pkey:
            return self.det
        net = self.net
        ps = [p for p in net.parameters()]
        if not ps or any(p.dtype != torch.float32 or not p.is_cuda for p in ps):
            raise NotTrainable("accelerate_model(training): fp32 parameters on the GPU (the captured step keeps 16-bit features over fp32 weights)")
        det = SecondDetector(self.cfg)
        missing = share_state(det, net)
        if missing:
            raise NotTrainable(f"accelerate_model(training): the network lacks {missing[:4]}")
        det.anchors = None
        det = det.train()
        for m in det.modules():            # the mirror's own non-shared leftovers (its anchors / index buffers) live where the weights do
            for k, b in list(m._buffers.items()):
                if b is not None and b.device != ps[0].device:
                    m._buffers[k] = b.to(ps[0].device)
        self.det = det
        self.bucket = GradBucket(net, track_presence=True)
        net._sec_grad_bucket = self.bucket             # second_amd.launch's all-reduce seam reuses it: the gradients are already packed
        import spconv
        from .models import RPN_TRAIN_BACKEND
        sparse = [(m.weight, bool(m.subm), m.weight.requires_grad) for m in det.middle_feature_extractor.modules()
                  if isinstance(m, spconv.SparseConvolution)]
        dense = []
        if RPN_TRAIN_BACKEND == "hip":
            for m in det.rpn.modules():
                if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.bias is None \
                        and (m.in_channels, m.out_channels) == (128, 128):
                    dense.append(m.weight)
                elif isinstance(m, torch.nn.ConvTranspose2d) and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.bias is None \
                        and (m.in_channels, m.out_channels) == (128, 128):
                    dense.append(m.weight)
        self.prepack_items = (sparse, dense)
        self._pkey = key
        self._sessions.clear()
        self.stats["train_adoptions"] += 1
        return det

    # ------------------------------------------------------------------ dispatch
    def accepts(self, example):
        for k in ("voxels", "num_points", "coordinates", "anchors", "labels", "reg_targets"):
            v = example.get(k)
            if not isinstance(v, torch.Tensor) or not v.is_cuda:
                return False
        return (example["num_points"].dim() == 1 and "anchors_mask" not in example and example["voxels"].shape[0] > 0
                and example["labels"].dim() == 2 and torch.is_grad_enabled())

    def _calibrate(self, det, example, batch):
        """Capacities of the strided layers from one dynamic-shape pass of this example through the sparse stack (live outputs x
        margin, 256-row granules).  Eval mode for that pass: BatchNorm statistics untouched."""
        import spconv
        mid = det.middle_feature_extractor
        was = mid.training
        prev = ops.set_rulebook_numbering(det.rulebook_numbering)
        try:
            mid.eval()
            with torch.no_grad():
                nf = det.cfg["num_point_features"]
                feats = example["voxels"][:, :, :nf].float().sum(1) / example["num_points"].float().unsqueeze(1)
                mid(feats, example["coordinates"].int(), batch)
        finally:
            mid.train(was)
            ops.set_rulebook_numbering(prev)
        return [int(-(-int(m.last_num_out * self.margin) // 256) * 256) for m in mid.modules()
                if isinstance(m, spconv.SparseConvolution) and not m.subm]

    def _session(self, det, example, batch):
        voxels, anchors = example["voxels"], example["anchors"].reshape(batch, -1, 7)
        n = voxels.shape[0]
        key = (batch, tuple(voxels.shape[1:]), voxels.dtype, voxels.device, anchors.shape[1])
        sess = self._sessions.get(key)
        if sess is not None and sess.cap >= n:
            return sess
        cap = -(-int(n * self.margin) // self.row_bucket) * self.row_bucket
        caps = self._calibrate(det, example, batch)
        if sess is not None:
            caps = [max(a, b) for a, b in zip(caps, sess.caps)]
        new = _TrainSession(self, batch, cap, voxels.shape[1:], voxels.dtype, anchors[0], caps)
        self._fill(new, example)
        new.build()
        self._sessions[key] = new
        return new

    @staticmethod
    def _fill(sess, example):
        n = example["voxels"].shape[0]
        sess.voxels[:n].copy_(example["voxels"], non_blocking=True)
        sess.num_points[:n].copy_(example["num_points"], non_blocking=True)
        sess.coors[:n].copy_(example["coordinates"], non_blocking=True)
        sess.n_dev.fill_(n)
        sess.labels.copy_(example["labels"], non_blocking=True)
        sess.reg_targets.copy_(example["reg_targets"], non_blocking=True)
        imp = example.get("importance")
        if imp is None:
            sess.importance.fill_(1.0)
        else:
            sess.importance.copy_(imp, non_blocking=True)

    def __call__(self, example):
        """-> the reference's loss dict (voxelnet.py:299-312), or None when this example has to take the original forward."""
        det = self.refresh()
        batch = example["anchors"].shape[0]
        anchors = example["anchors"].reshape(batch, -1, 7)
        for attempt in range(4):
            sess = self._session(det, example, batch)
            self._fill(sess, example)
            nc = sess.counters.numel()
            anc = anchors if (anchors.dtype == torch.float32 and anchors.is_contiguous()) else anchors.float().contiguous()
            sess.anchor_flag.zero_()
            ops.rows_differ_(sess.anchor_flag, anc, sess.anchors)               # one pass over the example's anchors
            sess.dev_flags[nc:].copy_(sess.anchor_flag)
            sess.serial += 1
            loss, scalars = _Replay.apply(sess, *sess.bucket.params)
            sess.dev_flags[:nc].copy_(sess.counters.reshape(-1))
            sess.host_flags.copy_(sess.dev_flags, non_blocking=True)
            sess.event.record()
            sess.event.synchronize()
            cnt = sess.host_flags.numpy()
            if cnt[nc]:                                   # another anchor table than the session's
                if not bool((anchors == anchors[:1]).all().item()):
                    self.stats["train_fallback_reason"] = "per-frame anchor sets"
                    return None
                sess.anchors.copy_(anchors[0])
                self._restore_bn(sess)
                continue
            over = [int(r) for r, c in zip(cnt[:nc], sess.limits) if int(r) > c]
            if over:                                      # a strided layer outgrew its capacity: size it from the raw counts, recapture, redo
                sess.caps = [max(c, int(-(-int(int(r) * self.margin) // 256) * 256)) for r, c in zip(cnt[:nc], sess.caps)]
                self._restore_bn(sess)
                sess.build()
                self.stats["train_overflow_recaptures"] += 1
                continue
            break
        else:
            raise RuntimeError("accelerate_model(training): static capacities did not settle after four attempts")
        self.stats["train_calls"] += 1
        s = dict(zip(SCALARS, scalars.unbind(0)))
        s["loss"] = loss
        cls_preds, cls_loss, loc_loss = (t.clone() for t in sess.terms)
        a, h, w = int(det.num_anchor_per_loc), *[int(v) for v in det.feature_map_size[1:]]
        res = {"loss": s["loss"], "cls_loss": cls_loss, "loc_loss": loc_loss, "cls_pos_loss": s["cls_pos_loss"],
               "cls_neg_loss": s["cls_neg_loss"], "cls_preds": cls_preds.view(batch, a, h, w, -1),
               "cls_loss_reduced": s["cls_loss_reduced"], "loc_loss_reduced": s["loc_loss_reduced"],
               "cared": example["labels"] >= 0}
        if self.loss_cfg["num_direction_bins"]:
            res["dir_loss_reduced"] = s["dir_loss_reduced"]
        return res

    def eager(self, example):
        """The same step WITHOUT static capacities and graphs: dynamic-shape launches one by one on the mirror, ordinary autograd
        (gradients reach the shared parameters through AccumulateGrad).  Debugging aid and the test's yardstick for the captured
        form: same kernels, same precision."""
        det, dt = self.refresh(), self.dtype
        batch = example["anchors"].shape[0]
        anchors = example["anchors"].reshape(batch, -1, 7)[0].float().contiguous()
        nf = det.cfg["num_point_features"]
        feats = (example["voxels"][:, :, :nf].float().sum(1) / example["num_points"].float().unsqueeze(1)).to(dt)
        imp = example.get("importance")
        imp = torch.ones_like(example["reg_targets"][..., 0]) if imp is None else imp
        prev = ops.set_rulebook_numbering(det.rulebook_numbering)
        try:
            with ops.deferred_bn_counters():
                spatial = det.middle_feature_extractor(feats, example["coordinates"].int(), batch, channels_last=True)
                x = spatial.to(dt).contiguous(memory_format=torch.channels_last)
                out = rpn_forward_mixed(det.rpn, x, dt, loss_terms=True,
                                        loss_args=(example["labels"].int().contiguous(), example["reg_targets"].float().contiguous(), anchors,
                                                   imp.float().contiguous(), self.loss_cfg))
        finally:
            ops.set_rulebook_numbering(prev)
        s = dict(zip(SCALARS, out["out6"].unbind(0)))
        a, h, w = int(det.num_anchor_per_loc), *[int(v) for v in det.feature_map_size[1:]]
        res = {"loss": out["loss"], "cls_loss": out["cls_loss"], "loc_loss": out["loc_loss"], "cls_pos_loss": s["cls_pos_loss"],
               "cls_neg_loss": s["cls_neg_loss"], "cls_preds": out["cls_preds"].view(batch, a, h, w, -1),
               "cls_loss_reduced": s["cls_loss_reduced"], "loc_loss_reduced": s["loc_loss_reduced"], "cared": example["labels"] >= 0}
        if self.loss_cfg["num_direction_bins"]:
            res["dir_loss_reduced"] = s["dir_loss_reduced"]
        return res

    def _restore_bn(self, sess):
        """A forward that has to be redone already moved the BatchNorm running statistics once: the redo moves them a second time
        with the same batch.  With momentum 0.01 that is a 1 % over-weighting of one batch, once per capacity growth; the counters
        are put back so that num_batches_tracked stays the number of optimisation steps."""
        with torch.no_grad():
            for m in self.det.modules():
                if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.num_batches_tracked is not None:
                    m.num_batches_tracked -= 1


def warn_once(msg, _seen=set()):
    if msg not in _seen:
        _seen.add(msg)
        warnings.warn(msg)
