"""Device-resident training step of the SECOND path, one process per GPU (BASELINE configs 3 / 5).

The reference trains through a worker-fed loop (second/pytorch/train.py:291-329): DataLoader workers voxelise and assign
targets in numpy, ``example_convert_to_torch`` copies padded tensors to the device, ``VoxelNet.forward`` runs network +
loss, then ``loss.backward()``, ``clip_grad_norm_(10)``, optimizer step; multi-GPU = single-process nn.DataParallel.

Here the whole step stays on the GPU and every rank owns one GPU:

    raw points + ground-truth boxes (HBM)
      -> sec_voxelize_f32 (+ SimpleVoxel mean)                         [no worker-side numpy, no padded H2D copies]
      -> SpMiddleFHD in train mode: rulebooks, sec_indice_conv_fwd, BatchNorm1d batch statistics, ReLU (autograd through
         sec_indice_conv_bwd / sec_dense_to_sparse)
      -> RPNV2: 16-bit features: 3x3 convs (forward, dgrad, wgrad) and BatchNorm + ReLU on the hand-written kernels
         (models.rpn_forward_mixed); fp32: torch convolutions (MIOpen's Winograd kernels)
      -> sec_assign_targets_f32 (anchor <-> ground truth, box encoding)
      -> sec_second_loss_f32 (focal + smooth-L1 + direction loss, values and head gradients in one pass)
      -> backward -> ONE all-reduce of the flat gradient bucket over RCCL/xGMI (distributed.GradBucket)
      -> clip_grad_norm_(10) -> AdamW step.

BatchNorm statistics stay per rank (the reference's DataParallel replicas never synchronise them either).
"""
import math
import os

import torch

from . import distributed as D
from . import ops


class FlatAdamW:
    """clip_grad_norm_ + AdamW (train.py:323-325) on ONE flat fp32 buffer in two launches (sec_flat_adamw_f32).

    The parameters become views of ``self.flat`` (same layout as the gradient bucket, whose flat buffer is the gradient), so the
    update needs no per-parameter work: torch.optim.AdamW over the 69 tensors of a SECOND network is ~15 multi-tensor launches plus
    -- in its capturable form -- two one-element ``pow`` launches PER PARAMETER (0.4-0.6 ms of a 4.5 ms captured step).  Same
    formulas as torch (tests/test_gpu_train_dense.py::test_flat_adamw_matches_torch_adamw_with_clipping)."""

    def __init__(self, params, grad_flat, lr, weight_decay, betas=(0.9, 0.99), eps=1e-8, max_grad_norm=10.0):
        from . import runtime as rt
        self.params = list(params)
        assert all(p.dtype == torch.float32 and p.is_cuda for p in self.params), "FlatAdamW: fp32 master weights on the GPU"
        n = sum(p.numel() for p in self.params)
        assert grad_flat.numel() == n and grad_flat.dtype == torch.float32
        dev = grad_flat.device
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:
                v = self.flat[off:off + p.numel()].view_as(p)
                v.copy_(p)
                p.data = v                                   # the module's parameter IS this slice from now on
                off += p.numel()
        self.grad = grad_flat
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.state = torch.zeros(4, dtype=torch.float32, device=dev)         # (gradient norm, step count, skipped flag, loss scale used)
        self.loss_scale = None                                               # device float[4] for fp16: see enable_loss_scaling
        self.ws = torch.zeros(rt.lib().sec_flat_adamw_workspace_bytes(), dtype=torch.uint8, device=dev)
        self.max_grad_norm = float(max_grad_norm)
        # The hyper-parameters live in ONE param group (what torch LR schedulers and checkpoint code attach to) and, for the kernels,
        # in six device floats that :meth:`sync_hyper` refreshes when the group changed: a step captured in a hipGraph follows a
        # schedule (the reference's one-cycle lr changes every step) by replaying with new values, not by re-capturing.
        self.param_groups = [{"params": self.params, "lr": float(lr), "betas": tuple(float(b) for b in betas), "eps": float(eps),
                              "weight_decay": float(weight_decay)}]
        self.hyper = torch.zeros(6, dtype=torch.float32, device=dev)
        self._hyper_last = None
        self.sync_hyper()

    lr = property(lambda self: self.param_groups[0]["lr"], lambda self, v: self.param_groups[0].__setitem__("lr", float(v)))
    weight_decay = property(lambda self: self.param_groups[0]["weight_decay"],
                            lambda self, v: self.param_groups[0].__setitem__("weight_decay", float(v)))
    betas = property(lambda self: self.param_groups[0]["betas"], lambda self, v: self.param_groups[0].__setitem__("betas", tuple(v)))
    eps = property(lambda self: self.param_groups[0]["eps"], lambda self, v: self.param_groups[0].__setitem__("eps", float(v)))

    def set_lr(self, lr):
        self.lr = lr

    def sync_hyper(self):
        """Upload (lr, beta1, beta2, eps, weight_decay, max_grad_norm) if they changed since the last upload.  Called by
        :meth:`step` and by the replay function of a captured step (outside the graph, on the replaying stream)."""
        g = self.param_groups[0]
        cur = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), float(self.max_grad_norm))
        if cur != self._hyper_last:
            # a FRESH pageable source per upload, copied synchronously with respect to the host (24 bytes): the host runs ahead of
            # captured steps, and a reused pinned staging buffer could be rewritten with step k+1's values before step k's
            # asynchronous copy had executed (one-cycle schedules change lr every step)
            self.hyper.copy_(torch.tensor(cur, dtype=torch.float32))
            self._hyper_last = cur

    def enable_loss_scaling(self, init_scale=2.0 ** 12, growth_interval=200):
        """Dynamic loss scaling kept on the device (fp16 features): ``self.loss_scale`` = (scale, clean steps in a row, growth
        interval, skipped steps).  Multiply the loss by ``self.loss_scale[0]`` before backward(); step() unscales, skips the update
        and halves the scale on a non-finite gradient norm, doubles it after ``growth_interval`` clean steps -- no host read."""
        self.loss_scale = torch.tensor([float(init_scale), 0.0, float(growth_interval), 0.0], dtype=torch.float32, device=self.flat.device)
        return self.loss_scale

    def step(self):
        from . import runtime as rt
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        rt.check(rt.lib().sec_flat_adamw_dev_f32(rt.ptr(self.flat), rt.ptr(self.grad), rt.ptr(self.exp_avg), rt.ptr(self.exp_avg_sq),
                                                 self.flat.numel(), rt.ptr(self.hyper), rt.ptr(self.state), rt.ptr(self.loss_scale),
                                                 rt.ptr(self.ws), self.ws.numel(), rt.stream()),
                 "sec_flat_adamw_dev_f32")

    def zero_grad(self, set_to_none=False):
        self.grad.zero_()

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "state": self.state,
                "hyper": {"lr": self.lr, "weight_decay": self.weight_decay, "betas": self.betas, "eps": self.eps}}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"]); self.state.copy_(sd["state"])
        for k, v in sd.get("hyper", {}).items():
            setattr(self, k, v)
        self.sync_hyper()


class DeviceTrainer:
    def __init__(self, det, lr=3e-3, weight_decay=0.01, max_grad_norm=10.0, matched_threshold=None, unmatched_threshold=None,
                 loss_cfg=None, amp_dtype=None, init_loss_scale=2.0 ** 12):
        """det: SecondDetector on this rank's GPU (training mode is set here).  ``amp_dtype`` (torch.bfloat16 / float16):
        mixed precision -- 16-bit features in the sparse stack and the RPN over fp32 master weights; None = fp32 throughout
        (the reference's default training precision); float16 adds dynamic loss scaling (``init_loss_scale``, see
        :meth:`_unscale_and_check`).  Thresholds default to the config's class_settings."""
        from . import models
        self.det = det.train()
        self.cfg = det.cfg
        self.max_grad_norm = float(max_grad_norm)
        mts, uts = list(self.cfg["matched_thresholds"]), list(self.cfg["unmatched_thresholds"])
        if matched_threshold is not None:
            mts = [float(matched_threshold)] * len(mts)
        if unmatched_threshold is not None:
            uts = [float(unmatched_threshold)] * len(uts)
        self.thresholds = (mts[0], uts[0])
        # several anchor generators: per-class matching (all.fhd.config:295) or all boxes with per-anchor thresholds
        # (all.pp.largea.config:269) over the class-major anchor ranges
        self.class_ranges = None
        if len(mts) > 1:
            ids = self.cfg["group_class_ids"] if self.cfg.get("assign_per_class", True) else [0] * len(mts)
            self.class_ranges = (models.anchor_class_ranges(self.cfg, det.feature_map_size), ids, mts, uts)
        self.loss_cfg = dict(ops.LOSS_DEFAULTS, direction_offset=self.cfg["direction_offset"], num_class=self.cfg["num_class"],
                             num_direction_bins=self.cfg["num_direction_bins"], **(loss_cfg or {}))
        self.amp_dtype = amp_dtype
        D.broadcast_parameters(det, 0)
        # the reference's adam_optimizer + fixed weight decay (car.fhd.config:180-188) -> AdamW
        params = [p for p in det.parameters() if p.requires_grad]
        self.bucket = D.GradBucket(det)
        self.flat_opt = bool(params) and all(p.is_cuda and p.dtype == torch.float32 for p in params)
        if self.flat_opt:
            # clip + AdamW as two launches on a flat master-weight buffer (the parameters become views of it); step counter on the
            # device, so a whole optimisation step can be captured in a hipGraph (capture_step)
            self.opt = FlatAdamW(params, self.bucket.flat, lr, weight_decay, betas=(0.9, 0.99), max_grad_norm=self.max_grad_norm)
        else:
            self.opt = torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay, betas=(0.9, 0.99))
        self.graph_rpn = True        # eager steps: the static-shape RPN segment replays as two hipGraphs (_rpn_mixed)
        self.static = False          # static-capacity rows (device-side live counts, no host sync): set by capture_step
        self._captured = None
        # fp16 features need loss scaling (5 exponent bits: head gradients are O(1 / num_pos / batch)); bf16 / fp32 do not
        # flat optimizer: the scale lives on the device (FlatAdamW.enable_loss_scaling: no host read, the step stays capturable);
        # otherwise a host float and one overflow-flag read per step (_unscale_and_check)
        self.loss_scale_dev = self.opt.enable_loss_scaling(init_loss_scale) if (amp_dtype == torch.float16 and self.flat_opt) else None
        self.loss_scale = float(init_loss_scale) if (amp_dtype == torch.float16 and not self.flat_opt) else None
        self._good_steps, self._skipped_host = 0, 0
        self._graphed_rpn = None
        self.fused_head_loss = os.environ.get("SEC_TRAIN_FUSED_HEAD_LOSS", "1") != "0"   # A/B switch (tests compare the two forms)
        self.prepack = os.environ.get("SEC_TRAIN_PREPACK", "1") != "0"                   # weight images of all layers in two launches per step
        self.steps = 0
        self.last = {}

    def forward_loss(self, points, point_offsets, gt_boxes, gt_offsets, gt_classes=None):
        prev = ops.set_rulebook_numbering(self.det.rulebook_numbering)    # row order of the strided layers is internal here
        try:
            return self._forward_loss(points, point_offsets, gt_boxes, gt_offsets, gt_classes)
        finally:
            ops.set_rulebook_numbering(prev)
            ops._PREPACK.clear()      # images the forward did not consume must not outlive this step's weights (prepack_training_weights)

    def _forward_loss(self, points, point_offsets, gt_boxes, gt_offsets, gt_classes=None):
        det, cfg = self.det, self.cfg
        batch = point_offsets.numel() - 1
        static = self.static and not det.pillars and self.amp_dtype is not None
        nd = None
        with torch.no_grad():
            if det.pillars:
                vox = det.voxel_generator.generate_device(points, point_offsets)
            else:
                vox = det.voxel_generator.generate_device(points, point_offsets, mean_features=cfg["num_point_features"],
                                                          mean_dtype=self.amp_dtype, sync=not static)
                if static:
                    nd = vox["voxel_offsets"][batch:]         # live voxel rows stay on the device
            if self.class_ranges is None:
                labels, reg_targets, importance = ops.assign_targets(det.anchors, gt_boxes, gt_offsets, *self.thresholds,
                                                                     gt_classes=gt_classes)
            else:
                begin, ids, mts, uts = self.class_ranges
                labels, reg_targets, importance = ops.assign_targets_per_class(det.anchors, gt_boxes, gt_offsets, gt_classes,
                                                                               begin, ids, mts, uts)
        if det.pillars:
            # PointPillars (nuscenes/all.pp.largea): PillarFeatureNet on sec_pfn_train_fwd / _bwd (batch statistics; the [P, T, C]
            # tensor of the reference formulation is never built), differentiable pillar scatter (sec_pillar_scatter / sec_dense_to_sparse),
            # the three-block RPN through _rpn_mixed (16-bit) or torch convolutions (fp32)
            with torch.autocast("cuda", dtype=self.amp_dtype or torch.float32, enabled=self.amp_dtype is not None):
                feats = det.voxel_feature_extractor(vox["voxels"], vox["num_points_per_voxel"], vox["coordinates"])
                if self.amp_dtype is not None:
                    feats = feats.to(self.amp_dtype)          # the fused PFN returns fp32; the pseudo image is built in 16 bits
                mixed = self.amp_dtype is not None
                # the mixed-precision RPN segment works on channels-last activations: the scatter writes them that way
                spatial = det.middle_feature_extractor(feats.float() if self.amp_dtype is None else feats, vox["coordinates"], batch,
                                                       channels_last=mixed)
                if not mixed:
                    preds = det.rpn(spatial)
            if mixed:
                # the same captured mixed-precision RPN segment as config 3: the 128 -> 128 3x3 layers of the second block on the
                # hand-written kernels, the other widths on MIOpen, forward and backward replayed as two hipGraphs
                preds = self._rpn_mixed(spatial)
        elif self.amp_dtype is not None:
            if self.prepack:
                ops.prepack_training_weights(*self._prepack_items(), self.amp_dtype)      # every layer's weight images: two launches
            # fp32 master weights; 16-bit features through the sparse stack (MFMA forward / dgrad / wgrad kernels) and,
            # under autocast, through the dense RPN; BatchNorm statistics and the loss in fp32
            # (channels_last: the dense scatter writes the RPN's layout and the gradient is gathered from it -- the [B, C, D, H, W]
            # form cost a 36 MB permute copy forward and another one backward, 48 us each)
            spatial = det.middle_feature_extractor(vox["mean"].to(self.amp_dtype), vox["coordinates"], batch, channels_last=True,
                                                   site_table=vox.get("site_table"), num_active_dev=nd)
            # 3x3 convs + BatchNorm/ReLU on the hand-written kernels; the loss from the stacked head tensor where its shape allows
            preds = self._rpn_mixed(spatial, loss_args=(labels, reg_targets, det.anchors, importance, self.loss_cfg) if self.fused_head_loss else None)
            if "loss" in preds:
                return preds["loss"], preds["out6"], labels
        else:
            preds = det.network_forward(vox["mean"], vox["coordinates"], batch, site_table=vox.get("site_table"))
        loss, out6 = ops.SecondLossFunction.apply(preds["cls_preds"], preds["box_preds"], preds.get("dir_cls_preds"), labels,
                                                  reg_targets, det.anchors, importance, self.loss_cfg)
        return loss, out6, labels

    def _prepack_items(self):
        """(sparse_items, dense_items) for ops.prepack_training_weights: the sparse convolutions of the middle extractor and the RPN
        convolutions rpn_forward_mixed runs on the hand-written kernels (3x3 / stride 1 128 -> 128 convs, 1x1 128 -> 128 deblocks)."""
        import spconv
        from .models import RPN_TRAIN_BACKEND
        sparse = [(m.weight, bool(m.subm), m.weight.requires_grad) for m in self.det.middle_feature_extractor.modules()
                  if isinstance(m, spconv.SparseConvolution)]
        dense = []
        if RPN_TRAIN_BACKEND == "hip":
            for m in self.det.rpn.modules():
                if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.bias is None \
                        and (m.in_channels, m.out_channels) == (128, 128):
                    dense.append(m.weight)
                elif isinstance(m, torch.nn.ConvTranspose2d) and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.bias is None \
                        and (m.in_channels, m.out_channels) == (128, 128):
                    dense.append(m.weight)
        return sparse, dense

    def _rpn_mixed(self, spatial, loss_args=None):
        """The dense part of the step has static shapes ([B, 128, H, W] whatever the clouds hold), ~100 launches forward + backward,
        and the eager step is HOST bound (541 launches at ~14 us each = 7.6 ms for 4.9 ms of kernels, profiles/r03_e_*): its
        forward and backward are therefore captured once as two hipGraphs (torch.cuda.make_graphed_callables over the same
        rpn_forward_mixed) and replayed.  ``self.graph_rpn = False``, a changed input shape or a failed capture fall back to eager."""
        import os
        from .models import rpn_forward_mixed
        x = spatial.to(self.amp_dtype).contiguous(memory_format=torch.channels_last)
        if self.static or not self.graph_rpn or not x.is_cuda:
            return rpn_forward_mixed(self.det.rpn, x, self.amp_dtype, loss_args=loss_args)      # static: the WHOLE step is one graph (capture_step)
        key = (tuple(x.shape), x.dtype)
        if self._graphed_rpn is None or self._graphed_rpn[0] != key:
            self._graphed_rpn = (key, self._capture_rpn(x))
        fn = self._graphed_rpn[1]
        if fn is None:
            return rpn_forward_mixed(self.det.rpn, x, self.amp_dtype)
        if not x.requires_grad:
            x = x.detach().requires_grad_()
        box, cls, dirp = fn(x)
        out = {"box_preds": box, "cls_preds": cls}
        if dirp.numel():
            out["dir_cls_preds"] = dirp
        return out

    def _capture_rpn(self, x):
        from .models import rpn_forward_mixed
        rpn, dt = self.det.rpn, self.amp_dtype

        class _Mixed(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.rpn = rpn

            def forward(self, inp):
                p = rpn_forward_mixed(self.rpn, inp, dt)
                return p["box_preds"], p["cls_preds"], p.get("dir_cls_preds", p["cls_preds"].new_zeros(0))
        # the warm-up iterations of the capture run BatchNorm in training mode: keep the running statistics out of it
        saved = {k: v.clone() for k, v in rpn.state_dict().items() if "running_" in k or "num_batches" in k}
        ops._bn_counter_stack.append([[], False])   # throw-away frame: the three non-capturing warm-up iterations of make_graphed_callables
        try:                                    # must not queue num_batches_tracked increments into the step's deferred list
            sample = x.detach().clone().requires_grad_()
            with ops.rt.capture_guard():        # (its captures use the global error mode: a collector run inside would abort the process)
                fn = torch.cuda.make_graphed_callables(_Mixed(), (sample,))
        except Exception as e:  # noqa: BLE001 -- capture is an optimisation: the eager path is always correct
            import warnings
            warnings.warn(f"second_amd: hipGraph capture of the RPN training segment failed ({e!r}); running it eagerly")
            fn = None
        finally:
            ops._bn_counter_stack.pop()
        with torch.no_grad():
            sd = rpn.state_dict()
            for k, v in saved.items():
                sd[k].copy_(v)
        for p in rpn.parameters():
            p.grad = None
        return fn

    def step(self, points, point_offsets, gt_boxes, gt_offsets, gt_classes=None):
        """One optimisation step on this rank's shard; returns the device tensor of the six loss scalars (no host sync,
        except with fp16 features on a non-flat optimizer: dynamic loss scaling then reads one overflow flag per step)."""
        out6 = self._step_backward(points, point_offsets, gt_boxes, gt_offsets, gt_classes)
        self.bucket.allreduce(average=True)                       # one flat bucket, zeros for parameters without a gradient
        if self.loss_scale is not None and not self._unscale_and_check():
            self.bucket.zero_grad()                               # overflow on some rank: skip the step everywhere (same bucket)
            self._skipped_host += 1
            return out6
        self._step_update()
        self.steps += 1
        return out6

    def _step_backward(self, points, point_offsets, gt_boxes, gt_offsets, gt_classes=None):
        with ops.deferred_bn_counters():
            loss, out6, _ = self.forward_loss(points, point_offsets, gt_boxes, gt_offsets, gt_classes)
        if self.loss_scale_dev is not None:
            (loss * self.loss_scale_dev[0]).backward()            # fp16 gradients: scaled (device scalar) so that small ones do not flush to zero
        elif self.loss_scale is None:
            loss.backward()
        else:
            (loss * self.loss_scale).backward()
        self.last = {"out6": out6}
        return out6

    def _step_update(self):
        if not self.flat_opt:
            torch.nn.utils.clip_grad_norm_(self.bucket.params, self.max_grad_norm)
        self.opt.step()                                           # FlatAdamW: clip + update, two launches
        self.bucket.zero_grad()                                   # fp32 p.grad are views of the bucket: one fill

    def capture_step(self, points, point_offsets, gt_boxes, gt_offsets, gt_classes=None, margin=1.25, warmup=2, restore_state=True):
        """The WHOLE optimisation step -- voxelise, targets, forward, loss, backward, gradient all-reduce, clip, AdamW -- as ONE
        hipGraph (second/pytorch/train.py:306-330 is ~520 dispatches here, host bound when launched one by one).  What makes it
        capturable: static-capacity rows through the sparse stack (row counts stay on the device; every strided layer gets
        ``margin`` x the rows this batch produced, rounded up to 256), BatchNorm statistics / dense scatter over the LIVE rows
        only (``rows_dev`` / ``num_dev`` arguments of the kernels), an optimizer whose step counter -- and, for fp16 features, whose
        dynamic loss scale -- live on the device (FlatAdamW).
        Returns ``replay(points=None, point_offsets=None, gt_boxes=None, gt_offsets=None, gt_classes=None) -> out6``: new inputs
        (same shapes or fewer rows) are copied into the graph's buffers first.  Raises what the capture raises: callers fall back
        to :meth:`step`.  Call :meth:`check_overflow` now and then (one host sync).  The hyper-parameters of the flat optimizer are read
        from device memory by the captured kernels: change ``trainer.opt.lr`` (or its ``param_groups``) between replays freely.
        ``restore_state`` (default): model, optimizer and BatchNorm state are snapshotted before and restored after the capture, whose
        1 + ``warmup`` + 1 real steps on the capture batch would otherwise count as training."""
        import spconv
        assert self.amp_dtype is not None and self.loss_scale is None and not self.det.pillars, "capture_step: 16-bit sparse-middle configs on the flat optimizer"
        ins = [points, point_offsets, gt_boxes, gt_offsets, gt_classes]
        names = ("points", "point_offsets", "gt_boxes", "gt_offsets", "gt_classes")
        # the dynamic step, the static warm-up steps and the capture itself are real optimizer steps on the capture batch: the
        # model / optimizer / BatchNorm state is put back afterwards, so that capturing does not train (restore_state=False keeps them)
        snap = None
        if restore_state:
            snap = ({k: v.clone() for k, v in self.det.state_dict().items()},
                    {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in self.opt.state_dict().items()} if self.flat_opt else None,
                    None if self.loss_scale_dev is None else self.loss_scale_dev.clone(), self.steps)
        self.static = False
        self.step(*ins)                                   # one dynamic step: records every strided layer's output rows
        for m in self.det.middle_feature_extractor.modules():
            if isinstance(m, spconv.SparseConvolution) and not m.subm and m.last_num_out is not None:
                m.static_out_rows = int(-(-int(m.last_num_out * margin) // 256) * 256)
        self.static = True
        bufs = [None if t is None else t.clone() for t in ins]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                       # static eager steps: allocator warm-up, optimizer state, autotuned plans
                self.step(*bufs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.check_overflow()
        # Multi-rank: the gradient all-reduce sits INSIDE the graph when RCCL captures (one replay per step, nothing on the host);
        # if that capture fails -- or SEC_TRAIN_ALLREDUCE_IN_GRAPH=0 -- the step becomes two graphs (forward / backward, then clip +
        # AdamW) with the all-reduce issued between them on the same stream.
        in_graph = os.environ.get("SEC_TRAIN_ALLREDUCE_IN_GRAPH", "1") != "0"
        graphs, out6 = None, None
        if in_graph:
            try:
                g = torch.cuda.CUDAGraph()
                with ops.rt.capture_guard(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                    out6 = self.step(*bufs)
                graphs = (g,)
            except Exception as e:  # noqa: BLE001
                if D.world_size() <= 1:
                    raise
                import warnings
                warnings.warn(f"second_amd: capturing the RCCL all-reduce inside the training graph failed ({e!r}); two graphs per step")
                torch.cuda.synchronize()
        if graphs is None:
            pool = torch.cuda.graph_pool_handle()
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            # graph 1 ends with the gradients PACKED into the flat bucket (the copies are captured: which tensors backward handed over
            # is host state of the capture, not of a replay); the collective runs on the bucket between the graphs
            with ops.rt.capture_guard(), torch.cuda.graph(g1, pool=pool, capture_error_mode="thread_local"):
                out6 = self._step_backward(*bufs)
                self.bucket.pack()
            self.bucket.reduce(average=True)
            with ops.rt.capture_guard(), torch.cuda.graph(g2, pool=pool, capture_error_mode="thread_local"):
                self.bucket.unpack()
                self._step_update()
            graphs = (g1, g2)
        self.allreduce_in_graph = len(graphs) == 1
        self._captured = (graphs, bufs, out6)
        if snap is not None:
            with torch.no_grad():
                sd = self.det.state_dict()
                for k, v in snap[0].items():
                    sd[k].copy_(v)
                if snap[1] is not None:
                    self.opt.load_state_dict(snap[1])
                if snap[2] is not None:
                    self.loss_scale_dev.copy_(snap[2])
            self.steps = snap[3]
            self.bucket.zero_grad()

        def replay(*new):
            assert len(new) <= len(bufs), f"replay takes at most {len(bufs)} tensors ({', '.join(names)})"
            for name, dst, src in zip(names, bufs, new):
                if src is None:
                    continue
                if dst is None:
                    raise ValueError(f"capture_step.replay: `{name}` was None when the step was captured; capture with a {name} tensor to feed one")
                if src.data_ptr() == dst.data_ptr():
                    continue
                if src.dim() != dst.dim() or tuple(src.shape[1:]) != tuple(dst.shape[1:]) or src.shape[0] > dst.shape[0]:
                    raise ValueError(f"capture_step.replay: `{name}` has shape {tuple(src.shape)}, the captured buffer holds {tuple(dst.shape)} "
                                     "(same trailing dimensions, at most as many rows)")
                if name in ("point_offsets", "gt_offsets") and src.shape[0] != dst.shape[0]:
                    raise ValueError(f"capture_step.replay: `{name}` must have the captured batch size ({dst.shape[0] - 1} frames)")
                dst[:src.shape[0]].copy_(src, non_blocking=True)
            if self.flat_opt:
                self.opt.sync_hyper()                     # lr / betas / weight decay changed since the last replay: six floats, no re-capture
            graphs[0].replay()
            if len(graphs) == 2:
                self.bucket.reduce(average=True)
                graphs[1].replay()
            self.steps += 1
            self.last = {"out6": out6}
            return out6
        return replay

    def check_overflow(self):
        """Static-capacity steps: raise if a strided layer produced more rows than its capacity (one host sync)."""
        for num, cap in getattr(self.det.middle_feature_extractor, "last_overflow_checks", []):
            raw = int(num[1].item())
            if raw > cap:
                raise RuntimeError(f"static-capacity overflow in training: a strided sparse conv produced {raw} rows, capacity {cap}")

    @property
    def skipped_steps(self):
        """Steps skipped by the dynamic loss scaling (fp16).  Device-side scaling: one host read, here, not in the step."""
        if self.loss_scale_dev is not None:
            return int(self.loss_scale_dev[3].item())
        return self._skipped_host

    def _unscale_and_check(self):
        """Dynamic loss scaling for fp16 features (the reference trains mixed precision through apex amp with
        ``loss_scale_factor``, train.py:209-216,318-322): unscale the reduced bucket; a non-finite value anywhere (every rank
        sees the same reduced bucket, so every rank decides alike) halves the scale and skips the step; 200 clean steps in a
        row double it."""
        flat = self.bucket.flat
        flat.mul_(1.0 / self.loss_scale)
        if bool(torch.isfinite(flat).all().item()):
            self._good_steps += 1
            if self._good_steps >= 200:
                self.loss_scale, self._good_steps = min(self.loss_scale * 2.0, 2.0 ** 24), 0
            return True
        self.loss_scale, self._good_steps = max(self.loss_scale * 0.5, 1.0), 0
        return False

    def loss_dict(self):
        names = ("loss", "cls_loss_reduced", "loc_loss_reduced", "dir_loss_reduced", "cls_pos_loss", "cls_neg_loss")
        v = self.last["out6"].tolist()
        return {k: (x if math.isfinite(x) else float("nan")) for k, x in zip(names, v)}
