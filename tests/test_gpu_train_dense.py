"""-m gpu parity of the dense-RPN TRAINING kernels (csrc/dense_train.hip + the data gradient on k_conv2d_halo_reg) against a plain
PyTorch fp32 reference of the same operator on the same 16-bit-rounded operands (nn.Conv2d(128, 128, 3, padding=1, bias=False) +
nn.BatchNorm2d(eps 1e-3, momentum 0.01) + nn.ReLU: second/pytorch/models/rpn.py:486-497 under loss.backward(),
second/pytorch/train.py:316-322), and of ONE whole device-resident training step against the CPU chain (oracle sparse ops + torch
CPU modules) with the golden-pinned loss kernel (VERDICT r2 missing #1, #5)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from second_amd import ops
    return ops


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("form", ["tap", "row"])
@pytest.mark.parametrize("shape", [(1, 13, 17), (2, 21, 12), (2, 40, 48), (4, 200, 176)])
def test_conv2d_wgrad_vs_torch(ops, dtype, shape, form, monkeypatch):
    """dW of the 3x3 / s1 / p1 128 -> 128 conv: ragged pixel counts (221 pixels: not a multiple of the 64-pixel step), a map narrower
    than the 16-pixel stride of the kernel's staging (its NARROW instantiation), a mid size and
    the car.fhd training shape (batch 4 x 200 x 176 = 140 800 pixels); vs torch autograd in fp32 on the same rounded operands; the
    fixed-order reduction makes it run-to-run identical."""
    # both forms of the kernel on every shape: a tap per workgroup (k_conv2d_wgrad3x3) and a kernel row per workgroup
    # (k_conv2d_wgrad3x3_row: the automatic choice from ~160 k pixels; maps narrower than 16 pixels always take the tap form)
    monkeypatch.setenv("SEC_WGRAD_ROW", "1" if form == "row" else "0")
    b, h, w = shape
    g = torch.Generator().manual_seed(b * 1000 + h)
    x = _cl(torch.randn(b, 128, h, w, generator=g).cuda().to(dtype))
    dy = _cl((torch.randn(b, 128, h, w, generator=g) / 8).cuda().to(dtype))
    dw = ops.conv2d_wgrad(x, dy)
    assert dw.shape == (128, 128, 3, 3) and dw.dtype == torch.float32
    wt = torch.zeros(128, 128, 3, 3, device="cuda", requires_grad=True)
    F.conv2d(x.float(), wt, padding=1).backward(dy.float())
    ref = wt.grad
    err = (dw - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-3, err                                   # fp32 accumulation of exact 16-bit products, different summation orders
    assert torch.equal(dw, ops.conv2d_wgrad(x, dy))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv3x3_function_forward_dgrad_wgrad_vs_torch(ops, dtype):
    g = torch.Generator().manual_seed(3)
    b, h, w = 2, 56, 48
    x = _cl(torch.randn(b, 128, h, w, generator=g).cuda().to(dtype)).requires_grad_()
    wgt = (torch.randn(128, 128, 3, 3, generator=g) / 34).cuda().requires_grad_()          # fp32 master weight
    dy = _cl((torch.randn(b, 128, h, w, generator=g) / 8).cuda().to(dtype))
    y = ops.Conv3x3Function.apply(x, wgt)
    y.backward(dy)
    x32 = x.detach().float().requires_grad_()
    w32 = wgt.detach().to(dtype).float().requires_grad_()    # the kernel multiplies the 16-bit rounding of the master weight
    y32 = F.conv2d(x32, w32, padding=1)
    y32.backward(dy.float())
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10   # one rounding of the 16-bit result
    for name, got, ref in (("y", y, y32), ("dx", x.grad, x32.grad)):
        assert got.dtype == dtype and got.is_contiguous(memory_format=torch.channels_last), name
        torch.testing.assert_close(got.float(), ref.detach(), rtol=tol, atol=tol * ref.abs().max().item(), msg=name)
    assert wgt.grad.dtype == torch.float32
    assert (wgt.grad - w32.grad).abs().max().item() < 2e-3 * w32.grad.abs().max().item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("c,relu", [(128, True), (64, True), (256, False)])
def test_bn_relu_forward_backward_vs_torch(ops, dtype, c, relu):
    g = torch.Generator().manual_seed(c)
    b, h, w = 3, 25, 31                                     # 2325 pixels: ragged against every workgroup stride
    y = _cl((torch.randn(b, c, h, w, generator=g) * 2 + torch.randn(1, c, 1, 1, generator=g)).cuda().to(dtype))
    gamma = torch.rand(c, generator=g).cuda() + 0.5
    beta = (torch.randn(c, generator=g) / 4).cuda()
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    dz = _cl(torch.randn(b, c, h, w, generator=g).cuda().to(dtype))
    z, mean, invstd = ops.bn_relu_forward(y, gamma, beta, 1e-3, 0.01, rm, rv, relu)
    dy, dgamma, dbeta = ops.bn_relu_backward(dz, y, gamma, beta, mean, invstd, relu)
    # reference: torch's batch_norm in fp32 on the same rounded input
    y32 = y.float().requires_grad_()
    g32, b32 = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rm2, rv2 = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    z32 = F.batch_norm(y32, rm2, rv2, g32, b32, True, 0.01, 1e-3)
    if relu:
        z32 = F.relu(z32)
    z32.backward(dz.float())
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
    torch.testing.assert_close(z.float(), z32.detach(), rtol=tol, atol=tol * z32.abs().max().item())
    torch.testing.assert_close(rm, rm2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv, rv2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dgamma, g32.grad, rtol=1e-3, atol=1e-3 * g32.grad.abs().max().item())
    torch.testing.assert_close(dbeta, b32.grad, rtol=1e-3, atol=1e-3 * b32.grad.abs().max().item())
    # dy: elements whose pre-activation sits within rounding of zero may take the other side of the ReLU mask
    d = (dy.float() - y32.grad).abs()
    bad = d > tol * y32.grad.abs().max().item() + tol * y32.grad.abs()
    assert bad.float().mean().item() < 1e-3, bad.float().mean().item()
    assert torch.equal(z, ops.bn_relu_forward(y, gamma, beta, 1e-3, 0.01, None, None, relu)[0])    # deterministic


def test_rpn_forward_mixed_hip_vs_fp32_and_torch_autocast(ops, monkeypatch):
    """models.rpn_forward_mixed on the car.fhd RPN in training mode: the hand-written bf16 path and torch's bf16 autocast path
    (models.RPN_TRAIN_BACKEND = "miopen") are two different 16-bit chains, so each is measured against the SAME network in fp32 (plain
    torch): the hand-written path may not be further from fp32 than autocast is (x 1.5 + a small floor), for the head outputs and
    for every parameter gradient; running statistics agree with fp32's."""
    from second_amd.models import RPNV2, rpn_forward_mixed
    torch.manual_seed(0)
    nets = [RPNV2().cuda().train() for _ in range(3)]
    for n in nets[1:]:
        n.load_state_dict(nets[0].state_dict())
    x = (torch.randn(2, 128, 48, 40, device="cuda") * (torch.rand(2, 1, 48, 40, device="cuda") < 0.3)).contiguous(memory_format=torch.channels_last)

    def loss_of(preds):
        return sum((p.float() ** 2).mean() for p in preds.values())
    from second_amd import models
    monkeypatch.setattr(models, "RPN_TRAIN_BACKEND", "hip")
    a = rpn_forward_mixed(nets[0], x, torch.bfloat16)
    loss_of(a).backward()
    monkeypatch.setattr(models, "RPN_TRAIN_BACKEND", "miopen")
    b = rpn_forward_mixed(nets[1], x, torch.bfloat16)
    loss_of(b).backward()
    r = nets[2](x)                                            # fp32 reference
    loss_of(r).backward()

    def rel(u, v):
        return (u.float() - v.float()).abs().max().item() / (v.float().abs().max().item() + 1e-20)
    for k in a:
        assert rel(a[k], r[k]) <= 1.5 * rel(b[k], r[k]) + 0.01, (k, rel(a[k], r[k]), rel(b[k], r[k]))
    worst = 0.0
    for (n, p), (_, q), (_, f) in zip(*(net.named_parameters() for net in nets)):
        assert p.grad is not None and q.grad is not None and f.grad is not None, n
        eh, ea = rel(p.grad, f.grad), rel(q.grad, f.grad)
        worst = max(worst, eh)
        # two 16-bit chains against the fp32 one: the BatchNorm-bias gradients are sums over 140 k pixels of cancelling terms, their
        # relative error moves by a few percent from run to run (MIOpen's strided / transposed convs of the block accumulate with
        # atomics), so the bound is a factor, not a match
        assert eh <= 2.0 * ea + 0.03, (n, eh, ea)
    assert worst < 0.5
    for (n, p), (_, f) in zip(nets[0].named_buffers(), nets[2].named_buffers()):
        if p.is_floating_point():
            torch.testing.assert_close(p, f, rtol=2e-2, atol=2e-3, msg=n)     # running statistics
        else:
            assert torch.equal(p, f), n                                         # num_batches_tracked


def test_device_trainer_step_matches_cpu_chain():
    """ONE whole training step of the device path (fp32: the reference's training precision) against the CPU chain on the same
    frame: oracle voxeliser / rulebooks / indice_conv forward AND backward + torch-CPU BatchNorm / RPN modules in train mode
    (tests/oracle_backend.py), target assignment and loss through the kernels that tests/test_gpu_train.py pins to the reference's
    own create_target_np / VoxelNet.loss outputs.  Compared: the six loss scalars and the gradients of parameters from every part of
    the network (first / middle / last sparse conv, BatchNorm1d affine, first and last RPN conv, BatchNorm2d, deblock, heads)."""
    import oracle_backend
    from second_amd import ops, synthetic as syn
    from second_amd.models import SecondDetector, CAR_FHD
    from second_amd.training import DeviceTrainer
    torch.manual_seed(0)
    det = SecondDetector(CAR_FHD)                            # default init: train-mode BatchNorm keeps every layer's activations O(1)
    cpu = SecondDetector(CAR_FHD)
    cpu.load_state_dict(det.state_dict())
    det = det.cuda()
    cloud = syn.syn_kitti_cloud(0)
    pts, offs = syn.batch_clouds([cloud])
    gt = syn.syn_kitti_boxes(0, 12).astype(np.float32)
    goffs = np.array([0, len(gt)], np.int32)
    tr = DeviceTrainer(det)
    d = lambda a: torch.from_numpy(a).cuda()
    loss, out6, labels = tr.forward_loss(d(pts), d(offs), d(gt), d(goffs))
    loss.backward()
    torch.cuda.synchronize()
    assert int((labels > 0).sum()) >= 10
    # CPU chain: the same modules, sparse ops through the oracle; targets / loss through the (golden-pinned) device kernels
    cpu.train()
    with oracle_backend.installed():
        vox = ops.voxelize(torch.from_numpy(pts), torch.from_numpy(offs), CAR_FHD["point_cloud_range"], CAR_FHD["voxel_size"],
                           CAR_FHD["max_points_per_voxel"], CAR_FHD["max_voxels"], mean_features=4)
        preds = cpu.network_forward(vox["mean"], vox["coordinates"], 1)
        lab, reg, imp = ops.assign_targets(det.anchors, d(gt), d(goffs), *tr.thresholds)
        heads = {k: v.detach().cuda().requires_grad_() for k, v in preds.items()}
        loss_c, out6_c = ops.SecondLossFunction.apply(heads["cls_preds"], heads["box_preds"], heads["dir_cls_preds"], lab, reg, det.anchors,
                                                      imp, tr.loss_cfg)
        loss_c.backward()
        torch.autograd.backward([preds[k] for k in ("cls_preds", "box_preds", "dir_cls_preds")],
                                [heads[k].grad.cpu() for k in ("cls_preds", "box_preds", "dir_cls_preds")])
    np.testing.assert_allclose(out6.cpu().numpy(), out6_c.cpu().numpy(), rtol=2e-3, atol=1e-5)
    names = ["middle_feature_extractor.middle_conv.0.weight", "middle_feature_extractor.middle_conv.1.weight",
             "middle_feature_extractor.middle_conv.18.weight", "middle_feature_extractor.middle_conv.39.weight",
             "rpn.blocks.0.1.weight", "rpn.blocks.0.2.bias", "rpn.blocks.0.16.weight", "rpn.deblocks.0.0.weight", "rpn.conv_cls.weight",
             "rpn.conv_box.bias", "rpn.conv_dir_cls.weight"]
    gp, cp = dict(det.named_parameters()), dict(cpu.named_parameters())
    for n in names:
        a, b = gp[n].grad.float().cpu(), cp[n].grad.float()
        rel = (a - b).abs().max().item() / (b.abs().max().item() + 1e-20)
        # fp32 on both sides; MIOpen vs torch-CPU convolutions and (round 4) the fp32 sparse convs on v_mfma_f32_32x32x2_f32, whose
        # summation order differs from the CPU loop's: 20 layers deep with train-mode BatchNorm, last-bit differences move ReLU
        # masks (measured 0.5-2.0 % on these parameters; the first BatchNorm weight, the deepest one, sits at the top of that range)
        assert rel < 4e-2, (n, rel)


@pytest.mark.parametrize("c", [16, 32, 64])
def test_bn_relu_on_sparse_rows_vs_torch(ops, c):
    """The same kernels on the [N, C] feature rows of a sparse tensor (BatchNorm1d + ReLU of SpMiddleFHD in training, middle.py:146-189)."""
    g = torch.Generator().manual_seed(c)
    n = 5003
    y = (torch.randn(n, c, generator=g) * 3 + 0.5).cuda().bfloat16()
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) / 4).cuda()
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    y_in = y.clone().requires_grad_()
    gp, bp = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    z = ops.BatchNormReluFunction.apply(y_in, gp, bp, rm, rv, 1e-3, 0.01, True)
    dz = torch.randn(n, c, generator=g).cuda().bfloat16()
    z.backward(dz)
    y32, g32, b32 = y.float().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rm2, rv2 = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    z32 = F.relu(F.batch_norm(y32, rm2, rv2, g32, b32, True, 0.01, 1e-3))
    z32.backward(dz.float())
    tol = 2 ** -7
    torch.testing.assert_close(z.float(), z32.detach(), rtol=tol, atol=tol * z32.abs().max().item())
    torch.testing.assert_close(rm, rm2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv, rv2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gp.grad, g32.grad, rtol=1e-3, atol=1e-3 * g32.grad.abs().max().item())
    torch.testing.assert_close(bp.grad, b32.grad, rtol=1e-3, atol=1e-3 * b32.grad.abs().max().item())
    d = (y_in.grad.float() - y32.grad).abs()
    assert (d > tol * y32.grad.abs().max().item() + tol * y32.grad.abs()).float().mean().item() < 1e-3


def test_device_trainer_bf16_step_uses_the_hand_written_path_and_tracks_fp32():
    """One bf16 training step (sparse stack + RPN on the hand-written forward / dgrad / wgrad / BatchNorm kernels) next to the
    fp32 step of the same network on the same frames: the six loss scalars within a few percent, every parameter receives a finite
    gradient, the op trace shows the hand-written RPN convolution (not MIOpen) in the step."""
    from second_amd import ops, synthetic as syn
    from second_amd.models import SecondDetector, CAR_FHD
    from second_amd.training import DeviceTrainer
    clouds = [syn.syn_kitti_cloud(s) for s in range(2)]
    pts, offs = syn.batch_clouds(clouds)
    gt = np.concatenate([syn.syn_kitti_boxes(s, 12) for s in range(2)]).astype(np.float32)
    goffs = np.array([0, 12, 24], np.int32)
    d = lambda a: torch.from_numpy(a).cuda()
    torch.manual_seed(0)
    a = SecondDetector(CAR_FHD).cuda()
    b = SecondDetector(CAR_FHD).cuda()
    b.load_state_dict(a.state_dict())
    t32, t16 = DeviceTrainer(a), DeviceTrainer(b, amp_dtype=torch.bfloat16)
    l32, o32, _ = t32.forward_loss(d(pts), d(offs), d(gt), d(goffs))
    l32.backward()
    seen = []
    ops.set_op_hook(lambda name, fn, args, kw, res: seen.append(name))
    try:
        l16, o16, _ = t16.forward_loss(d(pts), d(offs), d(gt), d(goffs))
        l16.backward()
    finally:
        ops.set_op_hook(None)
    assert seen.count("conv2d_nhwc") >= 12, seen.count("conv2d_nhwc")      # 6 forward + 6 data-gradient launches of the 3x3 kernel
    np.testing.assert_allclose(o16.float().cpu().numpy(), o32.cpu().numpy(), rtol=0.05, atol=1e-3)
    for (n, p), (_, q) in zip(b.named_parameters(), a.named_parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        assert q.grad is not None, n


def test_device_trainer_graphed_rpn_segment_equals_eager(monkeypatch):
    """DeviceTrainer captures the static-shape RPN segment of the bf16 step (forward and backward) into hipGraphs.  From the same
    initial state on the same frames, with the capture and without: the gradients of one forward / backward agree (same kernels in
    the same order; what differs is the launch mechanism and the fp32-atomic summation order of the sparse weight gradients), and
    three optimisation steps run in both modes.  (Parameters / later losses are not compared: AdamW's g / sqrt(v) turns the last
    bit of a near-zero gradient into a full-size update.)"""
    from second_amd import synthetic as syn
    from second_amd.models import SecondDetector, CAR_FHD
    from second_amd.training import DeviceTrainer
    clouds = [syn.syn_kitti_cloud(s) for s in range(2)]
    pts, offs = syn.batch_clouds(clouds)
    gt = np.concatenate([syn.syn_kitti_boxes(s, 12) for s in range(2)]).astype(np.float32)
    goffs = np.array([0, 12, 24], np.int32)
    d = lambda a: torch.from_numpy(a).cuda()
    torch.manual_seed(0)
    init = SecondDetector(CAR_FHD).state_dict()
    results = {}
    for mode in ("1", "0"):
        det = SecondDetector(CAR_FHD)
        det.load_state_dict(init)
        tr = DeviceTrainer(det.cuda(), amp_dtype=torch.bfloat16)
        tr.graph_rpn = mode == "1"
        loss, out6, _ = tr.forward_loss(d(pts), d(offs), d(gt), d(goffs))
        loss.backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().float().cpu().clone() for n, p in det.named_parameters()}
        for p in det.parameters():
            p.grad = None
        losses = [tr.step(d(pts), d(offs), d(gt), d(goffs)).clone() for _ in range(3)]
        torch.cuda.synchronize()
        if mode == "1":
            assert tr._graphed_rpn is not None and tr._graphed_rpn[1] is not None, "the capture fell back to eager"
        results[mode] = (out6.cpu(), grads, torch.stack(losses).cpu())
    torch.testing.assert_close(results["1"][0], results["0"][0], rtol=1e-3, atol=1e-5)
    for n, g in results["1"][1].items():
        w = results["0"][1][n]
        assert (g - w).abs().max().item() <= 2e-2 * w.abs().max().item() + 1e-7, n
    # three optimisation steps run in both modes; their FIRST losses (same parameters, same frames) agree -- the later ones are not
    # compared: the first AdamW steps of a freshly initialised network at lr 3e-3 amplify last-bit differences chaotically
    la, lb = results["1"][2], results["0"][2]
    assert torch.isfinite(la).all() and torch.isfinite(lb).all()
    torch.testing.assert_close(la[0], lb[0], rtol=2e-3, atol=1e-5)


@pytest.mark.parametrize("c", [16, 64])
def test_bn_relu_static_capacity_rows_equal_the_live_slice(ops, c):
    """rows_dev (static-capacity training): statistics, normalisation and the backward sums over the first rows_dev[0] rows only --
    bit-identical to the same call on the live slice, whatever the rows behind it hold (NaN / Inf garbage here); those rows are
    neither read nor written."""
    g = torch.Generator().manual_seed(c + 1)
    n, cap = 3001, 4096
    y = torch.full((cap, c), float("nan")).bfloat16()
    y[:n] = (torch.randn(n, c, generator=g) * 2 + 0.3).bfloat16()
    y[n + 5:] = float("inf")
    y = y.cuda()
    dz = torch.full((cap, c), float("nan")).bfloat16()
    dz[:n] = torch.randn(n, c, generator=g).bfloat16()
    dz = dz.cuda()
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) / 4).cuda()
    rows = torch.tensor([n], dtype=torch.int32, device="cuda")
    res = []
    for yy, dd, rd in ((y, dz, rows), (y[:n].contiguous(), dz[:n].contiguous(), None)):
        rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        yi, gp, bp = yy.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
        z = ops.BatchNormReluFunction.apply(yi, gp, bp, rm, rv, 1e-3, 0.01, True, rd)
        z.backward(dd)
        res.append((z[:n].clone(), yi.grad[:n].clone(), gp.grad.clone(), bp.grad.clone(), rm, rv))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def _train_inputs(frames=2):
    from second_amd import synthetic as syn
    clouds = [syn.syn_kitti_cloud(s) for s in range(frames)]
    pts, offs = syn.batch_clouds(clouds)
    gt = np.concatenate([syn.syn_kitti_boxes(s, 12) for s in range(frames)]).astype(np.float32)
    goffs = np.arange(frames + 1, dtype=np.int32) * 12
    return [torch.from_numpy(a).cuda() for a in (pts, offs, gt, goffs)]


def test_static_capacity_training_forward_backward_matches_the_dynamic_one():
    """The static-capacity form of the training forward / backward (row counts on the device, capacity rows of garbage behind the
    live ones: what DeviceTrainer.capture_step captures) against the dynamic one from the same state on the same frames: loss
    scalars and every gradient agree up to the fp32-atomic summation order of the sparse weight gradients."""
    import spconv
    from second_amd.models import SecondDetector, CAR_FHD
    from second_amd.training import DeviceTrainer
    ins = _train_inputs()
    torch.manual_seed(0)
    init = SecondDetector(CAR_FHD).state_dict()
    out = {}
    for static in (False, True):
        det = SecondDetector(CAR_FHD)
        det.load_state_dict(init)
        tr = DeviceTrainer(det.cuda(), amp_dtype=torch.bfloat16)
        if static:
            with torch.no_grad():
                tr.forward_loss(*ins)                       # dynamic pass: rows per strided layer
            for m in det.middle_feature_extractor.modules():
                if isinstance(m, spconv.SparseConvolution) and not m.subm:
                    m.static_out_rows = int(-(-int(m.last_num_out * 1.25) // 256) * 256)
            for mod in det.modules():                       # the extra pass must not leave traces in the BatchNorm statistics
                if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                    mod.reset_running_stats()
            det.load_state_dict(init)
            tr.static = True
        loss, out6, _ = tr.forward_loss(*ins)
        loss.backward()
        torch.cuda.synchronize()
        if static:
            tr.check_overflow()
        out[static] = (out6.float().cpu(), {n: p.grad.detach().float().cpu() for n, p in det.named_parameters()},
                       {n: b.detach().float().cpu() for n, b in det.named_buffers() if "running" in n})
    torch.testing.assert_close(out[True][0], out[False][0], rtol=1e-3, atol=1e-5)
    for n, g in out[True][1].items():
        w = out[False][1][n]
        assert (g - w).abs().max().item() <= 2e-2 * w.abs().max().item() + 1e-7, n
    for n, b in out[True][2].items():
        torch.testing.assert_close(b, out[False][2][n], rtol=1e-4, atol=1e-6, msg=n)


def test_whole_training_step_captured_in_one_graph_replays_and_learns():
    """DeviceTrainer.capture_step: the whole optimisation step as ONE hipGraph.  Replays run without host synchronisation, the
    loss on the fixed batch goes down, capacities are not exceeded, new inputs are taken through the graph's buffers."""
    from second_amd.models import SecondDetector, CAR_FHD
    from second_amd.training import DeviceTrainer
    ins = _train_inputs()
    torch.manual_seed(0)
    tr = DeviceTrainer(SecondDetector(CAR_FHD).cuda(), amp_dtype=torch.bfloat16)
    replay = tr.capture_step(*ins)
    first = replay().clone()
    for _ in range(14):
        last = replay(*ins)                                 # same tensors: copied into the graph's buffers first
    torch.cuda.synchronize()
    tr.check_overflow()
    first, last = first.float().cpu(), last.float().cpu()
    assert torch.isfinite(first).all() and torch.isfinite(last).all()
    assert last[0] < first[0], (first.tolist(), last.tolist())
    for p in tr.det.parameters():
        assert torch.isfinite(p).all()


def test_captured_step_has_no_memory_of_the_previous_batch():
    """Scratch that a step zeroes and then accumulates into with atomics (the target assigner's best-overlap word per ground truth)
    must be zeroed by a KERNEL inside a captured graph: ROCm 7.2's memset node does not reliably run before the atomics on replay
    (round 6: found on the drop-in sessions' checksum).  With the learning rate at zero the weights stand still, so the loss of a
    batch must not depend on which batch was replayed before it -- ground truth A, then B (other boxes), then A again."""
    from second_amd import synthetic as syn
    from second_amd.models import SecondDetector, CAR_FHD
    from second_amd.training import DeviceTrainer
    pts, offs, gt_a, goffs = _train_inputs()
    gt_b = torch.from_numpy(np.concatenate([syn.syn_kitti_boxes(s, 24)[12:] for s in (5, 6)]).astype(np.float32)).cuda()   # 12 other boxes per frame
    torch.manual_seed(0)
    tr = DeviceTrainer(SecondDetector(CAR_FHD).cuda(), amp_dtype=torch.bfloat16, lr=0.0, weight_decay=0.0)
    replay = tr.capture_step(pts, offs, gt_a, goffs)
    a1 = replay(pts, offs, gt_a, goffs).clone()
    b1 = replay(pts, offs, gt_b, goffs).clone()
    a2 = replay(pts, offs, gt_a, goffs).clone()
    b2 = replay(pts, offs, gt_b, goffs).clone()
    for _ in range(30):
        replay(pts, offs, gt_b, goffs)
    a3 = replay(pts, offs, gt_a, goffs).clone()
    torch.cuda.synchronize()
    assert not torch.allclose(a1, b1, rtol=1e-3), "the two ground-truth sets must give different losses"
    torch.testing.assert_close(a2, a1, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(b2, b1, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(a3, a1, rtol=1e-5, atol=1e-7)
    # and the eager step of a fresh trainer on the same weights agrees
    torch.manual_seed(0)
    tr2 = DeviceTrainer(SecondDetector(CAR_FHD).cuda(), amp_dtype=torch.bfloat16, lr=0.0, weight_decay=0.0)
    _, e_a, _ = tr2.forward_loss(pts, offs, gt_a, goffs)
    torch.testing.assert_close(a1.float(), e_a.float(), rtol=2e-3, atol=1e-5)


def test_captured_step_follows_the_learning_rate_and_leaves_the_state_as_it_found_it():
    """The captured kernels read the optimizer's hyper-parameters from device memory: a one-cycle schedule (the reference changes
    lr every step, car.fhd.config:171-188) is followed by replays, no re-capture; lr = 0 leaves the weights alone.  capture_step
    itself is not training: parameters, Adam moments, step counter and BatchNorm statistics are as before.  replay() validates what
    it is fed."""
    from second_amd.models import SecondDetector, CAR_FHD
    from second_amd.training import DeviceTrainer
    ins = _train_inputs()
    torch.manual_seed(0)
    tr = DeviceTrainer(SecondDetector(CAR_FHD).cuda(), amp_dtype=torch.bfloat16)
    before = {k: v.clone() for k, v in tr.det.state_dict().items()}
    replay = tr.capture_step(*ins)
    torch.cuda.synchronize()
    for k, v in tr.det.state_dict().items():
        assert torch.equal(v, before[k]), k
    assert tr.steps == 0 and float(tr.opt.state[1]) == 0 and float(tr.opt.exp_avg.abs().max()) == 0
    assert tr.opt.param_groups[0]["lr"] == tr.opt.lr and len(tr.opt.param_groups) == 1
    flat0 = tr.opt.flat.clone()
    tr.opt.param_groups[0]["lr"] = 0.0                       # what a torch LR scheduler does
    tr.opt.weight_decay = 0.0
    replay()
    torch.cuda.synchronize()
    assert torch.equal(tr.opt.flat, flat0), "lr = 0 must leave the weights alone: the captured step still carries the old lr"
    tr.opt.set_lr(1e-3)
    replay()
    torch.cuda.synchronize()
    d1 = float((tr.opt.flat - flat0).abs().max())
    assert 0 < d1 <= 1.01e-3, d1                             # the first Adam step moves every weight by at most lr
    flat1 = tr.opt.flat.clone()
    tr.opt.lr = 4e-3
    replay()
    torch.cuda.synchronize()
    assert float((tr.opt.flat - flat1).abs().max()) > 1.5 * d1
    with pytest.raises(ValueError, match="gt_classes"):
        replay(None, None, None, None, torch.ones(3, dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError, match="points"):
        replay(torch.zeros(ins[0].shape[0] + 5, 4, device="cuda"))
    with pytest.raises(ValueError, match="point_offsets"):
        replay(None, ins[1][:-1])


def test_two_graph_form_of_the_captured_step_equals_the_one_graph_form(monkeypatch):
    """The fallback for a RCCL that refuses capture (forward / backward graph, all-reduce on the stream, clip + AdamW graph;
    SEC_TRAIN_ALLREDUCE_IN_GRAPH=0) ends at the same weights as the single graph (one rank here: the all-reduce is the identity;
    the two-rank versions are tests/test_gpu_multi.py)."""
    from second_amd.models import SecondDetector, CAR_FHD
    from second_amd.training import DeviceTrainer
    ins = _train_inputs()
    flats, losses = [], []
    for mode in ("1", "1", "0"):
        monkeypatch.setenv("SEC_TRAIN_ALLREDUCE_IN_GRAPH", mode)
        torch.manual_seed(0)
        tr = DeviceTrainer(SecondDetector(CAR_FHD).cuda(), amp_dtype=torch.bfloat16)
        replay = tr.capture_step(*ins)
        assert tr.allreduce_in_graph == (mode == "1")
        for _ in range(3):
            out = replay()
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all()
        flats.append(tr.opt.flat.clone())
        losses.append(out.float().cpu())
    # Two runs of the SAME form are the yardstick: the sparse weight gradient combines its chunks with fp32 atomics and the features
    # are bf16, so after three Adam steps two runs agree statistically, not bit for bit.  The two-graph form must sit within that.
    def spread(a, b):
        d = (a - b).abs()
        return float((d > 1e-4).float().mean()), float(d.median())
    same_frac, same_med = spread(flats[0], flats[1])
    other_frac, other_med = spread(flats[0], flats[2])
    assert other_frac <= 2.0 * same_frac + 0.02 and other_med <= 2.0 * same_med + 1e-6, ((same_frac, same_med), (other_frac, other_med))
    assert float((flats[2] - flats[0]).abs().max()) <= 3 * 3e-3 * 2.2        # nobody moved further than three Adam steps can
    torch.testing.assert_close(losses[2][0], losses[0][0], rtol=0.05, atol=1e-3)


def test_flat_adamw_matches_torch_adamw_with_clipping():
    """sec_flat_adamw_f32 (clip_grad_norm_ + AdamW on one flat buffer, two launches) against torch.nn.utils.clip_grad_norm_ +
    torch.optim.AdamW on the same tensors for four steps: gradients large enough to be clipped in some steps and not in others."""
    from second_amd.training import FlatAdamW
    g = torch.Generator().manual_seed(3)
    shapes = [(3, 3, 3, 4, 16), (16,), (64, 32, 3, 3), (7,), (128, 128)]
    ref = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    n = sum(p.numel() for p in ref)
    flat_grad = torch.zeros(n, device="cuda")
    opt_ref = torch.optim.AdamW(ref, lr=3e-3, weight_decay=0.01, betas=(0.9, 0.99))
    opt = FlatAdamW(mine, flat_grad, 3e-3, 0.01, betas=(0.9, 0.99), max_grad_norm=10.0)
    assert all(p.data_ptr() >= opt.flat.data_ptr() and p.data_ptr() < opt.flat.data_ptr() + 4 * n for p in mine)
    for step, scale in enumerate((0.001, 0.5, 0.01, 2.0)):
        grads = [torch.randn(*s, generator=g).cuda() * scale for s in shapes]
        for p, gr in zip(ref, grads):
            p.grad = gr.clone()
        norm_ref = torch.nn.utils.clip_grad_norm_(ref, 10.0)
        opt_ref.step()
        flat_grad.copy_(torch.cat([gr.reshape(-1) for gr in grads]))
        opt.step()
        torch.cuda.synchronize()
        assert abs(float(opt.state[0]) - float(norm_ref)) <= 1e-5 * float(norm_ref) and float(opt.state[1]) == step + 1
        for a, b in zip(mine, ref):
            torch.testing.assert_close(a.detach(), b.detach(), rtol=2e-6, atol=2e-7)


def test_flat_adamw_device_side_loss_scaling_state_machine():
    """Dynamic loss scaling without a host read (sec_flat_adamw_f32's loss_scale4): scaled gradients are unscaled inside the update
    (same parameters as the unscaled torch step), an Inf in the bucket skips the update and halves the scale, `growth interval`
    clean steps double it; the step counter only counts applied steps."""
    from second_amd.training import FlatAdamW
    g = torch.Generator().manual_seed(5)
    ref = [torch.nn.Parameter(torch.randn(300, generator=g).cuda())]
    mine = [torch.nn.Parameter(ref[0].detach().clone())]
    flat_grad = torch.zeros(300, device="cuda")
    opt_ref = torch.optim.AdamW(ref, lr=3e-3, weight_decay=0.01, betas=(0.9, 0.99))
    opt = FlatAdamW(mine, flat_grad, 3e-3, 0.01, betas=(0.9, 0.99), max_grad_norm=10.0)
    ls = opt.enable_loss_scaling(1024.0, growth_interval=3)
    grad = torch.randn(300, generator=g).cuda() * 0.1
    # 1. a clean step: gradients arrive multiplied by the scale
    ref[0].grad = grad.clone()
    torch.nn.utils.clip_grad_norm_(ref, 10.0)
    opt_ref.step()
    flat_grad.copy_(grad * 1024.0)
    opt.step()
    torch.testing.assert_close(mine[0].detach(), ref[0].detach(), rtol=2e-6, atol=2e-7)
    assert ls.tolist() == [1024.0, 1.0, 3.0, 0.0] and opt.state[1].item() == 1.0
    # 2. overflow: nothing moves, the scale halves, the step is not counted
    before = mine[0].detach().clone()
    flat_grad[7] = float("inf")
    opt.step()
    assert torch.equal(mine[0].detach(), before) and ls.tolist() == [512.0, 0.0, 3.0, 1.0]
    assert opt.state[1].item() == 1.0 and opt.state[2].item() == 1.0
    # 3. three clean steps double the scale
    for k in range(3):
        ref[0].grad = grad.clone()
        torch.nn.utils.clip_grad_norm_(ref, 10.0)
        opt_ref.step()
        flat_grad.copy_(grad * float(ls[0].item()))
        opt.step()
    torch.testing.assert_close(mine[0].detach(), ref[0].detach(), rtol=5e-6, atol=5e-7)
    assert ls.tolist() == [1024.0, 0.0, 3.0, 1.0] and opt.state[1].item() == 4.0


def test_pack_weight_train_equals_the_three_separate_packs():
    """sec_pack_conv_weight_train: the 16-bit rounding, the forward MFMA image and the (mirrored / plain) data-gradient image of a
    layer in one launch == to(dtype) + sec_pack_conv_weight + the transposed pack sec_indice_conv_bwd builds itself; and a backward
    that is handed the image returns the same bits as one that packs its own; the optional zeroed weight-gradient accumulator."""
    from second_amd import ops
    torch.manual_seed(5)
    for (k, cin, cout), dt in [((27, 4, 16), torch.bfloat16), ((27, 16, 32), torch.float16), ((27, 64, 64), torch.bfloat16), ((3, 64, 64), torch.bfloat16)]:
        shape = (3, 3, 3) if k == 27 else (3, 1, 1)
        w = torch.randn(*shape, cin, cout, device="cuda")
        for subm in (True, False):
            w16, pk, pkt = ops.pack_weight_train(w, dt, subm)
            assert torch.equal(w16, w.to(dt))
            ref = ops.pack_weight(w.to(dt).contiguous())
            assert (pk is None) == (ref is None) and (pk is None or torch.equal(pk, ref))
            assert pkt is not None
            rng = np.random.default_rng(cin + cout)
            n = 300
            idx = np.unique(np.stack([np.zeros(n, np.int64), rng.integers(0, 6, n), rng.integers(0, 12, n), rng.integers(0, 12, n)], 1), axis=0).astype(np.int32)
            if subm:
                rb = ops.rulebook_subm(torch.from_numpy(idx).cuda(), 1, [6, 12, 12], shape, [1, 1, 1])
                nbr_out, nbr_in, n_out = rb["nbr_out"], None, len(idx)
            else:
                rb = ops.rulebook_conv(torch.from_numpy(idx).cuda(), 1, [6, 12, 12], shape, [2, 2, 2] if k == 27 else [2, 1, 1], [1, 1, 1] if k == 27 else [0, 0, 0], [1, 1, 1])
                nbr_out, nbr_in, n_out = rb["nbr_out"], rb["nbr_in"], int(rb["num_out"])
            feat = torch.randn(len(idx), cin, device="cuda").to(dt)
            dout = torch.randn(n_out, cout, device="cuda").to(dt)
            a = ops.indice_conv_backward(feat, w16, nbr_out[:n_out], nbr_in, dout, packed_dgrad=pkt)
            b = ops.indice_conv_backward(feat, w16, nbr_out[:n_out], nbr_in, dout)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
            # the zeroed weight-gradient accumulator of the same launch (no memset node in the backward): same gradient to the
            # order of the float atomics, and the returned tensor IS the accumulator
            dw0 = ops.pack_weight_train(w, dt, subm, zero_grad=True)[3]
            assert dw0.dtype == torch.float32 and dw0.shape == w.shape and float(dw0.abs().max()) == 0.0
            c = ops.indice_conv_backward(feat, w16, nbr_out[:n_out], nbr_in, dout, packed_dgrad=pkt, dweight_dtype=torch.float32, dweight_out=dw0)
            assert c[1].data_ptr() == dw0.data_ptr() and torch.equal(c[0], a[0])
            ref32 = ops.indice_conv_backward(feat, w16, nbr_out[:n_out], nbr_in, dout, dweight_dtype=torch.float32)[1]
            assert float((c[1] - ref32).abs().max()) <= 1e-4 * float(ref32.abs().max()) + 1e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv_transpose_1x1_function_vs_torch(dtype):
    """ops.ConvTranspose1x1Function (the RPN deblock, rpn.py:275-285: ConvTranspose2d(128, 128, 1, stride 1, bias=False)) forward,
    data gradient and weight gradient vs torch autograd in fp32 on the same 16-bit-rounded operands."""
    from second_amd import ops
    torch.manual_seed(3)
    b, h, w = 2, 37, 50                                      # ragged against every tile size
    x = torch.randn(b, 128, h, w, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    wt = (torch.randn(128, 128, 1, 1, device="cuda") / 11).requires_grad_()
    go = torch.randn(b, 128, h, w, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    y = ops.ConvTranspose1x1Function.apply(x, wt)
    y.backward(go)
    xr = x.detach().float().requires_grad_()
    wr = wt.detach().to(dtype).float().requires_grad_()
    yr = F.conv_transpose2d(xr, wr)
    yr.backward(go.float())
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -9
    for got, ref, what in ((y, yr, "forward"), (x.grad, xr.grad, "data gradient")):
        np.testing.assert_allclose(got.detach().float().cpu().numpy(), ref.detach().cpu().numpy(), rtol=tol, atol=tol * float(ref.abs().max()), err_msg=what)
    assert wt.grad.dtype == torch.float32 and wt.grad.shape == wt.shape
    np.testing.assert_allclose(wt.grad.cpu().numpy(), wr.grad.cpu().numpy(), rtol=2e-3, atol=2e-3 * float(wr.grad.abs().max()))
    # run-to-run identical (fixed-order reduction)
    x.grad = None; wt.grad = None
    ops.ConvTranspose1x1Function.apply(x, wt).backward(go)
    g1 = wt.grad.clone(); wt.grad = None
    ops.ConvTranspose1x1Function.apply(x, wt).backward(go)
    assert torch.equal(g1, wt.grad)


def test_heads_1x1_function_vs_torch():
    """ops.Heads1x1Function (the three 1x1 heads of the RPN stacked into one 128 -> 64 conv with bias, rpn.py:386-391): output, data
    gradient, weight and bias gradients vs torch autograd in fp32 on the same bf16-rounded operands."""
    from second_amd import ops
    torch.manual_seed(4)
    b, h, w = 2, 33, 48
    x = torch.randn(b, 128, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    wt = (torch.randn(64, 128, 1, 1, device="cuda") / 11).requires_grad_()
    bs = torch.randn(64, device="cuda").requires_grad_()
    go = torch.randn(b, 64, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = ops.Heads1x1Function.apply(x, wt, bs)
    y.backward(go)
    xr = x.detach().float().requires_grad_()
    wr = wt.detach().to(torch.bfloat16).float().requires_grad_()
    br = bs.detach().clone().requires_grad_()
    yr = F.conv2d(xr, wr, br)
    yr.backward(go.float())
    tol = 2 ** -7
    for got, ref, what in ((y, yr, "forward"), (x.grad, xr.grad, "data gradient")):
        np.testing.assert_allclose(got.detach().float().cpu().numpy(), ref.detach().cpu().numpy(), rtol=tol, atol=tol * float(ref.abs().max()), err_msg=what)
    np.testing.assert_allclose(wt.grad.cpu().numpy(), wr.grad.cpu().numpy(), rtol=2e-3, atol=2e-3 * float(wr.grad.abs().max()))
    np.testing.assert_allclose(bs.grad.cpu().numpy(), br.grad.cpu().numpy(), rtol=2e-3, atol=2e-3 * float(br.grad.abs().max()))


@pytest.mark.parametrize("bins", [2, 0])
@pytest.mark.parametrize("dtype,scale", [(torch.bfloat16, 1.0), (torch.float16, 512.0)])
def test_heads_loss_function_equals_the_three_tensor_formulation(dtype, scale, bins):
    """ops.HeadsLossFunction (the loss read from the stacked head tensor, its gradient written back into it: sec_heads_loss_fwd / _bwd)
    against the formulation it replaces -- Heads1x1Function, the reference's [B, A, H, W, code] views of the three heads
    (rpn.py:386-391), SecondLossFunction (voxelnet.py:239-312; pinned by tests/golden through sec_second_loss_f32): the six loss
    scalars to summation order, the data / weight / bias gradients of the head convolution exactly as autograd stitches them
    (the gradient is rounded once after the multiplication by the incoming scalar: `scale` plays the fp16 loss scale).  bins = 0: a
    network without the direction classifier (use_direction_classifier false: two heads)."""
    from second_amd import ops
    torch.manual_seed(11)
    b, h, w, a = 2, 24, 20, 2
    n = a * h * w
    x = torch.randn(b, 128, h, w, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    tot = a * (7 + 1 + bins)
    wt = torch.randn(64, 128, 1, 1, device="cuda") / 11
    wt[tot:] = 0
    bs = torch.randn(64, device="cuda") / 4
    bs[tot:] = 0
    g = torch.Generator().manual_seed(5)
    labels = (torch.rand(b, n, generator=g) < 0.03).int()                      # a few positives,
    labels[torch.rand(b, n, generator=g) < 0.05] = -1                            # some don't-cares
    labels = labels.cuda().contiguous()
    reg = (torch.randn(b, n, 7, generator=g) * 0.3).cuda() * (labels > 0).unsqueeze(-1)
    anchors = torch.randn(n, 7, generator=g).cuda()
    imp = (0.5 + torch.rand(b, n, generator=g)).cuda()
    cfg = dict(num_class=1, num_direction_bins=max(bins, 1), direction_offset=0.0)
    sc = torch.tensor(scale, device="cuda")

    def run(fused):
        xi = x.clone().requires_grad_()
        wi, bi = wt.clone().requires_grad_(), bs.clone().requires_grad_()
        if fused:
            loss, out6 = ops.HeadsLossFunction.apply(xi, wi, bi, labels, reg, anchors, imp, a, 1, bins, cfg)
        else:
            y = ops.Heads1x1Function.apply(xi, wi, bi)
            box = y[:, :14].reshape(-1, a, 7, h, w).permute(0, 1, 3, 4, 2).contiguous()
            cls = y[:, 14:16].reshape(-1, a, 1, h, w).permute(0, 1, 3, 4, 2).contiguous()
            dr = y[:, 16:20].reshape(-1, a, 2, h, w).permute(0, 1, 3, 4, 2).contiguous() if bins else None
            loss, out6 = ops.SecondLossFunction.apply(cls, box, dr, labels, reg, anchors, imp, cfg)
        (loss * sc).backward()
        return out6.detach(), xi.grad, wi.grad, bi.grad

    o_f, dx_f, dw_f, db_f = run(True)
    o_r, dx_r, dw_r, db_r = run(False)
    np.testing.assert_allclose(o_f.cpu().numpy(), o_r.cpu().numpy(), rtol=2e-5, atol=1e-7)
    assert torch.equal(dx_f, dx_r)                                               # same dY bit for bit -> same kernel, same result
    assert torch.equal(dw_f[:tot], dw_r[:tot]) and float(dw_f[tot:].abs().max()) == 0.0
    np.testing.assert_allclose(db_f[:tot].cpu().numpy(), db_r[:tot].cpu().numpy(), rtol=1e-4, atol=1e-6 * scale)
    assert float(db_f[tot:].abs().max()) == 0.0 and float(dx_f.float().abs().max()) > 0


def test_dense_channels_last_scatter_has_the_gradient_of_the_permuted_dense():
    """SparseConvTensor.dense_channels_last_2d() under autograd (the RPN input of the training step): values and gradient equal
    dense().view(B, C * D, H, W) (middle.py:206-210), without the permute copies."""
    import spconv
    torch.manual_seed(2)
    b, c, shape = 2, 64, (2, 24, 20)
    coords = torch.stack([torch.randint(0, b, (300,)), torch.randint(0, 2, (300,)), torch.randint(0, 24, (300,)),
                          torch.randint(0, 20, (300,))], 1).unique(dim=0).int().cuda()
    feats = torch.randn(coords.shape[0], c, device="cuda").to(torch.bfloat16)
    go = torch.randn(b, c * 2, 24, 20, device="cuda").to(torch.bfloat16)
    outs = []
    for cl in (True, False):
        f = feats.clone().requires_grad_()
        t = spconv.SparseConvTensor(f, coords, shape, b)
        d = t.dense_channels_last_2d() if cl else t.dense().view(b, c * 2, 24, 20)
        if cl:
            assert d.is_contiguous(memory_format=torch.channels_last)
        d.backward(go.contiguous(memory_format=torch.channels_last) if cl else go)
        outs.append((d.detach(), f.grad))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_prepack_training_weights_equals_the_per_layer_packs():
    """ops.prepack_training_weights (every layer's weight images in two launches: sec_pack_conv_weight_train_multi /
    sec_conv2d_pack_weight_train_multi) leaves in its table exactly what the per-layer launches produce, hands each image out once,
    and is cleared by the next call."""
    from second_amd import ops
    torch.manual_seed(9)
    dt = torch.bfloat16
    sp = [(torch.randn(3, 3, 3, 4, 16, device="cuda"), True, True), (torch.randn(3, 3, 3, 16, 32, device="cuda"), False, True),
          (torch.randn(3, 3, 3, 64, 64, device="cuda"), True, False), (torch.randn(3, 1, 1, 64, 64, device="cuda"), False, True)]
    de = [torch.randn(128, 128, 3, 3, device="cuda"), torch.randn(128, 128, 1, 1, device="cuda"), torch.randn(64, 128, 1, 1, device="cuda")]
    ref_sp = [ops.pack_weight_train(w, dt, subm) for w, subm, _ in sp]
    ref_de = [ops.conv2d_pack_weight_train(w, dt) for w in de]
    ops.prepack_training_weights(sp, de, dt)
    assert len(ops._PREPACK) == len(sp) + len(de)
    for (w, subm, want), ref in zip(sp, ref_sp):
        got = ops.pack_weight_train(w, dt, subm, zero_grad=True)
        for a, b in zip(got[:3], ref):
            assert (a is None) == (b is None) and (a is None or torch.equal(a, b))
        assert got[3].dtype == torch.float32 and got[3].shape == w.shape and float(got[3].abs().max()) == 0.0
    for w, ref in zip(de, ref_de):
        got = ops.conv2d_pack_weight_train(w, dt)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    assert len(ops._PREPACK) == 0                             # every image handed out once
    ops.prepack_training_weights(sp[:1], [], dt)
    ops.prepack_training_weights([], de[:1], dt)              # the table of the previous call is dropped
    assert list(ops._PREPACK) == [("2d", de[0].data_ptr(), dt)]
    ops._PREPACK.clear()
