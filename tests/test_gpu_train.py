"""-m gpu parity of the training-side kernels (csrc/train.hip) against fixtures produced by EXECUTING the reference's own
target assignment (second/core/target_ops.py:29 create_target_np) and loss (second/pytorch/models/voxelnet.py:239-312
VoxelNet.loss over second/pytorch/core/losses.py) -- tests/golden/make_golden.py::gen_train_targets_losses."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


@pytest.fixture(scope="module")
def ops():
    from second_amd import ops
    return ops


def _gt(g):
    frames = [g[f"gt_{i}"] for i in range(3)]
    offs = np.cumsum([0] + [len(f) for f in frames]).astype(np.int32)
    return np.concatenate(frames).astype(np.float32), offs


def test_assign_targets_matches_reference_create_target_np(ops, golden):
    g = golden("train_targets_losses")
    gt, offs = _gt(g)
    labels, targets, importance = ops.assign_targets(dev(g["anchors"]), dev(gt), dev(offs), float(g["matched_threshold"]),
                                                     float(g["unmatched_threshold"]))
    np.testing.assert_array_equal(labels.cpu().numpy(), g["labels"])            # positives, background AND don't-care anchors
    assert (g["labels"] == -1).any() and (g["labels"] > 0).sum() >= 20
    np.testing.assert_allclose(targets.cpu().numpy(), g["bbox_targets"], rtol=1e-5, atol=2e-6)
    np.testing.assert_array_equal(importance.cpu().numpy(), g["importance"])
    # explicit classes / importance: labels take the class of the matched box, positives its importance
    cls = np.arange(len(gt), dtype=np.int32) % 3 + 1
    imp = np.linspace(0.5, 2.0, len(gt)).astype(np.float32)
    l2, t2, i2 = ops.assign_targets(dev(g["anchors"]), dev(gt), dev(offs), 0.6, 0.45, gt_classes=dev(cls), gt_importance=dev(imp))
    l2, i2 = l2.cpu().numpy(), i2.cpu().numpy()
    assert np.array_equal(l2 > 0, g["labels"] > 0) and np.array_equal(l2 == 0, g["labels"] == 0)
    torch.testing.assert_close(t2, targets)
    assert set(np.unique(l2[l2 > 0])) <= {1, 2, 3} and np.all(i2[l2 <= 0] == 1.0)
    # a batch without any ground truth: everything background
    l3, t3, _ = ops.assign_targets(dev(g["anchors"]), dev(np.zeros((0, 7), np.float32)), dev(np.zeros(3, np.int32)), 0.6, 0.45)
    assert (l3 == 0).all() and (t3 == 0).all()


def test_second_loss_values_and_gradients_match_reference(ops, golden):
    g = golden("train_targets_losses")
    out6, d_cls, d_box, d_dir = ops.second_loss_raw(dev(g["cls_preds"]), dev(g["box_preds"]), dev(g["dir_preds"]), dev(g["labels"]),
                                                    dev(g["bbox_targets"]), dev(g["anchors"]), dev(g["importance"]))
    got = out6.cpu().numpy()
    want = [g[k] for k in ("loss", "cls_loss_reduced", "loc_loss_reduced", "dir_loss_reduced", "cls_pos_loss", "cls_neg_loss")]
    np.testing.assert_allclose(got, np.array(want, np.float32), rtol=1e-4)
    for name, got_g, ref_g in (("cls", d_cls, g["d_cls"]), ("box", d_box, g["d_box"]), ("dir", d_dir, g["d_dir"])):
        np.testing.assert_allclose(got_g.cpu().numpy(), ref_g, rtol=1e-4, atol=1e-6 * np.abs(ref_g).max() + 1e-9, err_msg=name)
    # deterministic: the two-stage reduction gives the same bits on every call
    again = ops.second_loss_raw(dev(g["cls_preds"]), dev(g["box_preds"]), dev(g["dir_preds"]), dev(g["labels"]),
                                dev(g["bbox_targets"]), dev(g["anchors"]), dev(g["importance"]))[0]
    assert torch.equal(out6, again)


def test_second_loss_autograd_function(ops, golden):
    """The autograd wrapper: head outputs in the RPN's [B, A, H, W, code] layout and in bf16, gradient scaling by grad_output."""
    g = golden("train_targets_losses")
    b, n = g["labels"].shape
    fm = [int(v) for v in g["feature_map_size"]]
    a = n // (fm[1] * fm[2])
    cls = dev(g["cls_preds"]).reshape(b, a, fm[1], fm[2], 1).clone().requires_grad_()
    box = dev(g["box_preds"]).reshape(b, a, fm[1], fm[2], 7).clone().requires_grad_()
    dirp = dev(g["dir_preds"]).reshape(b, a, fm[1], fm[2], 2).clone().requires_grad_()
    loss, out6 = ops.SecondLossFunction.apply(cls, box, dirp, dev(g["labels"]), dev(g["bbox_targets"]), dev(g["anchors"]),
                                              dev(g["importance"]), {"num_class": 1, "num_direction_bins": 2})
    (3.0 * loss).backward()
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-4)
    np.testing.assert_allclose(cls.grad.reshape(b, n, 1).cpu().numpy(), 3.0 * g["d_cls"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(box.grad.reshape(b, n, 7).cpu().numpy(), 3.0 * g["d_box"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(dirp.grad.reshape(b, n, 2).cpu().numpy(), 3.0 * g["d_dir"], rtol=1e-4, atol=1e-8)


# ---------------------------------------------------------------------------------------------- multi-class (configs 4 / 5)
def _gt_mc(g):
    frames = [g[f"gt_{i}"] for i in range(3)]
    offs = np.cumsum([0] + [len(f) for f in frames]).astype(np.int32)
    cls = np.concatenate([g[f"gt_classes_{i}"] for i in range(3)]).astype(np.int32)
    imp = np.concatenate([g[f"gt_importance_{i}"] for i in range(3)]).astype(np.float32)
    return np.concatenate(frames).astype(np.float32), offs, cls, imp


@pytest.mark.parametrize("mode", ["per_class", "all"])
def test_assign_targets_multiclass_matches_reference_target_assigner(ops, golden, mode):
    """TargetAssigner.assign over three anchor generators, executed by the reference (make_golden.py::gen_train_targets_multiclass):
    assign_per_class (class-filtered ground truth, class thresholds, the reference's importance indexing) and assign_all with
    per-anchor thresholds.  Frame 1 holds boxes of one class only, frame 2 none."""
    g = golden("train_targets_multiclass")
    gt, offs, cls, imp = _gt_mc(g)
    ids = [1, 2, 3] if mode == "per_class" else [0, 0, 0]
    labels, targets, importance = ops.assign_targets_per_class(dev(g["anchors"]), dev(gt), dev(offs), dev(cls),
                                                               g["class_anchor_begin"].tolist(), ids, g["matched"].tolist(),
                                                               g["unmatched"].tolist(), gt_importance=dev(imp))
    want = g[f"labels_{mode}"]
    np.testing.assert_array_equal(labels.cpu().numpy(), want)
    assert sorted(set(np.unique(want).tolist())) == [-1, 0, 1, 2, 3]
    np.testing.assert_allclose(targets.cpu().numpy(), g[f"bbox_targets_{mode}"], rtol=1e-5, atol=2e-6)
    np.testing.assert_array_equal(importance.cpu().numpy(), g[f"importance_{mode}"])
    assert (g[f"importance_{mode}"] != 1.0).any()


def test_second_loss_three_classes_matches_reference(ops, golden):
    g = golden("train_targets_multiclass")
    args = (dev(g["cls_preds"]), dev(g["box_preds"]), dev(g["dir_preds"]), dev(g["labels_per_class"]), dev(g["bbox_targets_per_class"]),
            dev(g["anchors"]), dev(g["importance_per_class"]))
    out6, d_cls, d_box, d_dir = ops.second_loss_raw(*args, direction_offset=0.78)
    want = [g[k] for k in ("loss", "cls_loss_reduced", "loc_loss_reduced", "dir_loss_reduced", "cls_pos_loss", "cls_neg_loss")]
    np.testing.assert_allclose(out6.cpu().numpy(), np.array(want, np.float32), rtol=1e-4)
    for name, got_g, ref_g in (("cls", d_cls, g["d_cls"]), ("box", d_box, g["d_box"]), ("dir", d_dir, g["d_dir"])):
        np.testing.assert_allclose(got_g.cpu().numpy(), ref_g, rtol=1e-4, atol=1e-6 * np.abs(ref_g).max() + 1e-9, err_msg=name)


def test_device_trainer_multiclass_step_runs_and_learns():
    """config 5 (nuscenes all.fhd, ten classes, assign_per_class) on a reduced range: three optimisation steps, finite losses,
    parameters move, per-class ranges line up with the anchor array."""
    from second_amd import models, synthetic
    from second_amd.training import DeviceTrainer
    cfg = dict(models.ALL_FHD_NUSC, point_cloud_range=[-12.8, -12.8, -5, 12.8, 12.8, 3], max_voxels=20000,
               anchor_ranges=[[-12.8, -12.8, r[2], 12.8, 12.8, r[5]] for r in models.ALL_FHD_NUSC["anchor_ranges"]],
               post_center_range=[-15, -15, -10, 15, 15, 10], block_filtering=None)
    torch.manual_seed(0)
    det = models.SecondDetector(cfg).cuda()
    tr = DeviceTrainer(det, lr=1e-3)
    begin = tr.class_ranges[0]
    assert begin[-1] == det.anchors.shape[0] and len(begin) == 11
    rng = np.random.default_rng(0)
    pts, offs, gts, goffs, gcls = [], [0], [], [0], []
    for f in range(2):
        p = rng.uniform([-12.7, -12.7, -4.9, 0], [12.7, 12.7, 2.9, 1], (6000, 4)).astype(np.float32)
        pts.append(p); offs.append(offs[-1] + len(p))
        k = 6
        cls = rng.integers(1, 11, k)
        anc = det.anchors.cpu().numpy()
        b = np.stack([anc[rng.integers(begin[c - 1], begin[c])] for c in cls]).astype(np.float32)
        b[:, :2] += rng.normal(0, 0.05, (k, 2)).astype(np.float32)
        gts.append(b); goffs.append(goffs[-1] + k); gcls.append(cls.astype(np.int32))
    args = (dev(np.concatenate(pts)), dev(np.array(offs, np.int32)), dev(np.concatenate(gts)), dev(np.array(goffs, np.int32)),
            dev(np.concatenate(gcls)))
    before = torch.cat([p.detach().reshape(-1) for p in det.parameters()]).clone()
    losses = []
    for _ in range(3):
        tr.step(*args)
        losses.append(tr.loss_dict())
    assert all(np.isfinite(l["loss"]) for l in losses), losses
    _, _, labels = tr.forward_loss(*args)
    lab = labels.cpu().numpy()
    gc = np.concatenate(gcls)
    for f in range(2):
        for c in set(gc[goffs[f]:goffs[f + 1]].tolist()):          # every ground-truth class got positives inside its own range
            assert (lab[f, begin[c - 1]:begin[c]] == c).any(), (f, c)
        for c in range(1, 11):                                      # and no range holds another class's label
            seg = lab[f, begin[c - 1]:begin[c]]
            assert set(np.unique(seg).tolist()) <= {-1, 0, c}
    after = torch.cat([p.detach().reshape(-1) for p in det.parameters()])
    assert not torch.equal(before, after)


def test_device_trainer_pointpillars_step_runs_and_learns():
    """nuscenes/all.pp.largea (PointPillars) on the device-resident training step: PillarFeatureNet through its differentiable torch
    formulation (second/pytorch/models/pointpillars.py:203-237; the fused sec_pfn_fwd kernel is the inference form), differentiable
    pillar scatter, assign_all target assignment with per-anchor thresholds (all.pp.largea.config:269) -- losses finite, every
    parameter receives a gradient, the loss falls over a few steps on a fixed batch."""
    from second_amd import synthetic as syn
    from second_amd.models import SecondDetector, ALL_PP_LARGEA
    from second_amd.training import DeviceTrainer
    cfg = ALL_PP_LARGEA
    r = cfg["point_cloud_range"]
    clouds = [syn.syn_nusc_cloud(s, num_points=60000, point_cloud_range=tuple(r), scene="urban") for s in range(2)]
    pts, offs = syn.batch_clouds(clouds)
    g = np.random.default_rng(5)
    boxes, classes = [], []
    for s in range(2):
        k = 10
        cls = g.integers(1, len(cfg["anchor_groups"]) + 1, k)
        size = np.array([cfg["anchor_sizes"][cfg["anchor_groups"][c - 1][0]] for c in cls], np.float32)
        z = np.array([cfg["anchor_ranges"][cfg["anchor_groups"][c - 1][0]][2] for c in cls], np.float32)
        xy = g.uniform(-40, 40, (k, 2))
        boxes.append(np.concatenate([xy, z[:, None], size, g.uniform(-np.pi, np.pi, (k, 1))], 1).astype(np.float32))
        classes.append(cls.astype(np.int32))
    gt, goffs = np.concatenate(boxes), np.array([0, 10, 20], np.int32)
    torch.manual_seed(0)
    det = SecondDetector(cfg).cuda()
    tr = DeviceTrainer(det, lr=1e-3)
    d = lambda a: torch.from_numpy(a).cuda()
    args = (d(pts), d(offs), d(gt), d(goffs), d(np.concatenate(classes)))
    loss, out6, labels = tr.forward_loss(*args)
    loss.backward()
    assert int((labels > 0).sum()) >= 5
    missing = [n for n, p in det.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    assert not missing, missing
    for p in det.parameters():
        p.grad = None
    losses = [float(tr.step(*args)[0]) for _ in range(6)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


@pytest.mark.parametrize("pillars,points", [(700, 60), (33, 5), (5000, 20)])
def test_pfn_training_kernels_vs_torch_autograd(pillars, points, monkeypatch):
    """sec_pfn_train_fwd / sec_pfn_train_bwd (PFNLayer under train(): Linear -> BatchNorm1d batch statistics -> ReLU -> max over the
    points, pointpillars.py:51-65, and its backward through the argmax) against torch autograd of the reference formulation in
    fp32: output, gradients of linear.weight / norm.weight / norm.bias, running statistics.  Ragged pillars (1 .. T points, some
    full), so padded slots enter the statistics and sometimes win the max."""
    from second_amd.models import PillarFeatureNet
    torch.manual_seed(pillars)
    g = torch.Generator().manual_seed(pillars + 1)
    t = points
    n = torch.randint(1, t + 1, (pillars,), generator=g)
    n[:: 7] = t
    vox = torch.zeros(pillars, t, 4)
    coors = torch.stack([torch.zeros(pillars, dtype=torch.long), torch.zeros(pillars, dtype=torch.long),
                         torch.randint(0, 400, (pillars,), generator=g), torch.randint(0, 400, (pillars,), generator=g)], 1).int()
    for p in range(pillars):
        k = int(n[p])
        cx, cy = coors[p, 3].item() * 0.25 - 50 + 0.125, coors[p, 2].item() * 0.25 - 50 + 0.125
        vox[p, :k, 0] = cx + (torch.rand(k, generator=g) - 0.5) * 0.25
        vox[p, :k, 1] = cy + (torch.rand(k, generator=g) - 0.5) * 0.25
        vox[p, :k, 2] = torch.rand(k, generator=g) * 4 - 3
        vox[p, :k, 3] = torch.rand(k, generator=g)
    res = {}
    for backend in ("torch", "hip"):
        torch.manual_seed(3)
        net = PillarFeatureNet(4, (64,), (0.25, 0.25, 8), (-50, -50, -5, 50, 50, 3)).cuda().train()
        net.train_backend = backend
        with torch.no_grad():
            net.pfn_layers[0].norm.weight.uniform_(0.5, 1.5)
            net.pfn_layers[0].norm.bias.uniform_(-0.3, 0.3)
        out = net(vox.cuda(), n.int().cuda(), coors.cuda())
        w = torch.randn(out.shape, generator=torch.Generator().manual_seed(9)).cuda()
        (out * w).sum().backward()
        l = net.pfn_layers[0]
        res[backend] = dict(out=out.detach().float().cpu(), dw=l.linear.weight.grad.cpu(), dg=l.norm.weight.grad.cpu(),
                            db=l.norm.bias.grad.cpu(), rm=l.norm.running_mean.cpu(), rv=l.norm.running_var.cpu(),
                            nb=int(l.norm.num_batches_tracked))
    a, b = res["torch"], res["hip"]
    assert a["nb"] == b["nb"] == 1
    for k, tol in (("out", 2e-4), ("dw", 2e-3), ("dg", 2e-3), ("db", 2e-3), ("rm", 1e-5), ("rv", 1e-5)):
        scale = max(a[k].abs().max().item(), 1e-6)
        err = (a[k] - b[k]).abs().max().item() / scale
        assert err <= tol, (k, err)
