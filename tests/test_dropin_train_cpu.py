"""Host side of the training drop-in (second_amd/dropin_train.py), no GPU: the stand-in network's training forward is the
reference's ``VoxelNet.loss`` (pinned to tests/golden/train_targets_losses.npz, which make_golden.py produced by executing the
reference), the loss settings are read off the network object, and the mirror shares -- not copies -- the network's tensors."""
import types

import numpy as np
import pytest
import torch


def test_standin_loss_is_the_reference_loss(golden):
    import reference_standin as rs
    z = golden("train_targets_losses")
    net = types.SimpleNamespace(
        _num_class=1, _pos_cls_weight=1.0, _neg_cls_weight=1.0, _cls_loss_weight=1.0, _loc_loss_weight=2.0, _direction_loss_weight=0.2,
        _sin_error_factor=1.0, _use_direction_classifier=True, _num_direction_bins=2, _dir_offset=0.0,
        _cls_loss_ftor=rs._named("SigmoidFocalClassificationLoss", _alpha=0.25, _gamma=2.0),
        _loc_loss_ftor=rs._named("WeightedSmoothL1LocalizationLoss", _sigma=3.0, _codewise=True, _code_weights=torch.ones(7)))
    cls = torch.from_numpy(z["cls_preds"]).requires_grad_()
    box = torch.from_numpy(z["box_preds"]).requires_grad_()
    dirp = torch.from_numpy(z["dir_preds"]).requires_grad_()
    ex = {"labels": torch.from_numpy(z["labels"]), "reg_targets": torch.from_numpy(z["bbox_targets"]),
          "importance": torch.from_numpy(z["importance"]), "anchors": torch.from_numpy(z["anchors"]).unsqueeze(0).expand(3, -1, -1)}
    r = rs.standin_loss(net, ex, {"box_preds": box, "cls_preds": cls, "dir_cls_preds": dirp})
    r["loss"].backward()
    for k in ("loss", "loc_loss_reduced", "cls_loss_reduced", "dir_loss_reduced", "cls_pos_loss", "cls_neg_loss"):
        np.testing.assert_allclose(float(r[k].detach()), float(z[k]), rtol=1e-6, err_msg=k)
    np.testing.assert_allclose(r["cls_loss"].detach().numpy(), z["cls_loss"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(r["loc_loss"].detach().numpy(), z["loc_loss"], rtol=1e-6, atol=1e-9)
    for g, k in ((cls.grad, "d_cls"), (box.grad, "d_box"), (dirp.grad, "d_dir")):
        np.testing.assert_allclose(g.numpy(), z[k], rtol=1e-5, atol=1e-8, err_msg=k)
    assert torch.equal(r["cared"], ex["labels"] >= 0)


def test_train_config_reads_the_loss_settings_and_refuses_what_the_kernel_lacks():
    from reference_standin import build_voxelnet
    from second_amd import dropin_train as T
    from second_amd.models import CAR_FHD
    net = build_voxelnet(CAR_FHD)
    cfg = T.train_config(net)
    assert cfg["alpha"] == 0.25 and cfg["gamma"] == 2.0 and cfg["sigma"] == 3.0 and cfg["localization_weight"] == 2.0
    assert cfg["direction_loss_weight"] == 0.2 and cfg["num_class"] == 1 and cfg["num_direction_bins"] == 2 and len(cfg["code_weights"]) == 7
    net._loss_norm_type.name = "NormByNumExamples"
    with pytest.raises(T.NotTrainable, match="loss_norm_type"):
        T.train_config(net)
    net._loss_norm_type.name = "NormByNumPositives"
    net._encode_rad_error_by_sin = False
    with pytest.raises(T.NotTrainable, match="encode_rad_error_by_sin"):
        T.train_config(net)


def test_mirror_shares_the_networks_tensors():
    from reference_standin import build_voxelnet
    from second_amd import dropin, dropin_train as T
    from second_amd.models import CAR_FHD, SecondDetector
    net = build_voxelnet(CAR_FHD)
    det = SecondDetector(dropin.model_config(net))
    assert T.share_state(det, net) == []
    theirs = dict(net.named_parameters())
    for n, p in det.named_parameters():
        assert p is theirs[n], n
    nb = dict(net.named_buffers())
    for n, b in det.named_buffers():
        if n.split(".")[0] in ("middle_feature_extractor", "rpn"):
            assert b is nb[n], n
    with torch.no_grad():
        net.rpn.conv_cls.weight.add_(1.0)
    assert torch.equal(det.rpn.conv_cls.weight, net.rpn.conv_cls.weight)
    # engines refuse what they cannot reproduce, with the reason
    with pytest.raises(T.NotTrainable, match="16-bit"):
        T.FusedTrainStep(net, dropin.model_config(net), torch.float32)
    step = T.FusedTrainStep(net, dropin.model_config(net), torch.bfloat16)
    with pytest.raises(T.NotTrainable, match="fp32 parameters on the GPU"):
        step.refresh()                                       # CPU parameters
    assert not step.accepts({"voxels": torch.zeros(1, 5, 4)})
