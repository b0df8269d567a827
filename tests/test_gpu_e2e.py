"""-m gpu END-TO-END parity: SecondDetector.forward_points (HIP path, fp32) vs the CPU restatement of the whole
VoxelNet.forward (oracle/cpu_forward.py: oracle voxeliser / rulebooks / indice_conv / NMS + torch-CPU RPN) on a BASELINE-size
frame (17 000 points -> 16 000 voxels, car.fhd; second/pytorch/models/voxelnet.py:314-375,377-645).

Compared stage by stage: voxel coordinates (bit-exact), every strided layer's output indices (bit-exact), every layer's
features and the dense RPN input (<= 1e-4 relative, BASELINE.json), the raw head outputs, the pre-NMS candidates, the NMS keep
list and the final boxes / scores.

Random-init weights give near-constant class logits (every empty region of the map produces the same value), so the top-k
would be decided by ties.  The test therefore rescales the class head ON BOTH SIDES (same state dict) so that empty regions
fall below the score threshold and only data-driven anchors pass it -- the selection, decode and NMS stages then see distinct
scores, as they do with trained weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _randomise_bn(det, seed=1):
    """A random network that behaves like a trained one: conv gains that keep the activations O(10-100) through the 14 + 6
    layers (the default init shrinks them 3x per sparse layer), BatchNorm statistics with positive means and zero beta (negative
    folded shifts), so that -- as with trained weights -- empty regions of the map carry exactly zero activations."""
    import spconv
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in det.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.running_mean.copy_(torch.empty_like(m.running_mean).uniform_(0.0, 0.1, generator=g))
                m.running_var.copy_(torch.empty_like(m.running_var).uniform_(0.5, 1.5, generator=g))
            if isinstance(m, spconv.SparseConvolution):
                m.weight.mul_(4.0)
        for blk in list(det.rpn.blocks) + list(det.rpn.deblocks):
            for m in blk:
                if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                    m.weight.mul_(2.5)


def _sharpen_class_head(det, cloud):
    """conv_cls <- a * (conv_cls - empty-region logit) - 2 with a = 14 / max: empty map regions score sigmoid(-2) = 0.12 < 0.3,
    the strongest anchor gets logit 12 (no fp32 sigmoid saturation, so no score ties), a few hundred anchors pass the 0.3
    threshold.  Calibrated with one CPU forward; applied to the module both pipelines load."""
    from oracle.cpu_forward import forward_frame
    with torch.no_grad():
        zero = det.rpn(torch.zeros(1, 128, 24, 24))["cls_preds"]          # [1, A, H, W, 1]: the map of an empty scene
        c_empty = zero[0, :, 12, 12, 0].clone()                            # interior value per anchor channel
        tr = forward_frame(det, cloud, collect=True)["trace"]
        d = torch.from_numpy(tr["cls_preds"])[0, :, :, :, 0] - c_empty.view(-1, 1, 1)
        a = 14.0 / d.max().item()
        det.rpn.conv_cls.weight.mul_(a)
        det.rpn.conv_cls.bias.copy_(a * (det.rpn.conv_cls.bias - c_empty) - 2.0)
        # box residuals of trained-network size (|delta| <= 0.5): decoded boxes stay near their anchors and inside the range
        sb = 0.5 / float(np.abs(tr["box_preds"]).max())
        det.rpn.conv_box.weight.mul_(sb)
        det.rpn.conv_box.bias.mul_(sb)


def test_detector_fp32_matches_cpu_forward_stage_by_stage():
    from oracle.cpu_forward import forward_frame
    from second_amd import ops, synthetic as syn
    from second_amd.models import SecondDetector, CAR_FHD
    torch.manual_seed(0)
    det = SecondDetector(CAR_FHD).eval()
    _randomise_bn(det)
    cloud = syn.syn_kitti_cloud(0)
    assert cloud.shape == (17000, 4)
    with torch.no_grad():
        _sharpen_class_head(det, cloud)
    ref = forward_frame(det, cloud, collect=True)
    tr = ref["trace"]
    assert len(tr["voxel_coordinates"]) == 16000 and 100 < len(tr["candidate_scores"]) <= 1000 and ref["num_detections"] >= 5
    assert len(np.unique(tr["candidate_scores"])) > 0.98 * len(tr["candidate_scores"])      # no tie-driven selection

    gpu = SecondDetector(CAR_FHD).eval()
    gpu.load_state_dict(det.state_dict())
    gpu = gpu.cuda()
    pts, offs = syn.batch_clouds([cloud])
    calls = []
    ops.set_op_hook(lambda name, fn, a, kw, res: calls.append((name, a, kw, res)))
    try:
        with torch.no_grad():
            vox = gpu.voxel_generator.generate_device(torch.from_numpy(pts).cuda(), torch.from_numpy(offs).cuda(), mean_features=4)
            spatial = gpu.middle_feature_extractor(vox["mean"], vox["coordinates"], 1)
            preds = gpu.rpn(spatial)
            out = gpu.predict_device(preds, 1)
    finally:
        ops.set_op_hook(None)
    # -- voxelise (bit-exact) + SimpleVoxel
    np.testing.assert_array_equal(vox["coordinates"].cpu().numpy(), tr["voxel_coordinates"])
    np.testing.assert_allclose(vox["mean"].cpu().numpy(), tr["voxel_features"], rtol=1e-6, atol=1e-7)
    # -- sparse middle: every strided layer's output numbering is bit-exact, every layer's features within 1e-4
    convs = [c for c in calls if c[0] == "indice_conv"]
    downs = [c for c in calls if c[0] == "rulebook_conv"]
    assert len(convs) == 14 and len(downs) == 4
    ref_down = [l for l in tr["layers"] if not l["subm"]]
    for c, l in zip(downs, ref_down):
        np.testing.assert_array_equal(c[3]["out_indices"].cpu().numpy(), l["out_indices"])
    for i, (c, l) in enumerate(zip(convs, tr["layers"])):
        got = c[3].float().cpu().numpy()
        assert got.shape == l["features"].shape, i
        np.testing.assert_allclose(got, l["features"], rtol=1e-4, atol=1e-4 * np.abs(l["features"]).max(), err_msg=f"sparse layer {i}")
    # -- dense RPN input
    np.testing.assert_allclose(spatial.float().cpu().numpy(), tr["spatial_features"], rtol=1e-4,
                               atol=1e-4 * np.abs(tr["spatial_features"]).max())
    # -- RPN heads (fp32 MIOpen vs torch-CPU convolutions: different summation orders)
    for k in ("cls_preds", "box_preds", "dir_cls_preds"):
        r = tr[k]
        np.testing.assert_allclose(preds[k].float().cpu().numpy(), r, rtol=2e-3, atol=2e-4 * np.abs(r).max(), err_msg=k)
    # -- predict: the detections (voxelnet.py:616-643)
    m = out["valid"][0].cpu().numpy()
    boxes, scores = out["boxes"][0].cpu().numpy()[m], out["scores"][0].cpu().numpy()[m]
    assert len(boxes) == ref["num_detections"], (len(boxes), ref["num_detections"])
    np.testing.assert_allclose(scores, ref["scores"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(boxes[:, :6], ref["boxes"][:, :6], rtol=1e-3, atol=2e-3)
    dr = np.abs(boxes[:, 6] - ref["boxes"][:, 6])
    assert np.all(np.minimum(dr, np.abs(dr - 2 * np.pi)) < 2e-3)


def test_detector_bf16_static_graph_detections_close_to_cpu_forward():
    """The bench configuration itself (bf16, static capacities, hipGraph) against the fp32 CPU forward on the same frames:
    the same number of detections per frame, every CPU detection present on the device within 0.25 m / 0.05 score."""
    from oracle.cpu_forward import forward_frame
    from second_amd import synthetic as syn
    from second_amd.models import SecondDetector, CAR_FHD
    torch.manual_seed(0)
    det = SecondDetector(CAR_FHD).eval()
    _randomise_bn(det)
    clouds = [syn.syn_kitti_cloud(s) for s in range(2)]
    with torch.no_grad():
        _sharpen_class_head(det, clouds[0])
    refs = [forward_frame(det, c) for c in clouds]
    gpu = SecondDetector(CAR_FHD).eval()
    gpu.load_state_dict(det.state_dict())
    gpu = gpu.cuda().prepare_inference(torch.bfloat16)
    pts, offs = syn.batch_clouds(clouds)
    pts, offs = torch.from_numpy(pts).cuda(), torch.from_numpy(offs).cuda()
    with torch.no_grad():
        gpu.calibrate(pts, offs)
        replay, out = gpu.make_graphed(pts, offs)
        replay()
        torch.cuda.synchronize()
        gpu.check_overflow()
    found = total = 0
    for f, r in enumerate(refs):
        m = out["valid"][f].cpu().numpy()
        gb, gs = out["boxes"][f].float().cpu().numpy()[m], out["scores"][f].float().cpu().numpy()[m]
        assert abs(len(gb) - r["num_detections"]) <= max(2, r["num_detections"] // 10), (f, len(gb), r["num_detections"])
        for bx, sc in zip(r["boxes"], r["scores"]):
            total += 1
            d = np.hypot(gb[:, 0] - bx[0], gb[:, 1] - bx[1])
            found += bool(((d < 0.25) & (np.abs(gs - sc) < 0.05)).any())
    assert total >= 10 and found >= 0.8 * total, (found, total)     # bf16 features flip a few near-threshold NMS decisions
