"""-m gpu: the reference's MODULE GRAPH on the MI355X over libsecond_hip.so -- the drop-in path, not the fused mirror.

/root/reference does not exist on the GPU box, so this file builds the sparse middle and the RPN the way the reference's own
constructors do (second/pytorch/models/middle.py:119-192, rpn.py:248-286,468-497):
  * layer classes are made by a ``change_default_args``-style wrapper that SUBCLASSES spconv.SubMConv3d / SparseConv3d /
    nn.BatchNorm1d and injects defaults after inspecting the base ``__init__`` signature (torchplus/tools.py:32-45 does exactly
    that: a layer whose ``bias`` is hidden in **kwargs breaks here),
  * one ``spconv.SparseSequential`` of 14 x (conv, BatchNorm1d(eps 1e-3, momentum 0.01), ReLU) with ``indice_key`` reuse inside a
    stage, positional kernel / stride arguments, ``padding=[0, 1, 1]`` and the (3,1,1)/(2,1,1) last layer,
  * ``forward``: ``coors.int()`` -> ``SparseConvTensor(features, coors, sparse_shape, batch)`` -> sequential -> ``.dense()`` ->
    ``view(N, C * D, H, W)`` (middle.py:196-210),
on CUDA tensors in eager (dynamic-shape) mode with the drop-in default rulebook numbering (spconv's CPU first-touch order).
Checked against oracle/cpu_forward.py (fp32, <= 1e-4) layer by layer and at the dense RPN input, with the inference peephole
(conv + BN + ReLU fused) ON (what the unmodified reference gets in eval mode) and OFF (three separate modules per layer)."""
import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


from reference_standin import build_middle, with_defaults  # noqa: E402,F401  (the construction recipe, shared with bench.py)


@pytest.fixture(scope="module")
def setup():
    from oracle.cpu_forward import forward_frame
    from second_amd import synthetic as syn
    from second_amd.models import CAR_FHD
    from e2e_trace import trained_like_detector
    clouds = [syn.syn_kitti_cloud(s, num_points=9000, num_voxels=8000) for s in range(2)]
    det = trained_like_detector(CAR_FHD, clouds[0])              # CPU, fp32, eval: the oracle side
    refs = [forward_frame(det, c, collect=True) for c in clouds]
    return det, clouds, refs


def test_layer_classes_survive_default_injection_and_keep_the_state_dict_keys(setup):
    import spconv
    det, _, _ = setup
    mid = build_middle([1, 40, 1600, 1408, 64])
    convs = [m for m in mid.middle_conv if isinstance(m, spconv.SparseConvolution)]
    assert len(convs) == 14 and all(c.bias is None for c in convs)
    assert convs[2].stride == [2, 2, 2] and convs[9].padding == [0, 1, 1] and convs[13].kernel_size == [3, 1, 1]
    bns = [m for m in mid.middle_conv if isinstance(m, nn.BatchNorm1d)]
    assert all(b.eps == 1e-3 and b.momentum == 0.01 for b in bns)
    assert isinstance(spconv.SubMConv3d(4, 8, 3).bias, nn.Parameter)                       # the un-wrapped default is bias=True
    ref_keys = {k for k in det.middle_feature_extractor.state_dict()}
    assert set(mid.state_dict()) == ref_keys                                               # checkpoints interchange


@pytest.mark.parametrize("fused_peephole", [True, False])
def test_reference_module_graph_on_the_device_matches_the_cpu_forward(setup, fused_peephole):
    import spconv
    from second_amd import ops, synthetic as syn
    det, clouds, refs = setup
    mid = build_middle([1, 40, 1600, 1408, 64])
    mid.load_state_dict(det.middle_feature_extractor.state_dict())
    mid = mid.eval().cuda()
    mid.middle_conv.fuse_inference = fused_peephole
    # the reference's example dict for a batch of two (preprocess.py:44-50: batch index prepended to the coordinates)
    gen = det.voxel_generator
    vox = [gen.generate(c, 20000) for c in clouds]
    feats = [v["voxels"][:, :, :4].sum(1) / v["num_points_per_voxel"][:, None].astype(np.float32) for v in vox]      # SimpleVoxel
    coors = [np.concatenate([np.full((len(v["coordinates"]), 1), b, np.int32), v["coordinates"]], 1) for b, v in enumerate(vox)]
    for b in range(2):
        np.testing.assert_array_equal(coors[b][:, 1:], refs[b]["trace"]["voxel_coordinates"][:, 1:])
    f = torch.from_numpy(np.concatenate(feats)).cuda()
    c = torch.from_numpy(np.concatenate(coors)).cuda()
    calls = []
    ops.set_op_hook(lambda name, fn, a, kw, res: calls.append((name, a, kw, res)))
    try:
        with torch.no_grad():
            spatial = mid(f, c, 2)
    finally:
        ops.set_op_hook(None)
    names = [n for n, *_ in calls]
    assert names.count("indice_conv") == 14 and names.count("rulebook_conv") == 4 and names.count("rulebook_subm") == 4   # indice_key reuse
    assert names.count("rulebook_chain") == 0 and names.count("sparse_to_dense") == 1
    assert tuple(spatial.shape) == (2, 128, 200, 176) and spatial.dtype == torch.float32
    # strided layers: output numbering of the drop-in default == the CPU rulebook's, per frame
    downs = [r for n, a, kw, r in calls if n == "rulebook_conv"]
    for li, rb in enumerate(downs):
        oi = rb["out_indices"].cpu().numpy()
        for b in range(2):
            want = [l for l in refs[b]["trace"]["layers"] if not l["subm"]][li]["out_indices"]
            got = oi[oi[:, 0] == b]
            assert np.array_equal(np.sort(_lin(got), kind="stable"), np.sort(_lin(want), kind="stable")), (li, b)
    for b in range(2):
        ref = refs[b]["trace"]["spatial_features"][0]
        got = spatial[b].cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()), err_msg=f"dense RPN input, frame {b}")


def _lin(idx):
    return ((idx[:, 1].astype(np.int64) * 4096) + idx[:, 2]) * 4096 + idx[:, 3]


def test_detector_forward_example_through_the_modules_matches_the_cpu_detections(setup):
    """VoxelNet.forward's inference contract (voxelnet.py:339-375,616-643): example dict in, list of per-frame dicts out, through
    the unfused-capable module path (SecondDetector.forward: SimpleVoxel module, SpMiddleFHD sequential, .dense(), torch RPN,
    predict) in fp32 -- detections equal the CPU forward's."""
    from second_amd.models import SecondDetector, CAR_FHD
    det, clouds, refs = setup
    gpu = SecondDetector(CAR_FHD).eval()
    gpu.load_state_dict(det.state_dict())
    gpu = gpu.cuda()
    gen = det.voxel_generator
    vox = [gen.generate(c, 20000) for c in clouds]
    example = {
        "voxels": torch.from_numpy(np.concatenate([v["voxels"] for v in vox])).cuda(),
        "num_points": torch.from_numpy(np.concatenate([v["num_points_per_voxel"] for v in vox])).cuda(),
        "coordinates": torch.from_numpy(np.concatenate([np.concatenate([np.full((len(v["coordinates"]), 1), b, np.int32), v["coordinates"]], 1)
                                                        for b, v in enumerate(vox)])).cuda(),
        "anchors": gpu.anchors.unsqueeze(0).expand(2, -1, -1).contiguous(),
    }
    out = gpu(example)
    assert isinstance(out, list) and len(out) == 2 and set(out[0]) >= {"box3d_lidar", "scores", "label_preds"}
    for b in range(2):
        ref = refs[b]
        assert out[b]["box3d_lidar"].shape[0] == ref["num_detections"], (b, out[b]["box3d_lidar"].shape, ref["num_detections"])
        np.testing.assert_allclose(out[b]["scores"].cpu().numpy(), ref["scores"], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(out[b]["box3d_lidar"][:, :6].cpu().numpy(), ref["boxes"][:, :6], rtol=1e-3, atol=2e-3)
