"""Drop-in check (build container only: needs /root/reference): the UNMODIFIED reference
(second/pytorch/train.py build_network + VoxelNet.forward) runs over this repo's `spconv` package, and our
SecondDetector mirror produces the same detections from the same state dict.

No GPU here, so the HIP ops are replaced by the CPU oracle through tests/oracle_backend.py (test-only
injection).  What this exercises is the host side of the boundary: package API surface, module classes,
`change_default_args` subclassing, indice_key rulebook caching, SparseSequential, checkpoint key names,
VoxelGeneratorV2 attributes, the numpy-facing NMS helpers and the predict path.  BASELINE config 1
("car.fhd VoxelNet forward on 1 synthetic cloud, CPU-only plumbing")."""
import os

import numpy as np
import pytest
import torch

REF = os.environ.get("SECOND_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "second")), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    from second_amd import compat
    compat.install(REF)
    from google.protobuf import text_format
    from second.protos import pipeline_pb2
    import second.pytorch.train as train
    cfg = pipeline_pb2.TrainEvalPipelineConfig()
    text_format.Merge(open(os.path.join(REF, "second/configs/car.fhd.config")).read(), cfg)
    return train, cfg


def test_all_benchmark_configs_parse(ref):
    from google.protobuf import text_format
    from second.protos import pipeline_pb2
    for rel in ("car.fhd.config", "all.fhd.config", "nuscenes/all.pp.largea.config", "nuscenes/all.fhd.config"):
        cfg = pipeline_pb2.TrainEvalPipelineConfig()
        text_format.Merge(open(os.path.join(REF, "second/configs", rel)).read(), cfg)
        assert cfg.model.second.voxel_generator.voxel_size


def test_reference_voxelnet_forward_over_our_spconv(ref):
    import oracle_backend
    from second_amd import synthetic as syn
    from second_amd.models import SecondDetector, CAR_FHD
    train, cfg = ref
    model_cfg = cfg.model.second
    model_cfg.target_assigner.class_settings[0].nms_pre_max_size = 150   # keep the pure-Python iou_jit fast
    torch.manual_seed(0)
    with oracle_backend.installed():
        net = train.build_network(model_cfg).eval()
        g = torch.Generator().manual_seed(1)
        for m in net.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.running_mean.copy_(torch.empty_like(m.running_mean).uniform_(-0.1, 0.1, generator=g))
                m.running_var.copy_(torch.empty_like(m.running_var).uniform_(0.5, 1.5, generator=g))
        vg = net.voxel_generator
        assert vg.grid_size.tolist() == [1408, 1600, 40]
        np.testing.assert_allclose(vg.point_cloud_range[[0, 1, 3, 4]], [0, -40, 70.4, 40], rtol=1e-6)
        cloud = syn.syn_kitti_cloud(0, num_points=6000, num_voxels=5000)
        vox = vg.generate(cloud, 40000)
        assert set(vox) >= {"voxels", "coordinates", "num_points_per_voxel"} and vox["coordinates"].shape[1] == 3
        fm = [1, 200, 176]
        anchors = net.target_assigner.generate_anchors(fm)["anchors"].reshape(1, -1, 7)
        example = {
            "voxels": vox["voxels"], "num_points": vox["num_points_per_voxel"],
            "coordinates": np.pad(vox["coordinates"], ((0, 0), (1, 0)), mode="constant", constant_values=0),
            "anchors": anchors,
        }
        ex = train.example_convert_to_torch(example, torch.float32, torch.device("cpu"))
        with torch.no_grad():
            ref_out = net(ex)                                           # the whole unmodified forward incl. predict
            ref_preds = net.network_forward(ex["voxels"], ex["num_points"], ex["coordinates"], 1)
        assert len(ref_out) == 1 and ref_out[0]["box3d_lidar"].shape[1] == 7 and ref_out[0]["box3d_lidar"].shape[0] > 0

        # our mirror, same weights (the reference's state dict loads by key), same example
        cfg_m = dict(CAR_FHD, nms_pre_max_size=150)
        det = SecondDetector(cfg_m).eval()
        missing = det.load_state_dict({k: v for k, v in net.state_dict().items() if k in det.state_dict()})
        assert not missing.missing_keys
        np.testing.assert_allclose(det.anchors.numpy(), anchors[0], rtol=0, atol=1e-5)
        with torch.no_grad():
            feats = det.voxel_feature_extractor(ex["voxels"], ex["num_points"])
            ours_preds = det.network_forward(feats, ex["coordinates"], 1)
        for k in ("box_preds", "cls_preds", "dir_cls_preds"):
            np.testing.assert_allclose(ours_preds[k].numpy(), ref_preds[k].numpy(), rtol=1e-4, atol=1e-5)

        # predict: most of the map is empty so real scores tie; top-k / argsort tie-breaking is implementation
        # defined in the reference itself, hence distinct synthetic scores for the comparison
        gen = torch.Generator().manual_seed(3)
        fake = {k: v.clone() for k, v in ref_preds.items()}
        fake["cls_preds"] = torch.randn(fake["cls_preds"].shape, generator=gen) * 0.7 - 1.2
        fake["box_preds"] = torch.randn(fake["box_preds"].shape, generator=gen) * 0.2
        fake["dir_cls_preds"] = torch.randn(fake["dir_cls_preds"].shape, generator=gen)
        with torch.no_grad():
            r = net.predict(ex, {k: v.clone() for k, v in fake.items()})[0]
            o = det.predict(fake, ex["anchors"].view(1, -1, 7))[0]
    assert r["scores"].shape[0] > 5
    np.testing.assert_allclose(o["scores"].numpy(), r["scores"].numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o["box3d_lidar"].numpy(), r["box3d_lidar"].numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(o["label_preds"].numpy(), r["label_preds"].numpy())


def test_api_surface_used_by_reference():
    """Names / signatures the reference touches."""
    import inspect
    import spconv
    from spconv import utils
    for name in ("SparseConvTensor", "SubMConv3d", "SparseConv3d", "SparseSequential", "SparseModule", "ops", "utils"):
        assert hasattr(spconv, name)
    for name in ("VoxelGeneratorV2", "points_to_voxel", "non_max_suppression", "non_max_suppression_cpu",
                 "rotate_non_max_suppression_cpu", "rbbox_iou", "rbbox_intersection"):
        assert hasattr(utils, name)
    # torchplus.tools.change_default_args(bias=False)(cls) looks `bias` up in __init__'s signature
    for cls in (spconv.SubMConv3d, spconv.SparseConv3d):
        params = inspect.signature(cls.__init__).parameters
        assert "bias" in params and params["bias"].kind == inspect.Parameter.POSITIONAL_OR_KEYWORD
        assert list(params)[1:4] == ["in_channels", "out_channels", "kernel_size"] and "indice_key" in params
    conv = spconv.SparseConv3d(64, 64, (3, 1, 1), (2, 1, 1))
    assert tuple(conv.weight.shape) == (3, 1, 1, 64, 64) and conv.bias.shape == (64,)
    sig = inspect.signature(utils.VoxelGeneratorV2.__init__).parameters
    for kw in ("voxel_size", "point_cloud_range", "max_num_points", "max_voxels", "full_mean", "block_filtering",
               "block_factor", "block_size", "height_threshold"):
        assert kw in sig


def test_reference_pointpillars_over_our_stack(ref):
    """BASELINE config 4 (nuscenes/all.pp.largea): the unmodified reference VoxelNet (PillarFeatureNet +
    PointPillarsScatter + 3-block RPNV2 + class-agnostic axis-aligned NMS) vs our mirror, same weights."""
    import oracle_backend
    from google.protobuf import text_format
    from second.protos import pipeline_pb2
    from second_amd import synthetic as syn
    from second_amd.models import SecondDetector, ALL_PP_LARGEA
    train, _ = ref
    cfg = pipeline_pb2.TrainEvalPipelineConfig()
    text_format.Merge(open(os.path.join(REF, "second/configs/nuscenes/all.pp.largea.config")).read(), cfg)
    model_cfg = cfg.model.second
    for cs in model_cfg.target_assigner.class_settings:
        cs.nms_pre_max_size = 200
    torch.manual_seed(0)
    with oracle_backend.installed():
        net = train.build_network(model_cfg).eval()
        g = torch.Generator().manual_seed(1)
        for m in net.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.running_mean.copy_(torch.empty_like(m.running_mean).uniform_(-0.1, 0.1, generator=g))
                m.running_var.copy_(torch.empty_like(m.running_var).uniform_(0.5, 1.5, generator=g))
        cloud = syn.syn_nusc_cloud(0, num_points=20000, point_cloud_range=(-50, -50, -5, 50, 50, 3))
        vox = net.voxel_generator.generate(cloud, 30000)
        fm = [1, 50, 50]
        anchors = net.target_assigner.generate_anchors(fm)["anchors"].reshape(1, -1, 7)
        example = {"voxels": vox["voxels"], "num_points": vox["num_points_per_voxel"],
                   "coordinates": np.pad(vox["coordinates"], ((0, 0), (1, 0)), mode="constant", constant_values=0),
                   "anchors": anchors}
        ex = train.example_convert_to_torch(example, torch.float32, torch.device("cpu"))
        with torch.no_grad():
            ref_preds = net.network_forward(ex["voxels"], ex["num_points"], ex["coordinates"], 1)
        det = SecondDetector(dict(ALL_PP_LARGEA, nms_pre_max_size=200)).eval()
        missing = det.load_state_dict({k: v for k, v in net.state_dict().items() if k in det.state_dict()})
        assert not missing.missing_keys
        assert det.anchors.shape == (anchors.shape[1], 7)
        np.testing.assert_allclose(det.anchors.numpy(), anchors[0], rtol=0, atol=1e-5)
        with torch.no_grad():
            feats = det.voxel_feature_extractor(ex["voxels"], ex["num_points"], ex["coordinates"])
            ours_preds = det.network_forward(feats, ex["coordinates"], 1)
        for k in ("box_preds", "cls_preds", "dir_cls_preds"):
            np.testing.assert_allclose(ours_preds[k].numpy(), ref_preds[k].numpy(), rtol=1e-3, atol=1e-4)
        gen = torch.Generator().manual_seed(3)
        fake = {k: v.clone() for k, v in ref_preds.items()}
        fake["cls_preds"] = torch.randn(fake["cls_preds"].shape, generator=gen) * 0.8 - 3.2
        fake["box_preds"] = torch.randn(fake["box_preds"].shape, generator=gen) * 0.2
        fake["dir_cls_preds"] = torch.randn(fake["dir_cls_preds"].shape, generator=gen)
        with torch.no_grad():
            r = net.predict(ex, {k: v.clone() for k, v in fake.items()})[0]
            o = det.predict(fake, ex["anchors"].view(1, -1, 7))[0]
    assert r["scores"].shape[0] > 5
    np.testing.assert_allclose(o["scores"].numpy(), r["scores"].numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o["box3d_lidar"].numpy(), r["box3d_lidar"].numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(o["label_preds"].numpy(), r["label_preds"].numpy())


def test_reference_nuscenes_fhd_over_our_stack(ref):
    """BASELINE config 5 (nuscenes/all.fhd): block-filtered voxels (max 1 point), SpMiddleFHD on a
    1984x1984x40 grid, RPNV2 whose "upsample" is a stride-2 conv, 22 anchors per location."""
    import oracle_backend
    from google.protobuf import text_format
    from second.protos import pipeline_pb2
    from second_amd import synthetic as syn
    from second_amd.models import SecondDetector, ALL_FHD_NUSC
    train, _ = ref
    cfg = pipeline_pb2.TrainEvalPipelineConfig()
    text_format.Merge(open(os.path.join(REF, "second/configs/nuscenes/all.fhd.config")).read(), cfg)
    model_cfg = cfg.model.second
    torch.manual_seed(0)
    with oracle_backend.installed():
        net = train.build_network(model_cfg).eval()
        g = torch.Generator().manual_seed(1)
        for m in net.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.running_mean.copy_(torch.empty_like(m.running_mean).uniform_(-0.1, 0.1, generator=g))
                m.running_var.copy_(torch.empty_like(m.running_var).uniform_(0.5, 1.5, generator=g))
        vg = net.voxel_generator
        assert vg.grid_size.tolist() == [1984, 1984, 40]
        cloud = syn.syn_nusc_cloud(0, num_points=6000, point_cloud_range=(-49.6, -49.6, -5, 49.6, 49.6, 3))
        vox = vg.generate(cloud, 90000)
        det = SecondDetector(ALL_FHD_NUSC).eval()
        ours_vox = det.voxel_generator.generate(cloud, 90000)
        for k in ("voxels", "coordinates", "num_points_per_voxel"):
            np.testing.assert_array_equal(ours_vox[k], vox[k])
        assert 0 < vox["voxel_num"] < 6000          # the block filter removed something
        fm = [1, 124, 124]
        anchors = net.target_assigner.generate_anchors(fm)["anchors"].reshape(1, -1, 7)
        example = {"voxels": vox["voxels"], "num_points": vox["num_points_per_voxel"],
                   "coordinates": np.pad(vox["coordinates"], ((0, 0), (1, 0)), mode="constant", constant_values=0),
                   "anchors": anchors}
        ex = train.example_convert_to_torch(example, torch.float32, torch.device("cpu"))
        with torch.no_grad():
            ref_preds = net.network_forward(ex["voxels"], ex["num_points"], ex["coordinates"], 1)
        missing = det.load_state_dict({k: v for k, v in net.state_dict().items() if k in det.state_dict()})
        assert not missing.missing_keys
        assert det.anchors.shape == (anchors.shape[1], 7) and det.feature_map_size == fm
        np.testing.assert_allclose(det.anchors.numpy(), anchors[0], rtol=0, atol=1e-5)
        with torch.no_grad():
            feats = det.voxel_feature_extractor(ex["voxels"], ex["num_points"], ex["coordinates"])
            ours_preds = det.network_forward(feats, ex["coordinates"], 1)
        for k in ("box_preds", "cls_preds", "dir_cls_preds"):
            np.testing.assert_allclose(ours_preds[k].numpy(), ref_preds[k].numpy(), rtol=1e-3, atol=1e-4)


def test_accelerated_nms_patch_equals_reference_path(ref):
    """compat.accelerate_nms(): the device-resident rotate_nms / nms replacements return what the reference's own
    host round-trip functions return (same kept indices, same order) -- here both over the CPU oracle."""
    import oracle_backend
    from second_amd import compat
    rng = np.random.default_rng(0)
    n = 300
    boxes = np.concatenate([rng.uniform(0, 40, (n, 2)), rng.uniform(1.5, 4.5, (n, 2)), rng.uniform(-3.14, 3.14, (n, 1))], 1)
    scores = rng.permutation(n).astype(np.float32) / n           # distinct scores
    rb, sc = torch.from_numpy(boxes.astype(np.float32)), torch.from_numpy(scores)
    # axis-aligned boxes (x1, y1, x2, y2)
    ab = torch.from_numpy(np.concatenate([boxes[:, :2] - boxes[:, 2:4] / 2, boxes[:, :2] + boxes[:, 2:4] / 2], 1).astype(np.float32))
    with oracle_backend.installed():
        bto = compat.accelerate_nms()
        orig = bto._second_amd_original_nms
        try:
            bto._second_amd_force = True
            for pre, post, thr in ((None, None, 0.3), (200, 50, 0.01), (1000, 100, 0.5)):
                got = bto.rotate_nms(rb, sc, pre, post, thr)
                want = orig["rotate_nms"](rb, sc, pre, post, thr)
                assert got.dtype == torch.int64 and torch.equal(got, want) and len(got) > 3
                got = bto.nms(ab, sc, pre, post, thr)
                want = orig["nms"](ab, sc, pre, post, thr)
                assert torch.equal(got, want) and len(got) > 3
        finally:
            bto._second_amd_force = False
