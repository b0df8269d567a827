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


def _example_of(train, net, clouds, max_voxels=40000, fm=(1, 200, 176)):
    vox = [net.voxel_generator.generate(c, max_voxels) for c in clouds]
    anchors = net.target_assigner.generate_anchors(list(fm))["anchors"].reshape(1, -1, 7)
    example = {
        "voxels": np.concatenate([v["voxels"] for v in vox]),
        "num_points": np.concatenate([v["num_points_per_voxel"] for v in vox]),
        "coordinates": np.concatenate([np.pad(v["coordinates"], ((0, 0), (1, 0)), mode="constant", constant_values=b)
                                       for b, v in enumerate(vox)]),
        "anchors": np.repeat(anchors, len(clouds), 0),
        "metadata": [{"image_idx": 10 + b} for b in range(len(clouds))],
    }
    return train.example_convert_to_torch(example, torch.float32, torch.device("cpu"))


def test_accelerate_model_serves_the_reference_voxelnet(ref):
    """compat.accelerate_model on the network the reference's own build_network returns: the configuration is read off the
    object, the parameters move by state-dict key, and net(example) (voxelnet.py:339-375) returns what the original forward
    returns -- same list of dicts, same dtypes, metadata passed through.  Dynamic-shape mode here (CPU, oracle backend); the
    static-capacity / graph mode of the same engine is tests/test_gpu_dropin_fused.py."""
    import oracle_backend
    from e2e_trace import trained_like_detector
    from second_amd import compat, dropin, synthetic as syn
    from second_amd.models import CAR_FHD
    train, cfg = ref
    model_cfg = cfg.model.second
    model_cfg.target_assigner.class_settings[0].nms_pre_max_size = 150
    clouds = [syn.syn_kitti_cloud(s, num_points=6000, num_voxels=5000) for s in range(2)]
    with oracle_backend.installed():
        net = train.build_network(model_cfg).eval()
        like = trained_like_detector(dict(CAR_FHD, nms_pre_max_size=150), clouds[0])      # distinct scores: no tie-order dependence
        net.load_state_dict(like.state_dict(), strict=False)
        ex = _example_of(train, net, clouds)
        with torch.no_grad():
            want = net(ex)
        assert sum(w["box3d_lidar"].shape[0] for w in want) >= 4
        assert compat.accelerate_model(net) is net and compat.accelerate_model(net) is net          # idempotent
        eng = net._second_amd_engine
        c = eng.cfg
        assert c["middle"] == "SpMiddleFHD" and c["downsample_factor"] == 8 and c["num_anchor_per_loc"] == 2
        assert c["nms_pre_max_size"] == 150 and c["use_rotate_nms"] and abs(c["nms_iou_threshold"] - 0.01) < 1e-6
        assert c["rpn"]["layer_nums"] == [5] and c["max_points_per_voxel"] == 5 and c["direction_limit_offset"] == 1.0
        np.testing.assert_allclose(c["post_center_range"], [0, -40, -2.2, 70.4, 40, 0.8], rtol=1e-6)
        with torch.no_grad():
            got = net(ex)
        assert eng.stats["fused_calls"] == 1 and eng.stats["original_calls"] == 0 and eng.stats["adoptions"] == 1
        assert isinstance(got, list) and len(got) == 2

        def same(a, b):
            for g, w in zip(a, b):
                assert set(g) == set(w) and g["metadata"] == w["metadata"]
                assert g["box3d_lidar"].dtype == w["box3d_lidar"].dtype == torch.float32 and g["label_preds"].dtype == w["label_preds"].dtype
                assert g["scores"].shape == w["scores"].shape
                np.testing.assert_allclose(g["scores"].numpy(), w["scores"].numpy(), rtol=1e-4, atol=1e-5)
                np.testing.assert_allclose(g["box3d_lidar"].numpy(), w["box3d_lidar"].numpy(), rtol=1e-4, atol=1e-4)
                np.testing.assert_array_equal(g["label_preds"].numpy(), w["label_preds"].numpy())
        same(got, want)
        # a checkpoint loaded AFTER acceleration is followed (evaluate() restores after build_network, train.py:476-480)
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        sd["rpn.conv_cls.bias"] = sd["rpn.conv_cls.bias"] + 0.4
        net.load_state_dict(sd)
        with torch.no_grad():
            want2 = net._second_amd_original_forward(ex)
            got2 = net(ex)
        assert eng.stats["adoptions"] == 2 and sum(w["scores"].shape[0] for w in want2) != sum(w["scores"].shape[0] for w in want)
        same(got2, want2)
        # training mode and DataParallel-padded examples keep the reference's own forward
        assert not eng.accepts(dict(ex, num_points=ex["num_points"].view(1, -1)))
        assert not eng.accepts(dict(ex, anchors_mask=torch.ones(2, ex["anchors"].shape[1], dtype=torch.bool)))
        net.train()
        assert not eng.accepts(ex)
        net.eval()


@pytest.mark.parametrize("rel, want", [
    ("nuscenes/all.pp.largea.config", dict(middle="PointPillarsScatter", vfe="PillarFeatureNet", vfe_filters=[64], downsample_factor=8,
                                           num_anchor_per_loc=12, num_class=10, use_rotate_nms=False, max_points_per_voxel=60)),
    ("nuscenes/all.fhd.config", dict(middle="SpMiddleFHD", downsample_factor=16, num_anchor_per_loc=20, num_class=10,
                                     use_rotate_nms=False, max_points_per_voxel=1)),
])
def test_model_config_of_the_other_benchmark_networks(ref, rel, want):
    """dropin.model_config on BASELINE configs 4 / 5 as the reference builds them, and the adopted detector takes their state dict."""
    import oracle_backend
    from google.protobuf import text_format
    from second.protos import pipeline_pb2
    from second_amd import dropin
    train, _ = ref
    cfg = pipeline_pb2.TrainEvalPipelineConfig()
    text_format.Merge(open(os.path.join(REF, "second/configs", rel)).read(), cfg)
    with oracle_backend.installed():
        net = train.build_network(cfg.model.second).eval()
        c = dropin.model_config(net)
        for k, v in want.items():
            assert c[k] == v, (k, c[k], v)
        eng = dropin.FusedVoxelNet(net)
        det = eng.refresh()
        assert det.feature_map_size == ([1, 50, 50] if "pp" in rel else [1, 124, 124])
        for k, v in det.state_dict().items():
            if k in net.state_dict() and v.is_floating_point():
                assert torch.equal(v, net.state_dict()[k].float()), k
    # a network outside the fused path says why
    net._multiclass_nms = True
    with pytest.raises(dropin.NotAccelerable, match="multiclass_nms"):
        dropin.model_config(net)
    from second_amd import compat
    assert compat.accelerate_model(net, strict=False) is net and getattr(net, "_second_amd_engine", None) is None


@pytest.mark.parametrize("rel, name", [("car.fhd.config", "CAR_FHD"), ("nuscenes/all.pp.largea.config", "ALL_PP_LARGEA"),
                                       ("nuscenes/all.fhd.config", "ALL_FHD_NUSC")])
def test_gpu_box_standin_is_shaped_like_the_reference_network(ref, rel, name):
    """tests/reference_standin.py (what the -m gpu tests and bench.py's dropin_fused leg accelerate, because the GPU box has no
    reference checkout) against the real build_network result: same sub-module type names, same state-dict keys and shapes,
    and dropin.model_config reads the SAME configuration from both objects."""
    import oracle_backend
    import reference_standin
    from google.protobuf import text_format
    from second.protos import pipeline_pb2
    from second_amd import dropin, models
    train, _ = ref
    cfg = pipeline_pb2.TrainEvalPipelineConfig()
    text_format.Merge(open(os.path.join(REF, "second/configs", rel)).read(), cfg)
    with oracle_backend.installed():
        real = train.build_network(cfg.model.second).eval()
        mcfg = dict(getattr(models, name), max_voxels=int(real.voxel_generator._max_voxels))
        fake = reference_standin.build_voxelnet(mcfg).eval()
    for part in ("voxel_feature_extractor", "middle_feature_extractor", "rpn"):
        assert type(getattr(real, part)).__name__ == type(getattr(fake, part)).__name__
    rs, fs = real.state_dict(), fake.state_dict()
    keys = lambda sd: {k: tuple(v.shape) for k, v in sd.items() if k.split(".")[0] in ("voxel_feature_extractor", "middle_feature_extractor", "rpn")}
    assert keys(rs) == keys(fs)
    c_real, c_fake = dropin.model_config(real), dropin.model_config(fake)
    c_real.pop("name"), c_fake.pop("name")
    for k in c_real:
        if isinstance(c_real[k], list) and c_real[k] and isinstance(c_real[k][0], float):
            np.testing.assert_allclose(c_fake[k], c_real[k], rtol=1e-6, err_msg=k)
        else:
            assert c_fake[k] == c_real[k], (k, c_fake[k], c_real[k])
    assert set(c_real) == set(c_fake)
    # the attributes VoxelNet.predict reads exist on the stand-in under the same names
    for attr in ("_num_class", "_use_rotate_nms", "_multiclass_nms", "_nms_score_thresholds", "_nms_pre_max_sizes", "_nms_post_max_sizes",
                 "_nms_iou_thresholds", "_use_sigmoid_score", "_encode_background_as_zeros", "_use_direction_classifier",
                 "_post_center_range", "_dir_offset", "_num_direction_bins", "_dir_limit_offset", "_box_coder", "target_assigner",
                 "voxel_generator"):
        assert hasattr(real, attr) and hasattr(fake, attr), attr


def test_standin_training_forward_is_the_reference_loss(ref):
    """What the -m gpu training tests and bench.py's dropin_train leg compare against: reference_standin.standin_loss must be
    VoxelNet.loss (voxelnet.py:239-312) of the REAL network, and dropin_train.train_config must read the same loss settings from
    both objects."""
    import oracle_backend
    import reference_standin
    from second_amd import dropin_train, models
    train, cfg = ref
    with oracle_backend.installed():
        real = train.build_network(cfg.model.second).train()
        fake = reference_standin.build_voxelnet(dict(models.CAR_FHD)).train()
    c_real, c_fake = dropin_train.train_config(real), dropin_train.train_config(fake)
    assert set(c_real) == set(c_fake)
    for k in c_real:
        np.testing.assert_allclose(np.asarray(c_fake[k], np.float64), np.asarray(c_real[k], np.float64), rtol=1e-6, err_msg=k)
    g = torch.Generator().manual_seed(7)
    b, a, h, w = 2, 2, 20, 16
    n = a * h * w
    preds = {"box_preds": (torch.randn(b, a, h, w, 7, generator=g) * 0.3), "cls_preds": torch.randn(b, a, h, w, 1, generator=g) * 2 - 2,
             "dir_cls_preds": torch.randn(b, a, h, w, 2, generator=g)}
    labels = (torch.rand(b, n, generator=g) < 0.02).int() - (torch.rand(b, n, generator=g) < 0.1).int()      # 1 / 0 / -1, a few overlaps -> 0
    ex = {"labels": labels, "reg_targets": torch.randn(b, n, 7, generator=g) * 0.4, "importance": torch.rand(b, n, generator=g) + 0.5,
          "anchors": torch.randn(b, n, 7, generator=g)}
    pr = {k: v.clone().requires_grad_() for k, v in preds.items()}
    pf = {k: v.clone().requires_grad_() for k, v in preds.items()}
    r = real.loss(ex, pr)
    f = reference_standin.standin_loss(fake, ex, pf)
    assert set(r) == set(f)
    for k in r:
        np.testing.assert_allclose(f[k].detach().float().numpy(), r[k].detach().float().numpy(), rtol=1e-5, atol=1e-7, err_msg=k)
    r["loss"].backward(), f["loss"].backward()
    for k in pr:
        np.testing.assert_allclose(pf[k].grad.numpy(), pr[k].grad.numpy(), rtol=1e-4, atol=1e-8, err_msg=k)


def test_zero_call_acceleration_through_the_import_hook(tmp_path):
    """SEC_ACCELERATE_MODEL=1 + `import spconv`: the reference's VoxelNet class is wrapped when its module is imported; a network
    built by the unmodified build_network serves net(example) from the fused engine without any call into this package."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent(f"""
        import os, sys
        sys.path[:0] = [{os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "second.pytorch_amd")!r},
                        {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}, {os.path.dirname(os.path.abspath(__file__))!r}]
        import numpy as np, torch
        from second_amd import compat
        compat.install({REF!r})
        import spconv                                   # installs the hook (SEC_ACCELERATE_MODEL=1)
        from google.protobuf import text_format
        from second.protos import pipeline_pb2
        import second.pytorch.train as train             # imports second.pytorch.models.voxelnet -> VoxelNet.forward is wrapped
        import oracle_backend
        from second_amd import synthetic as syn
        cfg = pipeline_pb2.TrainEvalPipelineConfig()
        text_format.Merge(open(os.path.join({REF!r}, "second/configs/car.fhd.config")).read(), cfg)
        cfg.model.second.target_assigner.class_settings[0].nms_pre_max_size = 100
        with oracle_backend.installed():
            net = train.build_network(cfg.model.second)
            cloud = syn.syn_kitti_cloud(0, num_points=3000, num_voxels=2500)
            vox = net.voxel_generator.generate(cloud, 40000)
            anchors = net.target_assigner.generate_anchors([1, 200, 176])["anchors"].reshape(1, -1, 7)
            ex = train.example_convert_to_torch({{"voxels": vox["voxels"], "num_points": vox["num_points_per_voxel"],
                    "coordinates": np.pad(vox["coordinates"], ((0, 0), (1, 0))), "anchors": anchors}}, torch.float32, torch.device("cpu"))
            net.train()
            assert "_second_amd_engine" not in net.__dict__
            net.eval()
            with torch.no_grad():
                out = net(ex)
            eng = net.__dict__["_second_amd_engine"]
            assert eng and eng.stats["fused_calls"] == 1 and eng.stats["original_calls"] == 0, eng.stats
            assert isinstance(out, list) and set(out[0]) == {{"box3d_lidar", "scores", "label_preds", "metadata"}}
        print("HOOK_OK")
    """)
    env = dict(os.environ, SEC_ACCELERATE_MODEL="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "HOOK_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
