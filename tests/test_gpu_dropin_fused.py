"""-m gpu: ``compat.accelerate_model`` -- the fused static-capacity, graph-captured pipeline BEHIND the reference's own
``VoxelNet.forward(example)`` (voxelnet.py:339-375, the call train.py:524 makes in evaluate()).

The network object is tests/reference_standin.py's (the GPU box has no reference checkout; tests/test_dropin_reference.py proves
in the build container that ``dropin.model_config`` reads the same configuration from it and from the real ``build_network``
result, and runs the same engine on the REAL network in its dynamic-shape mode).  Every case runs the object's own module-graph
forward (eager, dynamic shapes, first-touch rulebooks, ``.dense()``, torch RPN, torch formulation of predict) and the accelerated
forward on the same example and compares the returned lists: fp32 detections identical (count, order, labels; scores 1e-4,
boxes 1e-3) and head outputs within 1e-4; fp16 (``net.half()``) / forced bf16 by the detection-level rule bench.py uses."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def car():
    from e2e_trace import trained_like_detector
    from reference_standin import build_voxelnet
    from second_amd import synthetic as syn
    from second_amd.models import CAR_FHD
    clouds = [syn.syn_kitti_cloud(s, num_points=9000, num_voxels=8000) for s in range(3)]
    small = [syn.syn_kitti_cloud(10 + s, num_points=2500, num_voxels=2200) for s in range(2)]
    like = trained_like_detector(CAR_FHD, clouds[0])            # CPU, fp32: distinct scores, empty regions below the threshold

    def make():
        net = build_voxelnet(CAR_FHD)
        net.load_state_dict(like.state_dict())
        return net.eval().cuda()
    return make, clouds, small


def _canonical(d):
    """rows ordered by (score desc, x, y, z): two anchors of one location can tie in score, and tie order is implementation
    defined in the reference itself (torch.topk)"""
    b, s, l = d["box3d_lidar"].cpu().numpy(), d["scores"].cpu().numpy(), d["label_preds"].cpu().numpy()
    o = np.lexsort((b[:, 2].round(3), b[:, 1].round(3), b[:, 0].round(3), -s.round(4)))
    return dict(d, box3d_lidar=d["box3d_lidar"][torch.from_numpy(o).to(d["box3d_lidar"].device)],
                scores=d["scores"][torch.from_numpy(o).to(d["scores"].device)],
                label_preds=d["label_preds"][torch.from_numpy(o).to(d["label_preds"].device)])


def _same(got, want, score_tol=1e-4, box_tol=2e-3, canonical=False):
    assert isinstance(got, list) and len(got) == len(want)
    if canonical:
        got, want = [_canonical(g) for g in got], [_canonical(w) for w in want]
    for g, w in zip(got, want):
        assert set(g) >= {"box3d_lidar", "scores", "label_preds", "metadata"} and g["metadata"] == w["metadata"]
        assert g["box3d_lidar"].is_cuda and g["box3d_lidar"].dtype == torch.float32 and g["scores"].dtype == torch.float32
        assert g["label_preds"].dtype == torch.int64
        assert g["box3d_lidar"].shape == w["box3d_lidar"].shape, (g["box3d_lidar"].shape, w["box3d_lidar"].shape)
        np.testing.assert_allclose(g["scores"].cpu().numpy(), w["scores"].cpu().numpy(), rtol=score_tol, atol=score_tol)
        np.testing.assert_allclose(g["box3d_lidar"].cpu().numpy(), w["box3d_lidar"].cpu().numpy(), rtol=1e-3, atol=box_tol)
        np.testing.assert_array_equal(g["label_preds"].cpu().numpy(), w["label_preds"].cpu().numpy())


def _found(got, want, dist=0.15, dscore=0.08):
    """share of ``want``'s detections that ``got`` holds too (centre within ``dist`` m, score within ``dscore``)"""
    hit = tot = 0
    for g, w in zip(got, want):
        gb, wb = g["box3d_lidar"].cpu().numpy(), w["box3d_lidar"].cpu().numpy()
        gs, ws = g["scores"].cpu().numpy(), w["scores"].cpu().numpy()
        tot += len(wb)
        for b, s in zip(wb, ws):
            if len(gb) and ((np.linalg.norm(gb[:, :3] - b[:3], axis=1) < dist) & (np.abs(gs - s) < dscore)).any():
                hit += 1
    return hit, tot


def test_fp32_forward_example_is_served_by_one_graph_and_returns_the_module_paths_detections(car):
    from reference_standin import example_of
    from second_amd import compat
    make, clouds, small = car
    net = make()
    ex = example_of(net, clouds[:2], "cuda")
    with torch.no_grad():
        want = net(ex)
        heads_want = net.network_forward(ex["voxels"], ex["num_points"], ex["coordinates"], 2)
    assert sum(w["scores"].shape[0] for w in want) >= 6
    assert compat.accelerate_model(net) is net
    eng = net._second_amd_engine
    with torch.no_grad():
        got = net(ex)
    assert eng.stats == dict(eng.stats, fused_calls=1, original_calls=0, adoptions=1, captures=1, overflow_recaptures=0)
    assert eng.run_dtype() is None and eng._det._infer_dtype in (None, torch.float32)
    _same(got, want)
    # head outputs of the adopted pipeline (its own modules, sorted rulebooks) vs the module graph's: <= 1e-4
    det = eng._det
    with torch.no_grad():
        feats = ex["voxels"][:, :, :4].sum(1) / ex["num_points"].float().unsqueeze(1)
        heads_got = det.network_forward(feats, ex["coordinates"], 2)
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        a, b = heads_got[k].float().cpu().numpy(), heads_want[k].float().cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-4 * float(np.abs(b).max()), err_msg=k)
    # another example of the same batch size (different row count): same session, same graph; the first call's tensors stay intact
    keep = [g["box3d_lidar"].clone() for g in got]
    ex2 = example_of(net, clouds[1:3], "cuda")
    with torch.no_grad():
        got2 = net(ex2)
        want2 = net._second_amd_original_forward(ex2)
    assert eng.stats["captures"] == 1 and eng.stats["fused_calls"] == 2 and len(eng._sessions) == 1, eng.stats
    _same(got2, want2)
    for g, k in zip(got, keep):
        assert torch.equal(g["box3d_lidar"], k), "a returned tensor aliases the session's static buffers"
    # training mode: the reference's own forward (here the stand-in's) is called
    net.train()
    assert not eng.accepts(ex)
    net.eval()


def test_capacities_follow_the_data_overflow_recaptures_and_row_growth_opens_a_new_session(car):
    from reference_standin import example_of
    from second_amd import compat
    make, clouds, small = car
    net = compat.accelerate_model(make())
    eng = net._second_amd_engine
    eng.row_bucket = 4096
    ex_small = example_of(net, small, "cuda")
    n_small = ex_small["voxels"].shape[0]
    with torch.no_grad():
        _same(net(ex_small), net._second_amd_original_forward(ex_small))
    (sess,) = eng._sessions.values()
    assert sess.cap == -(-n_small // 4096) * 4096 and eng.stats["captures"] == 1
    caps0 = list(sess.caps)
    # more rows than the session holds -> a new, larger session; strided capacities calibrated on the new data
    ex_big = example_of(net, clouds[:2], "cuda")
    assert ex_big["voxels"].shape[0] > sess.cap
    with torch.no_grad():
        _same(net(ex_big), net._second_amd_original_forward(ex_big))
    (sess2,) = eng._sessions.values()
    assert sess2 is not sess and sess2.cap >= ex_big["voxels"].shape[0] and eng.stats["captures"] == 2
    # strided capacities too small for the data (forced): the overflow is seen in the counters that travel with the results,
    # the graph is re-captured with capacities from the raw counts, and the call still returns the right detections
    sess2.caps = [256 for _ in sess2.caps]
    sess2.build(True)
    before = eng.stats["overflow_recaptures"]
    with torch.no_grad():
        got = net(ex_big)
    assert eng.stats["overflow_recaptures"] == before + 1 and all(c > 256 for c in sess2.caps) and caps0
    with torch.no_grad():
        _same(got, net._second_amd_original_forward(ex_big))


def test_half_network_runs_the_fp16_pipeline_and_bf16_can_be_forced(car):
    """train.py:468-472: ``net.half(); net.metrics_to_float(); net.convert_norm_to_float(net)`` and float16 examples."""
    from reference_standin import example_of
    from second_amd import compat
    make, clouds, small = car
    ref32 = make()
    ex32 = example_of(ref32, clouds[:2], "cuda")
    with torch.no_grad():
        want = ref32(ex32)
    net = make().half()
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.float()
    compat.accelerate_model(net)
    eng = net._second_amd_engine
    ex16 = example_of(net, clouds[:2], "cuda", dtype=torch.float16)
    with torch.no_grad():
        got = net(ex16)
    assert eng.run_dtype() == torch.float16 and eng._det._infer_dtype == torch.float16 and eng.stats["captures"] == 1
    hit, tot = _found(got, want)
    assert tot >= 6 and hit >= 0.85 * tot, (hit, tot)
    assert abs(sum(g["scores"].shape[0] for g in got) - tot) <= 2 * len(got)
    nb = compat.accelerate_model(make(), dtype=torch.bfloat16)
    with torch.no_grad():
        gotb = nb(ex32)
    assert nb._second_amd_engine._det._infer_dtype == torch.bfloat16
    hit, tot = _found(gotb, want)
    assert hit >= 0.85 * tot, (hit, tot)


def test_a_checkpoint_loaded_after_acceleration_is_adopted(car):
    """evaluate() restores the checkpoint AFTER build_network (train.py:476-480): the engine follows parameter changes."""
    from reference_standin import example_of
    from second_amd import compat
    make, clouds, small = car
    net = compat.accelerate_model(make())
    eng = net._second_amd_engine
    ex = example_of(net, clouds[:2], "cuda", metadata=False)
    with torch.no_grad():
        first = net(ex)
    assert first[0]["metadata"] is None
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    sd["rpn.conv_cls.bias"] = sd["rpn.conv_cls.bias"] + 0.6
    net.load_state_dict(sd)
    with torch.no_grad():
        got = net(ex)
        want = net._second_amd_original_forward(ex)
    assert eng.stats["adoptions"] == 2
    assert sum(g["scores"].shape[0] for g in got) > sum(g["scores"].shape[0] for g in first)
    _same(got, want)


def test_pointpillars_network_is_accelerated_too():
    """BASELINE config 4's network (PillarFeatureNet + PointPillarsScatter + three-block RPNV2, ten classes, axis-aligned NMS)."""
    from reference_standin import build_voxelnet, example_of
    from second_amd import compat, synthetic as syn
    from second_amd.models import ALL_PP_LARGEA
    torch.manual_seed(0)
    net = build_voxelnet(ALL_PP_LARGEA)
    syn.randomise_like_trained(net, seed=1)
    net = net.eval().cuda()
    clouds = [syn.syn_nusc_cloud(s, num_points=40000, point_cloud_range=(-50, -50, -5, 50, 50, 3)) for s in range(2)]
    ex = example_of(net, clouds, "cuda")
    with torch.no_grad():
        p = net.network_forward(ex["voxels"][ex["coordinates"][:, 0] == 0], ex["num_points"][ex["coordinates"][:, 0] == 0],
                                ex["coordinates"][ex["coordinates"][:, 0] == 0], 1)
        syn.sharpen_heads(net, p["cls_preds"].float(), p["box_preds"].float())
        want = net(ex)
    assert sum(w["scores"].shape[0] for w in want) >= 4
    compat.accelerate_model(net)
    eng = net._second_amd_engine
    assert eng.cfg["middle"] == "PointPillarsScatter" and eng.cfg["num_anchor_per_loc"] == 12
    with torch.no_grad():
        got = net(ex)
    assert eng.stats["captures"] == 1 and eng.stats["fused_calls"] == 1
    _same(got, want, score_tol=2e-4, box_tol=5e-3, canonical=True)


def test_anchor_sets_are_checked_on_the_device_and_followed(car):
    """The example carries its anchors (voxelnet.py:358).  A new tensor with the same content costs nothing on the host (the
    comparison runs on the device and its flag travels with the results); a different anchor set shared by the frames is adopted and
    the call re-run; per-frame anchor sets take the reference's own forward."""
    from reference_standin import example_of
    from second_amd import compat
    make, clouds, small = car
    net = compat.accelerate_model(make())
    eng = net._second_amd_engine
    ex = example_of(net, clouds[:2], "cuda")
    with torch.no_grad():
        first = net(ex)
        ex_b = dict(ex, anchors=ex["anchors"].clone())                    # what a data loader hands over: a fresh tensor per batch
        again = net(ex_b)
    assert eng.stats["anchor_refreshes"] == 0 and eng.stats["fused_calls"] == 2
    _same(again, first)
    shifted = ex["anchors"].clone()
    shifted[..., 0] += 0.5                                                # every anchor half a metre further along x
    ex_c = dict(ex, anchors=shifted)
    with torch.no_grad():
        got = net(ex_c)
        want = net._second_amd_original_forward(ex_c)
    assert eng.stats["anchor_refreshes"] == 1 and eng.stats["original_calls"] == 0
    _same(got, want)
    assert abs(float(got[0]["box3d_lidar"][0, 0] - first[0]["box3d_lidar"][0, 0]) - 0.5) < 1e-3
    per_frame = ex["anchors"].clone()
    per_frame[1, :, 1] += 0.25                                            # frame 1 has its own anchor set
    ex_d = dict(ex, anchors=per_frame)
    with torch.no_grad():
        got = net(ex_d)
        want = net._second_amd_original_forward(ex_d)
    assert eng.stats["original_calls"] == 1
    _same(got, want)


def test_weight_updates_through_dot_data_are_seen(car):
    """The reference's own optimizers write weights through ``.data`` (torchplus/train/fastai_optim.py: ``p.data.mul_(1 - wd * lr)``,
    ``model.data.copy_(master)``; optim.py: ``p.data.copy_``) -- no version counter moves, BatchNorm running statistics of a frozen
    network do not move either.  Every call carries a content check of the adopted tensors (norms compared inside the graph): a
    changed head bias must change the detections of the NEXT call, not keep serving the packed copy."""
    from reference_standin import example_of
    from second_amd import compat
    make, clouds, small = car
    net = compat.accelerate_model(make())
    eng = net._second_amd_engine
    ex = example_of(net, clouds[:2], "cuda")
    with torch.no_grad():
        first = net(ex)
        v0 = [t._version for t in eng._watch]
        net.rpn.conv_cls.bias.data.add_(0.6)                              # what `p.data.copy_(master)` amounts to
        net.middle_feature_extractor.middle_conv[0].weight.data.mul_(1.0 - 1e-3)
        assert [t._version for t in eng._watch] == v0                     # nothing for a version-based key to see
        got = net(ex)
        want = net._second_amd_original_forward(ex)
    assert eng.stats["content_readoptions"] == 1 and eng.stats["adoptions"] == 2
    assert sum(g["scores"].shape[0] for g in got) > sum(g["scores"].shape[0] for g in first)
    _same(got, want)
    with torch.no_grad():
        net(ex)
    assert eng.stats["adoptions"] == 2                                    # unchanged weights: no further adoption
    eng.refresh(force=True)
    assert eng.stats["adoptions"] == 3


def test_anchors_edited_in_place_or_resized_are_followed(car):
    """Same tensor object, same address, new content (and then another anchor COUNT): the comparison runs on the device every call."""
    from reference_standin import example_of
    from second_amd import compat
    make, clouds, small = car
    net = compat.accelerate_model(make())
    eng = net._second_amd_engine
    ex = example_of(net, clouds[:2], "cuda")
    with torch.no_grad():
        first = net(ex)
        ex["anchors"].data[..., 0] += 0.5                                 # in place, through .data: same data_ptr, same _version
        got = net(ex)
        want = net._second_amd_original_forward(ex)
    assert eng.stats["anchor_refreshes"] == 1 and eng.stats["original_calls"] == 0
    _same(got, want)
    assert abs(float(got[0]["box3d_lidar"][0, 0] - first[0]["box3d_lidar"][0, 0]) - 0.5) < 1e-3
    fewer = dict(ex, anchors=ex["anchors"][:, :-2].contiguous())          # an anchor table of another length never reaches the fused graph
    with torch.no_grad(), pytest.raises(Exception):
        net._second_amd_original_forward(fewer)                           # (the reference asserts num_anchors == num_output, voxelnet.py:367)


def test_deferred_calls_return_at_once_and_fill_themselves_when_read(car):
    """``accelerate_model(net, deferred=True)``: what evaluate() does (train.py:519-539) -- collect ``net(example)`` in a list for a
    whole loop, look inside afterwards.  Every deferred result must equal the synchronous engine's for the same example, whatever
    ran in between (the calls alternate between two sessions on their own streams; results are clones, not views of a session)."""
    import pickle
    from reference_standin import example_of
    from second_amd import compat, dropin
    make, clouds, small = car
    sync = compat.accelerate_model(make())
    net = compat.accelerate_model(make(), deferred=True)
    eng = net._second_amd_engine
    examples = [example_of(net, clouds[:2], "cuda"), example_of(net, clouds[1:3], "cuda"), example_of(net, [clouds[2], clouds[0]], "cuda")]
    with torch.no_grad():
        want = [sync(ex) for ex in examples]
        collected = []
        for rep in range(3):
            for ex in examples:
                collected.append(net(ex))                                  # nothing is read here
    assert all(isinstance(d, dropin.DeferredDetection) for r in collected for d in r)
    assert eng.stats["deferred_calls"] == 9 and eng.stats["original_calls"] == 0
    assert eng.stats["captures"] == eng.lanes == 3                         # one graph per lane
    for i, got in enumerate(collected):
        _same(got, want[i % 3])
        assert got[0]["metadata"] == examples[i % 3]["metadata"][0]
    assert eng.stats["fused_calls"] == 9 and eng.stats["deferred_redone"] == 0 and not eng._pending
    blob = pickle.loads(pickle.dumps([{k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in d.items()} for d in collected[0]]))
    assert type(blob[0]) is dict and set(blob[0]) == {"box3d_lidar", "scores", "label_preds", "metadata"}
    assert type(pickle.loads(pickle.dumps(collected[1][0]))) is dict


def test_deferred_calls_that_fail_a_check_are_redone_from_their_example(car):
    """Overflow of a strided layer's capacity, another anchor table, weights changed through ``.data``: found when the call is
    resolved, settled through the synchronous path, and the lanes take the larger capacities for the calls that follow."""
    from reference_standin import example_of
    from second_amd import compat
    make, clouds, small = car
    sync = compat.accelerate_model(make())
    net = compat.accelerate_model(make(), deferred=True)
    eng = net._second_amd_engine
    ex_small, ex_big = example_of(net, small, "cuda"), example_of(net, clouds[:2], "cuda")
    with torch.no_grad():
        want = sync(ex_small)
        for r in [net(ex_small) for _ in range(eng.lanes)]:                  # every lane has its session now
            _same(r, want)
        for sess in eng._sessions.values():                                  # squeeze the lanes: the next calls overflow
            sess.grow_to = None
            sess.caps = [256 for _ in sess.caps]
            sess.build(True)
        for r in [net(ex_small) for _ in range(eng.lanes)]:
            _same(r, want)
        assert eng.stats["deferred_redone"] == eng.lanes
        for r in [net(ex_small) for _ in range(2 * eng.lanes)]:             # lanes rebuilt with the settled capacities: no redo any more
            _same(r, want)
        assert eng.stats["deferred_redone"] == eng.lanes
        net.rpn.conv_cls.bias.data.add_(0.6)
        sync.rpn.conv_cls.bias.data.add_(0.6)
        g = net(ex_big)
        _same(g, sync(ex_big))
    assert eng.stats["original_calls"] == 0
