"""Background tiles of the dense RPN (sec_rpn_tile_live + sec_conv2d_nhwc_tiles / _gather): a tile of conv j's output that no site of
the sparse middle can reach within j + 1 steps holds exactly what the network computes for an EMPTY frame at that position, so only
the reachable tiles are convolved and the others are copied from the empty frame's activations.  The tile lists are checked
against a numpy dilation; the RPN with and without the skipping must agree BIT FOR BIT on networks whose background is not zero
(rpn.py:486-497 semantics: Conv2d 3x3 + BatchNorm2d + ReLU with arbitrary statistics)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from second_amd import ops as o
    return o


def _dilate(m):
    p = np.zeros((m.shape[0] + 2, m.shape[1] + 2), bool)
    p[1:-1, 1:-1] = m
    out = np.zeros_like(m)
    for dy in range(3):
        for dx in range(3):
            out |= p[dy:dy + m.shape[0], dx:dx + m.shape[1]]
    return out


def _live_reference(sites, layers):
    """sites [B, H, W] bool -> live [layers, B, tiles]; layer 0 = the first (gathered) conv"""
    b, h, w = sites.shape
    ty, tx = (h + 7) // 8, (w + 15) // 16
    out = np.zeros((layers, b, ty * tx), np.uint8)
    for f in range(b):
        cur = sites[f]
        for l in range(layers):
            cur = _dilate(cur)                       # conv l sees the image within l + 1 steps
            pad = np.zeros((ty * 8, tx * 16), bool)
            pad[:h, :w] = cur
            out[l, f] = pad.reshape(ty, 8, tx, 16).any(axis=(1, 3)).reshape(-1)
    return out


@pytest.mark.parametrize("batch,h,w,n", [(3, 200, 176, 900), (2, 37, 50, 12), (1, 8, 16, 1), (2, 64, 33, 0), (2, 120, 97, 3000),
                                          (1, 200, 176, 1), (8, 200, 176, 6000), (1, 400, 400, 5000)])
def test_rpn_tile_live_matches_a_numpy_dilation(ops, batch, h, w, n):
    rng = np.random.default_rng(h * 7 + n)
    m = np.zeros((batch, 2, h, w), np.int32)
    if n:
        b, z = rng.integers(0, batch, n), rng.integers(0, 2, n)
        y, x = rng.integers(0, h, n), rng.integers(0, w, n)
        if n > 100:                                  # clusters: whole regions stay empty; plus the four corners
            y = y % max(h // 3, 1)
            y[:4], x[:4] = [0, 0, h - 1, h - 1], [0, w - 1, 0, w - 1]
        m[b, z, y, x] = rng.integers(1, 1000, n)
        if batch > 1:
            m[batch - 1] = 0                          # an empty frame
    order, counts = ops.rpn_tile_live(torch.from_numpy(m).cuda(), 6)
    ref = _live_reference((m != 0).any(1), 6)
    order, counts = order.cpu().numpy().astype(np.int64), counts.cpu().numpy()
    tiles = ref.shape[2]
    np.testing.assert_array_equal(counts, ref.sum(2))
    for l in range(6):
        for f in range(batch):
            c = counts[l, f]
            np.testing.assert_array_equal(order[l, f, :c], np.flatnonzero(ref[l, f]))                 # live: ascending
            np.testing.assert_array_equal(order[l, f, c:][::-1], np.flatnonzero(ref[l, f] == 0))     # background: from the end
    if h >= 24 and w >= 48 and n <= 1:
        assert counts[1].sum() < batch * tiles            # an empty / one-site frame keeps background tiles
    # the masks of the lazy consumers (sec_rpn_tile_live_masks): same lists, plus per conv l >= 1 and tile which of its 3 x 3
    # neighbours conv l - 1 wrote (outside the image counts as written); by list rank for the live tiles and by tile index for all
    order2, counts2, nbr = ops.rpn_tile_live(torch.from_numpy(m).cuda(), 6, masks=True)
    np.testing.assert_array_equal(order2.cpu().numpy().astype(np.int64), order)
    np.testing.assert_array_equal(counts2.cpu().numpy(), counts)
    nbr = nbr.cpu().numpy().astype(np.int64) & 0xffff
    ty, tx = (h + 7) // 8, (w + 15) // 16
    for l in range(1, 6):
        prev = np.ones((batch, ty + 2, tx + 2), np.int64)
        prev[:, 1:-1, 1:-1] = ref[l - 1].reshape(batch, ty, tx)
        want = np.zeros((batch, ty, tx), np.int64)
        for q in range(9):
            want |= prev[:, q // 3:q // 3 + ty, q % 3:q % 3 + tx] << q
        want = want.reshape(batch, tiles)
        np.testing.assert_array_equal(nbr[l, 1], want)
        for f in range(batch):
            c = counts[l, f]
            np.testing.assert_array_equal(nbr[l, 0, f, :c], want[f, order[l, f, :c]])


def _rpn_pair(dtype, seed):
    from second_amd.models import RPNV2, RPNInference
    torch.manual_seed(seed)
    rpn = RPNV2().cuda().eval()
    g = torch.Generator(device="cpu").manual_seed(seed)
    for mod in rpn.modules():                         # statistics that leave a NON-ZERO background after every layer
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.empty(mod.num_features).uniform_(-0.3, 0.3, generator=g))
            mod.running_var.copy_(torch.empty(mod.num_features).uniform_(0.5, 1.5, generator=g))
            mod.bias.data.copy_(torch.empty(mod.num_features).uniform_(-0.2, 0.4, generator=g))
            mod.weight.data.copy_(torch.empty(mod.num_features).uniform_(0.5, 1.5, generator=g))
    return RPNInference(rpn, dtype)


def _bev(features, smap):
    """what RPNInference needs of models.SparseBEV: the rows and their site map"""
    from second_amd import models

    class BEV(models.SparseBEV):
        def __init__(self):
            self.features = features

        def site_map(self):
            return smap
    return BEV()                                      # tile_lists() is the base class's


@pytest.mark.parametrize("lazy", [False, True])           # lazy: live tiles written only, background read from the empty frame's maps
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("batch,h,w,n", [(2, 200, 176, 1500), (3, 50, 70, 200), (1, 40, 48, 1), (2, 33, 17, 0),
                                          (2, 96, 64, 4000),       # sites reach every tile: the kernels take the plain tile order
                                          (2, 120, 112, 700)])     # ... only in the later layers (lists first, plain order after)
def test_rpn_with_background_tiles_is_bit_identical_to_the_full_convs(ops, dtype, batch, h, w, n, lazy):
    rng = np.random.default_rng(n + h)
    idx = np.zeros((0, 4), np.int32)
    if n:
        idx = np.stack([rng.integers(0, batch, n), rng.integers(0, 2, n), rng.integers(0, h, n) % max(h // 2, 1), rng.integers(0, w, n)], 1)
        idx[0, 2:] = [0, 0]                                                # on the image border
        if n > 1:
            idx[1, 2:] = [h // 2 - 1, w - 1]
        idx = np.unique(idx, axis=0).astype(np.int32)
    feat = torch.randn(max(len(idx), 1), 64, device="cuda").to(dtype)
    smap = ops.sparse_site_map(torch.from_numpy(idx).cuda(), batch, [2, h, w])
    rpn = _rpn_pair(dtype, 3)
    assert rpn.background_convs == 6
    empty = rpn.empty_frame_maps(h, w)
    assert all(float(e.float().abs().max()) > 0 for e in empty), "the test needs a non-zero background"
    if h >= 24:
        assert not torch.equal(empty[3][0, :, 0, 0], empty[3][0, :, h // 2, w // 2]), "zero padding must leave its imprint along the border"
    bev = _bev(feat, smap)
    with torch.no_grad():
        rpn.skip_background, rpn.lazy_background = True, lazy
        ops.POISON_LAZY_OUTPUTS = lazy           # tiles a lazy conv does not write hold NaN: reading one would reach the output
        try:
            a = {k: v.clone() for k, v in rpn(bev).items()}
        finally:
            ops.POISON_LAZY_OUTPUTS = False
        counts = rpn.last_live_counts.sum(dim=1).cpu().numpy()
        rpn.skip_background = False
        b = rpn(bev)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    tiles = batch * ((h + 7) // 8) * ((w + 15) // 16)
    assert (counts <= tiles).all() and (np.diff(counts) >= 0).all()
    if h >= 40 and n <= 1:
        assert counts[1] < tiles, "nothing was skipped: the comparison would be vacuous"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("batch,h,w,n", [(3, 50, 70, 200), (1, 40, 48, 1), (2, 33, 17, 0), (2, 96, 64, 4000), (2, 120, 112, 700), (1, 8, 16, 1)])
def test_fused_tail_on_ragged_maps_matches_the_two_launches(ops, dtype, batch, h, w, n):
    """sec_conv2d_nhwc_tiles_tail on maps that are not multiples of the 8 x 16 tile, empty frames, one-site frames and scenes that take the
    plain tile order: the head tiles the last conv's list names equal, bit for bit, what sec_conv2d_nhwc_tiles_lazy +
    sec_conv1x1_chain_nhwc_tiles write (RPNInference.fused_tail off); every other tile is poisoned in both."""
    rng = np.random.default_rng(n + h + 1)
    idx = np.zeros((0, 4), np.int32)
    if n:
        idx = np.stack([rng.integers(0, batch, n), rng.integers(0, 2, n), rng.integers(0, h, n), rng.integers(0, w, n)], 1)
        idx[0, 2:] = [h - 1, w - 1]                                        # the ragged corner tile holds a site
        idx = np.unique(idx, axis=0).astype(np.int32)
    feat = torch.randn(max(len(idx), 1), 64, device="cuda").to(dtype)
    smap = ops.sparse_site_map(torch.from_numpy(idx).cuda(), batch, [2, h, w])
    rpn = _rpn_pair(dtype, 4)
    bev = _bev(feat, smap)
    seen, outs = [], {}
    with torch.no_grad():
        rpn.skip_background, rpn.lazy_background, rpn.lazy_heads = True, True, True
        for fused in (True, False):
            rpn.fused_tail = fused
            ops.set_op_hook(lambda name, fn, a, kw, res: seen.append((fused, name)))
            ops.POISON_LAZY_OUTPUTS = True
            try:
                outs[fused] = rpn(bev)
            finally:
                ops.POISON_LAZY_OUTPUTS = False
                ops.set_op_hook(None)
    assert (True, "conv2d_nhwc_tiles_tail") in seen and (False, "conv2d_nhwc_tiles_tail") not in seen and (False, "conv1x1_chain") in seen
    mask_f, mask_u = outs[True]["lazy_heads"][0], outs[False]["lazy_heads"][0]
    assert torch.equal(mask_f, mask_u)
    live = ((mask_u.int() >> 4) & 1).bool()
    th, tw = -(-h // 8), -(-w // 16)
    m = live.view(batch, th, tw).repeat_interleave(8, 1).repeat_interleave(16, 2)[:, :h, :w]
    if n:
        assert bool(m.any())
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        a, b = outs[True][k].permute(0, 2, 3, 1, 4)[m], outs[False][k].permute(0, 2, 3, 1, 4)[m]
        assert not torch.isnan(b.float()).any() and torch.equal(a, b), k


def test_lazy_background_across_a_change_from_lists_to_the_plain_order(ops):
    """A scene whose live share crosses the list threshold (225 / 256 of the tiles, csrc/dense.hip g_list_max_live_q8) between two layers: the early convs use their lists and write their live tiles
    only, a later conv takes the plain tile order and must read that output through the tile-indexed masks (every tile it computes, the
    true background tiles included, from the right source).  Unwritten tiles hold NaN."""
    dtype, batch, h, w = torch.bfloat16, 2, 120, 112
    ys, xs = np.meshgrid(np.arange(0, 103, 3), np.arange(0, w, 5), indexing="ij")     # sites down to y = 102: tile rows 0..12 of 15
    idx = np.concatenate([np.stack([np.full(ys.size, f), np.zeros(ys.size, np.int64), ys.ravel(), xs.ravel()], 1) for f in range(batch)]).astype(np.int32)
    feat = torch.randn(len(idx), 64, device="cuda").to(dtype)
    smap = ops.sparse_site_map(torch.from_numpy(idx).cuda(), batch, [2, h, w])
    rpn = _rpn_pair(dtype, 5)
    bev = _bev(feat, smap)
    with torch.no_grad():
        rpn.skip_background, rpn.lazy_background = True, True
        ops.POISON_LAZY_OUTPUTS = True
        try:
            a = {k: v.clone() for k, v in rpn(bev).items()}
        finally:
            ops.POISON_LAZY_OUTPUTS = False
        share = rpn.last_live_counts.sum(dim=1).cpu().numpy() / float(batch * 15 * 7)
        rpn.skip_background = False
        b = rpn(bev)
    assert share[0] <= 225 / 256 < share[-1], f"the scene must change from lists to the plain order between layers: {share}"
    for k in a:
        assert not torch.isnan(a[k].float()).any(), k
        assert torch.equal(a[k], b[k]), k


def test_detector_with_and_without_background_tiles_gives_identical_detections():
    from second_amd.models import SecondDetector, CAR_FHD
    from second_amd import synthetic as syn
    clouds = [syn.syn_kitti_cloud(s) for s in range(2)]
    pts, offs = syn.batch_clouds(clouds)
    pts, offs = torch.from_numpy(pts).cuda(), torch.from_numpy(offs).cuda()
    torch.manual_seed(0)
    det = SecondDetector(CAR_FHD).cuda().eval()
    g = torch.Generator().manual_seed(1)
    for m in det.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.copy_(torch.empty(m.num_features).uniform_(-0.1, 0.1, generator=g))
            m.running_var.copy_(torch.empty(m.num_features).uniform_(0.5, 1.5, generator=g))
            m.bias.data.copy_(torch.empty(m.num_features).uniform_(-0.2, 0.3, generator=g))
    det.prepare_inference(torch.bfloat16)
    from second_amd import ops
    outs = []
    with torch.no_grad():
        for skip, lazy in ((True, False), (True, True), (False, False)):
            det.rpn.skip_background, det.rpn.lazy_background = skip, lazy
            ops.POISON_LAZY_OUTPUTS = lazy
            try:
                o = det.forward_points(pts, offs, static=True)
            finally:
                ops.POISON_LAZY_OUTPUTS = False
            outs.append({k: v.clone() for k, v in o.items() if isinstance(v, torch.Tensor)})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[2][k]), k
        assert torch.equal(outs[1][k], outs[2][k]), k


def test_lazy_heads_give_the_detections_of_the_materialised_head_tensor(monkeypatch):
    """The fused 1x1 tail writes the head tensor's live tiles only (background == NULL) and select / decode read every other tile
    from the empty frame's head map (sec_predict_select_lazy / _decode_lazy).  A network whose EMPTY regions score above the
    threshold (random BatchNorm shifts, default heads: thousands of background candidates), unwritten tiles poisoned with NaN:
    every output of the step must equal, bit for bit, the step that materialises the whole tensor (SEC_RPN_LAZY_HEADS=0)."""
    from second_amd.models import SecondDetector, CAR_FHD
    from second_amd import ops, synthetic as syn
    clouds = [syn.syn_kitti_cloud(s) for s in range(3)]
    pts, offs = syn.batch_clouds(clouds)
    pts, offs = torch.from_numpy(pts).cuda(), torch.from_numpy(offs).cuda()
    torch.manual_seed(0)
    det = SecondDetector(dict(CAR_FHD, nms_score_threshold=0.3)).cuda().eval()
    g = torch.Generator().manual_seed(2)
    for m in det.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.copy_(torch.empty(m.num_features).uniform_(-0.1, 0.1, generator=g))
            m.running_var.copy_(torch.empty(m.num_features).uniform_(0.5, 1.5, generator=g))
            m.bias.data.copy_(torch.empty(m.num_features).uniform_(-0.2, 0.3, generator=g))
    det.rpn.conv_cls.bias.data.fill_(-0.2)                     # sigmoid(-0.2 + small) straddles 0.3 ... 0.5: background anchors are candidates
    det.prepare_inference(torch.bfloat16)
    det.calibrate(pts, offs)
    with torch.no_grad():
        with det.lazy_heads():                                   # the lazy form is really taken, and the head tensor really has holes
            ops.POISON_LAZY_OUTPUTS = True
            try:
                vox = det.voxel_generator.generate_device(pts, offs, mean_features=4, sync=False, mean_dtype=torch.bfloat16)
                preds = det.network_forward(vox["mean"], vox["coordinates"], 3, num_active_dev=vox["voxel_offsets"][3:])
            finally:
                ops.POISON_LAZY_OUTPUTS = False
        assert "lazy_heads" in preds and torch.isnan(preds["cls_preds"].float()).any()
        written = ((preds["lazy_heads"][0].int() >> 4) & 1).sum().item()
        assert 0 < written < preds["lazy_heads"][0].numel()
        ops.POISON_LAZY_OUTPUTS = True
        try:
            lazy = {k: v.clone() for k, v in det.forward_points(pts, offs, static=True).items() if isinstance(v, torch.Tensor)}
        finally:
            ops.POISON_LAZY_OUTPUTS = False
        monkeypatch.setenv("SEC_RPN_LAZY_HEADS", "0")
        eager = {k: v.clone() for k, v in det.forward_points(pts, offs, static=True).items() if isinstance(v, torch.Tensor)}
        plain = det.network_forward(vox["mean"], vox["coordinates"], 3, num_active_dev=vox["voxel_offsets"][3:])
        assert "lazy_heads" not in plain and not torch.isnan(plain["cls_preds"].float()).any()
    assert int(eager["valid"].sum()) >= 20
    bg_share = 1.0 - written / preds["lazy_heads"][0].numel()
    assert bg_share > 0.3
    for k in eager:
        assert torch.equal(lazy[k], eager[k]), k



@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("list_pct", ["", "0"])
def test_last_conv_with_the_tail_in_its_epilogue_is_bit_identical_to_the_two_launches(monkeypatch, dtype, list_pct):
    """sec_conv2d_nhwc_tiles_tail (the last 3x3 conv with deblock + heads in its epilogue: its output tile goes from LDS into the two 1x1
    GEMMs) against sec_conv2d_nhwc_tiles_lazy + sec_conv1x1_chain_nhwc_tiles (SEC_RPN_FUSED_TAIL=0): the head tensor's live tiles and
    every output of the step, bit for bit; unwritten tiles poisoned.  list_pct "0": the conv falls back to the plain tile order
    (every tile computed) while the consumers still follow the lists."""
    import subprocess, sys, os
    if list_pct:          # the list threshold is read once per process: its own interpreter
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, SEC_RPN_LIST_MAX_LIVE=list_pct, SEC_TAIL_TEST_DTYPE=str(dtype).split(".")[-1],
                   PYTHONPATH=os.pathsep.join([root, os.path.join(root, "second.pytorch_amd"), os.environ.get("PYTHONPATH", "")]))
        r = subprocess.run([sys.executable, __file__, "--tail-child"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return
    _tail_vs_two_launches(dtype, monkeypatch.setenv)


def _tail_vs_two_launches(dtype, setenv):
    from second_amd.models import SecondDetector, CAR_FHD
    from second_amd import ops, synthetic as syn
    clouds = [syn.syn_kitti_cloud(s) for s in range(3)]
    pts, offs = syn.batch_clouds(clouds)
    pts, offs = torch.from_numpy(pts).cuda(), torch.from_numpy(offs).cuda()
    torch.manual_seed(0)
    det = SecondDetector(dict(CAR_FHD, nms_score_threshold=0.3)).cuda().eval()
    g = torch.Generator().manual_seed(5)
    for m in det.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.copy_(torch.empty(m.num_features).uniform_(-0.1, 0.1, generator=g))
            m.running_var.copy_(torch.empty(m.num_features).uniform_(0.5, 1.5, generator=g))
            m.bias.data.copy_(torch.empty(m.num_features).uniform_(-0.2, 0.3, generator=g))
    det.rpn.conv_cls.bias.data.fill_(-0.2)
    det.prepare_inference(dtype)
    det.calibrate(pts, offs)
    seen = []
    outs, heads = {}, {}
    with torch.no_grad():
        vox = det.voxel_generator.generate_device(pts, offs, mean_features=4, sync=False, mean_dtype=dtype)
        for mode in ("1", "0"):
            setenv("SEC_RPN_FUSED_TAIL", mode)
            ops.set_op_hook(lambda name, fn, a, kw, res: seen.append((mode, name)))
            ops.POISON_LAZY_OUTPUTS = True
            try:
                with det.lazy_heads():
                    preds = det.network_forward(vox["mean"], vox["coordinates"], 3, num_active_dev=vox["voxel_offsets"][3:])
                outs[mode] = {k: v.clone() for k, v in det.forward_points(pts, offs, static=True).items() if isinstance(v, torch.Tensor)}
            finally:
                ops.POISON_LAZY_OUTPUTS = False
                ops.set_op_hook(None)
            heads[mode] = preds
    assert ("1", "conv2d_nhwc_tiles_tail") in seen and ("1", "conv1x1_chain") not in seen
    assert ("0", "conv1x1_chain") in seen and ("0", "conv2d_nhwc_tiles_tail") not in seen
    live = ((heads["0"]["lazy_heads"][0].int() >> 4) & 1).bool()                 # [B, tiles]: the last conv's list holds the tile
    assert torch.equal(heads["1"]["lazy_heads"][0], heads["0"]["lazy_heads"][0]) and 0 < int(live.sum()) < live.numel()
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        a, b = heads["1"][k], heads["0"][k]
        assert a.shape == b.shape
        # the tiles of the list, bit for bit (the fused conv may have written more: plain order); [B, A, H, W, code] views of the head map
        bsz, hh, ww = a.shape[0], a.shape[2], a.shape[3]
        th, tw = -(-hh // 8), -(-ww // 16)
        m = live.view(bsz, th, tw).repeat_interleave(8, 1).repeat_interleave(16, 2)[:, :hh, :ww]
        av, bv = a.permute(0, 2, 3, 1, 4)[m], b.permute(0, 2, 3, 1, 4)[m]
        assert not torch.isnan(bv.float()).any() and torch.equal(av, bv), k
    assert int(outs["0"]["valid"].sum()) >= 20
    for k in outs["0"]:
        assert torch.equal(outs["1"][k], outs["0"][k]), k

@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_prepared_rpn_follows_a_state_dict_loaded_later(dtype):
    """RPNInference keeps packed copies of its folded weights (MFMA slab order, gather permutation, hi | lo pairs of the fp32
    form) and the empty frame's activations of the background tiles: load_state_dict into a PREPARED network re-packs them in
    place -- the next forward is the other network's, bit for bit, including its background tiles."""
    from second_amd.models import RPNV2, RPNInference, SparseBEV
    import spconv

    def make(seed):
        torch.manual_seed(seed)
        rpn = RPNV2().eval()
        g = torch.Generator().manual_seed(seed)
        for m in rpn.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.empty_like(m.running_mean).uniform_(-0.3, 0.1, generator=g))
                m.running_var.copy_(torch.empty_like(m.running_var).uniform_(0.5, 1.5, generator=g))
                m.bias.data.uniform_(-0.1, 0.3, generator=g)               # a non-zero background
        return RPNInference(rpn.cuda(), dtype)
    a, b = make(1), make(2)
    g = torch.Generator().manual_seed(5)
    n = 600
    idx = torch.stack([torch.randint(0, 2, (n,), generator=g), torch.randint(0, 2, (n,), generator=g),
                       torch.randint(20, 90, (n,), generator=g), torch.randint(30, 120, (n,), generator=g)], 1).int()
    idx = torch.unique(idx, dim=0).cuda()
    feats = torch.randn(idx.shape[0], 64, generator=g).abs().cuda().to(torch.bfloat16 if dtype != torch.float32 else torch.float32)
    sp = spconv.SparseConvTensor(feats, idx, [2, 200, 176], 2)

    def run(net):
        with torch.no_grad():
            x = SparseBEV(sp) if net.gather_packed is not None else sp.dense_channels_last_2d()
            return {k: v.float().clone() for k, v in net(x).items()}
    out_a, out_b = run(a), run(b)
    assert not torch.equal(out_a["cls_preds"], out_b["cls_preds"])
    a.load_state_dict(b.state_dict())
    got = run(a)
    for k in out_b:
        assert torch.equal(got[k], out_b[k]), k


if __name__ == "__main__" and "--tail-child" in __import__("sys").argv:
    import os as _os
    _tail_vs_two_launches(getattr(torch, _os.environ["SEC_TAIL_TEST_DTYPE"]), lambda k, v: _os.environ.__setitem__(k, v))
    print("ok")
