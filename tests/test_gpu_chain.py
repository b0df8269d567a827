"""-m gpu: the fused rulebook chain (sec_rulebook_chain_sorted: every SubM and strided rulebook of a SpMiddleFHD-type stack in
4 + (levels - 1) launches) against the CPU oracle, element for element, and against the layer-by-layer sorted builds it replaces
(second/pytorch/models/middle.py:146-189: one spconv.ops.get_indice_pairs call per layer / indice_key)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402  (test infrastructure only)
from test_gpu_parity import dev, _random_indices, _tables_from_pairs  # noqa: E402

MIDDLE = [(3, 2, 1), (3, 2, 1), (3, 2, (0, 1, 1)), ((3, 1, 1), (2, 1, 1), 0)]      # the four strided layers of SpMiddleFHD


@pytest.fixture(scope="module")
def ops():
    from second_amd import ops
    return ops


def _oracle_levels(idx, batch, shape, layers):
    """[(indices, shape, conv nbr_out (level >= 1), subm nbr_out)] per level, from the oracle's sorted-numbering rulebooks."""
    out = []
    cur, cur_shape = idx, list(shape)
    conv_tbl = None
    for li in range(len(layers) + 1):
        _, sp, spn = orc.rulebook_subm(cur, batch, cur_shape, 3)
        sub_tbl, _ = _tables_from_pairs(sp, spn, len(cur), len(cur))
        out.append((cur, cur_shape, conv_tbl, sub_tbl))
        if li == len(layers):
            break
        ks, st, pd = layers[li]
        nxt, pairs, pair_num, out_shape = orc.rulebook_conv_sorted(cur, batch, cur_shape, ks, st, pd, 1)
        conv_tbl, _ = _tables_from_pairs(pairs, pair_num, len(cur), len(nxt))
        cur, cur_shape = nxt, [int(v) for v in out_shape]
    return out


def _check(r, want, caps):
    for l, (idx, shape, conv_tbl, sub_tbl) in enumerate(want):
        L = r["levels"][l]
        assert L["shape"] == shape
        m = len(idx)
        if l >= 1:
            cap = caps[l - 1]
            assert L["num_dev"].tolist() == [min(m, cap), m], (l, L["num_dev"].tolist(), m, cap)
            live = min(m, cap)
            np.testing.assert_array_equal(L["indices"][:live].cpu().numpy(), idx[:live], err_msg=f"level {l} out_indices")
            got = L["nbr_out"][:live].cpu().numpy()
            ref = conv_tbl[:live].copy()
            prev_cap = caps[l - 2] if l >= 2 else None
            if prev_cap is not None:
                ref[ref >= prev_cap] = -1          # inputs past the capacity of the level below are not rows
            np.testing.assert_array_equal(got, ref, err_msg=f"level {l} conv table")
        else:
            live = m
        if L["subm_nbr"] is not None:
            ref = sub_tbl[:live].copy()
            ref[ref >= live] = -1
            np.testing.assert_array_equal(L["subm_nbr"][:live].cpu().numpy(), ref, err_msg=f"level {l} SubM table")


@pytest.mark.parametrize("shape", [(41, 64, 96), (11, 24, 19), (9, 16, 64), (21, 33, 96)])
def test_chain_on_random_sites_vs_oracle(ops, shape):
    """Levels 1 .. L (level-0 SubM needs the voxeliser's table: next test); grids whose rows are and are not multiples of 32 bits,
    garbage rows behind the live count, every level against the oracle."""
    rng = np.random.default_rng(11)
    batch = 3
    idx = _random_indices(rng, batch, shape, 700)
    layers = []
    cur = list(shape)
    for ks, st, pd in MIDDLE:
        nxt = orc.conv_output_size(cur, [ks] * 3 if isinstance(ks, int) else list(ks), [st] * 3 if isinstance(st, int) else list(st),
                                   [pd] * 3 if isinstance(pd, int) else list(pd), [1, 1, 1])
        if min(int(v) for v in nxt) < 1 or min(cur) < 3:
            break
        layers.append((ks, st, pd))
        cur = [int(v) for v in nxt]
    want = _oracle_levels(idx, batch, shape, layers)
    caps = [len(w[0]) + 37 for w in want[1:]]
    padded = np.concatenate([idx, np.full((50, 4), 3, np.int32)])
    n_dev = dev(np.array([len(idx)], np.int32))
    r = ops.rulebook_chain(dev(padded), batch, shape, [(ks, st, pd, c) for (ks, st, pd), c in zip(layers, caps)], n_dev=n_dev,
                           want_subm=[False] + [True] * len(layers), want_site_map=True)
    assert r is not None
    _check(r, want, caps)
    # the BEV site map of the last level == the generic fill + scatter map of its rows
    last = r["levels"][-1]
    ref = ops.sparse_site_map(last["indices"], batch, last["shape"], num_dev=last["num_dev"])
    assert torch.equal(r["site_map"], ref)


def test_chain_capacity_overflow_is_reported_and_contained(ops):
    rng = np.random.default_rng(3)
    batch, shape = 2, (21, 40, 64)
    idx = _random_indices(rng, batch, shape, 900)
    layers = MIDDLE[:3]
    want = _oracle_levels(idx, batch, shape, layers)
    caps = [len(want[1][0]) - 100, len(want[2][0]) + 10, len(want[3][0]) - 5]
    guard = 64
    r = ops.rulebook_chain(dev(idx), batch, shape, [(ks, st, pd, c) for (ks, st, pd), c in zip(layers, caps)],
                           want_subm=[False, True, True, True])
    _check(r, want, caps)


def test_chain_from_the_voxeliser_matches_oracle_and_layerwise(ops):
    """The detector's own call: clouds -> sec_voxelize_f32 (static) -> chain with the level-0 SubM table through the voxeliser's
    hash table; every table vs the oracle and vs the layer-by-layer sorted builds."""
    from second_amd import synthetic as syn
    clouds = [syn.syn_kitti_cloud(s, num_points=5000 + 700 * s, num_voxels=4300 + 500 * s) for s in range(3)]
    pts, offs = syn.batch_clouds(clouds)
    vox = ops.voxelize(dev(pts), dev(offs), syn.CAR_FHD_RANGE, syn.CAR_FHD_VOXEL, 5, 40000, sync=False, mean_features=4)
    n = int(vox["voxel_offsets"][-1].item())
    idx = vox["coordinates"][:n].cpu().numpy()
    shape = [41, 1600, 1408]
    want = _oracle_levels(idx, 3, shape, MIDDLE)
    caps = [-(-int(len(w[0]) * 1.25) // 256) * 256 for w in want[1:]]
    r = ops.rulebook_chain(vox["coordinates"], 3, shape, [(ks, st, pd, c) for (ks, st, pd), c in zip(MIDDLE, caps)],
                           n_dev=vox["voxel_offsets"][3:], site_table=vox["site_table"], want_subm=[True, True, True, True, False],
                           want_site_map=True)
    assert r is not None and r["levels"][-1]["shape"] == [2, 200, 176]
    _check(r, want, caps)
    # layer by layer (the path this replaces)
    prev = ops.set_rulebook_numbering("sorted")
    try:
        cur, cur_shape, nd, sites = vox["coordinates"], shape, vox["voxel_offsets"][3:], None
        for l, ((ks, st, pd), cap) in enumerate(zip(MIDDLE, caps)):
            rb = ops.rulebook_conv(cur, 3, cur_shape, ks, st, pd, 1, n_dev=nd, out_cap=cap, want_nbr_in=False, in_sites=sites)
            m = int(rb["num_out_dev"][0].item())
            L = r["levels"][l + 1]
            assert torch.equal(rb["num_out_dev"], L["num_dev"])
            assert torch.equal(rb["out_indices"][:m], L["indices"][:m]) and torch.equal(rb["nbr_out"][:m], L["nbr_out"][:m])
            cur, cur_shape, nd, sites = rb["out_indices"], rb["out_shape"], rb["num_out_dev"], rb["site_table"]
    finally:
        ops.set_rulebook_numbering(prev)


def test_detector_with_and_without_the_fused_chain_agree():
    """SecondDetector static forward: fused chain (default) vs layer-by-layer rulebooks -- bit-identical detections."""
    from second_amd import synthetic as syn
    from second_amd.models import SecondDetector, CAR_FHD
    torch.manual_seed(0)
    det = SecondDetector(CAR_FHD)
    syn.randomise_like_trained(det, seed=1)
    det = det.eval().cuda()
    det.prepare_inference(torch.bfloat16)
    clouds = [syn.syn_kitti_cloud(s, num_points=9000, num_voxels=8000) for s in range(2)]
    pts, offs = syn.batch_clouds(clouds)
    pts, offs = dev(pts), dev(offs)
    with torch.no_grad():
        det.calibrate(pts, offs)
        mfe = det.middle_feature_extractor
        assert mfe.fused_chain
        a = det.forward_points(pts, offs, static=True)
        det.check_overflow()
        mfe.fused_chain = False
        b = det.forward_points(pts, offs, static=True)
        mfe.fused_chain = True
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert int(a["valid"].sum()) > 0
