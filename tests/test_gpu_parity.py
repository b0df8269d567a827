"""-m gpu parity tests: HIP path (through the C ABI) vs the CPU oracle and the golden fixtures.
Bit-exact for voxel indices, rulebooks and NMS keep lists; <= 1e-4 relative for features (BASELINE.json)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402  (test infrastructure only)


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


@pytest.fixture(scope="module")
def ops():
    from second_amd import ops
    return ops


@pytest.fixture(scope="module")
def syn():
    from second_amd import synthetic
    return synthetic


# ------------------------------------------------------------------ voxelisation
def _check_voxelize(ops, clouds, vs, rng_, max_points, max_voxels, cap_mode):
    from second_amd.synthetic import batch_clouds
    pts, offs = batch_clouds(clouds)
    res = ops.voxelize(dev(pts), dev(offs), rng_, vs, max_points, max_voxels, cap_mode, mean_features=pts.shape[1])
    voff = res["voxel_offsets"].cpu().numpy()
    for b, c in enumerate(clouds):
        ref = orc.points_to_voxel(c, vs, rng_, max_points, max_voxels, cap_mode)
        lo, hi = voff[b], voff[b + 1]
        assert hi - lo == ref["voxel_num"], (b, hi - lo, ref["voxel_num"])
        co = res["coordinates"][lo:hi].cpu().numpy()
        assert (co[:, 0] == b).all()
        np.testing.assert_array_equal(co[:, 1:], ref["coordinates"])
        np.testing.assert_array_equal(res["num_points_per_voxel"][lo:hi].cpu().numpy(), ref["num_points_per_voxel"])
        np.testing.assert_array_equal(res["voxels"][lo:hi].cpu().numpy(), ref["voxels"])
        if ref["voxel_num"]:
            mean = orc.simple_voxel_mean(ref["voxels"], ref["num_points_per_voxel"], c.shape[1])
            np.testing.assert_allclose(res["mean"][lo:hi].cpu().numpy(), mean, rtol=1e-6, atol=1e-7)
    return res


@pytest.mark.parametrize("tag", ["kitti", "coarse_cap"])
@pytest.mark.parametrize("cap_mode", ["break", "continue"])
def test_voxelize_golden_clouds(ops, golden, tag, cap_mode):
    g = golden("voxel_coords")
    pts, vs, rng_, cap = g[f"{tag}_points"], g[f"{tag}_voxel_size"], g[f"{tag}_range"], int(g[f"{tag}_cap"])
    res = _check_voxelize(ops, [pts], vs, rng_, 5, cap, cap_mode)
    if cap_mode == "break":  # fixture produced by the reference's own loop (simplevis.py:8-60)
        np.testing.assert_array_equal(res["coordinates"][:, 1:].cpu().numpy(), g[f"{tag}_coors"])


def test_voxelize_batched_ragged_and_edge_cases(ops, golden, syn):
    g = golden("voxel_coords")
    vs, rng_ = g["coarse_cap_voxel_size"], g["coarse_cap_range"]
    p = g["coarse_cap_points"]
    empty = np.zeros((0, 4), np.float32)
    outside = p[:50] + 1000.0
    one = p[5:6]
    nan = p[:20].copy()
    nan[3, 0] = np.nan
    nan[4, 1] = np.inf
    clouds = [p[:700], empty, outside, one, p[700:], nan]
    for cap_mode in ("break", "continue"):
        _check_voxelize(ops, clouds, vs, rng_, 3, 120, cap_mode)   # cap hit in some clouds
        _check_voxelize(ops, clouds, vs, rng_, 1, 100000, cap_mode)  # max_points 1, cap never hit
    # many points in few voxels: the atomicMin cascade must keep the first max_points arrivals in order
    rs = np.random.default_rng(0)
    dense = np.concatenate([rs.uniform([0, -8, -3], [1.2, -6.8, -2], (5000, 3)), rs.uniform(0, 1, (5000, 1))], 1)
    _check_voxelize(ops, [dense.astype(np.float32)], vs, rng_, 60, 1000, "break")


def test_voxelize_pillars_sorted_slot_path(ops, syn):
    """max_points > 8 takes the run path (csrc/voxelize.hip: k_vox_group_rank -> k_vox_count_scan -> k_vox_run_scatter ->
    k_vox_run_select: every voxel's points gathered into a run, its max_points smallest indices put in order by one wave) instead of
    the atomicMin cascade: nuScenes-size pillar clouds (0.25 m pillars, 60 points each, hundreds of points in the pillars near the
    sensor), a ragged batch with an empty cloud, the voxel cap hit and not hit, both cap modes, and the 100 / 200-point forms of the
    select kernel (KITTI pointpillars configs keep 100 points per pillar) -- voxel order, slot order, counts and contents bit-exact
    against the sequential oracle loop (pointpillars: all.pp.largea.config:6-15)."""
    rng_ = [-50, -50, -10, 50, 50, 10]
    vs = [0.25, 0.25, 20]
    clouds = [syn.syn_nusc_cloud(0, 120000, tuple(rng_), scene="urban"), np.zeros((0, 4), np.float32),
              syn.syn_nusc_cloud(1, 40000, tuple(rng_), scene="urban")]
    for cap_mode in ("break", "continue"):
        for max_points, max_voxels in ((100, 30000), (200, 8000), (60, 30000), (60, 4000), (9, 30000)):
            res = _check_voxelize(ops, clouds, vs, rng_, max_points, max_voxels, cap_mode)
    assert int(res["num_points_per_voxel"].max()) == 9 and res["voxel_num"] > 15000


def test_voxelize_car_fhd_batch8(ops, syn):
    clouds = [syn.syn_kitti_cloud(s) for s in range(8)]
    res = _check_voxelize(ops, clouds, syn.CAR_FHD_VOXEL, syn.CAR_FHD_RANGE, 5, 40000, "break")
    assert res["voxel_num"] == 8 * 16000
    # size-independent properties: one voxel per distinct cell, every stored point lies in its voxel
    co = res["coordinates"].long()
    lin = ((co[:, 0] * 41 + co[:, 1]) * 1600 + co[:, 2]) * 1408 + co[:, 3]
    assert torch.unique(lin).numel() == lin.numel()
    v = res["voxels"][:, 0, :3]
    lo = torch.tensor(syn.CAR_FHD_RANGE[:3], device="cuda")
    cell = torch.floor((v - lo) / torch.tensor(syn.CAR_FHD_VOXEL, device="cuda")).long()
    assert (cell[:, 0] == co[:, 3]).all() and (cell[:, 1] == co[:, 2]).all() and (cell[:, 2] == co[:, 1]).all()


# ------------------------------------------------------------------ rulebooks
def _random_indices(rng, batch, shape, n):
    idx = []
    for b in range(batch):
        lin = rng.choice(int(np.prod(shape)), size=n, replace=False)
        z, y, x = np.unravel_index(lin, shape)
        idx.append(np.stack([np.full_like(z, b), z, y, x], 1))
    idx = np.concatenate(idx).astype(np.int32)
    rng.shuffle(idx)
    return idx


def _tables_from_pairs(pairs, pair_num, n_in, n_out):
    K = pairs.shape[0]
    nbr_out = -np.ones((n_out, K), np.int32)
    nbr_in = -np.ones((n_in, K), np.int32)
    for k in range(K):
        i, o = pairs[k, 0, :pair_num[k]], pairs[k, 1, :pair_num[k]]
        nbr_out[o, k] = i
        nbr_in[i, k] = o
    return nbr_out, nbr_in


@pytest.mark.parametrize("ksize", [3, (3, 1, 1), (1, 3, 3), 5])
def test_rulebook_subm_bit_exact(ops, ksize):
    rng = np.random.default_rng(0)
    shape = (9, 20, 17)
    idx = _random_indices(rng, 3, shape, 400)
    rb = ops.rulebook_subm(dev(idx), 3, shape, ksize, 1, want_pairs=True)
    _, pairs, pair_num = orc.rulebook_subm(idx, 3, shape, ksize)
    np.testing.assert_array_equal(rb["pair_num"].cpu().numpy(), pair_num)
    np.testing.assert_array_equal(rb["pairs"].cpu().numpy(), pairs)
    nbr_out, _ = _tables_from_pairs(pairs, pair_num, len(idx), len(idx))
    np.testing.assert_array_equal(rb["nbr_out"].cpu().numpy(), nbr_out)


@pytest.mark.parametrize("ksize,stride,padding,dilation", [
    (3, 2, 1, 1), (3, 2, (0, 1, 1), 1), ((3, 1, 1), (2, 1, 1), 0, 1), (3, 1, 0, 1), (2, 2, 0, 1), (3, 1, 1, 1), (3, 3, 1, 1),
    (3, 1, 2, 2), (3, (2, 1, 1), 1, (1, 2, 2)), (5, 2, 2, 1), (4, 2, 1, 1)])
def test_rulebook_conv_bit_exact(ops, ksize, stride, padding, dilation):
    rng = np.random.default_rng(1)
    shape = (11, 24, 19)
    idx = _random_indices(rng, 2, shape, 500)
    rb = ops.rulebook_conv(dev(idx), 2, shape, ksize, stride, padding, dilation, want_pairs=True)
    out_idx, pairs, pair_num, out_shape = orc.rulebook_conv(idx, 2, shape, ksize, stride, padding, dilation)
    lean = ops.rulebook_conv(dev(idx), 2, shape, ksize, stride, padding, dilation, want_nbr_in=False)   # inference form
    assert lean["nbr_in"] is None and torch.equal(lean["nbr_out"], rb["nbr_out"])
    assert torch.equal(lean["out_indices"], rb["out_indices"])
    # SubM rulebook on the outputs, reusing the strided build's hash table == the stand-alone build
    sub_reuse = ops.rulebook_subm(lean["out_indices"], 2, lean["out_shape"], 3, 1, site_table=lean["site_table"])
    sub_plain = ops.rulebook_subm(lean["out_indices"].clone(), 2, lean["out_shape"], 3, 1)
    assert torch.equal(sub_reuse["nbr_out"], sub_plain["nbr_out"])
    assert rb["out_shape"] == out_shape.tolist()
    assert rb["num_out"] == len(out_idx)
    np.testing.assert_array_equal(rb["out_indices"].cpu().numpy(), out_idx)
    np.testing.assert_array_equal(rb["pair_num"].cpu().numpy(), pair_num)
    np.testing.assert_array_equal(rb["pairs"].cpu().numpy(), pairs)
    nbr_out, nbr_in = _tables_from_pairs(pairs, pair_num, len(idx), len(out_idx))
    np.testing.assert_array_equal(rb["nbr_out"].cpu().numpy(), nbr_out)
    np.testing.assert_array_equal(rb["nbr_in"].cpu().numpy(), nbr_in)


@pytest.mark.parametrize("ksize,stride,padding,dilation", [
    (3, 2, 1, 1), (3, 2, (0, 1, 1), 1), ((3, 1, 1), (2, 1, 1), 0, 1), (3, 1, 0, 1), (2, 2, 0, 1), (3, 1, 1, 1), (3, 3, 1, 1),
    (3, 1, 2, 2), (3, (2, 1, 1), 1, (1, 2, 2)), (5, 2, 2, 1), (4, 2, 1, 1)])
def test_rulebook_conv_sorted_numbering_bit_exact(ops, ksize, stride, padding, dilation):
    """numbering="sorted" (spconv's GPU path: outputs by ascending linear cell index) -- bitmap + rank build,
    no hash table -- against the oracle's restatement, same 11 geometries as the first-touch form."""
    rng = np.random.default_rng(1)
    shape = (11, 24, 19)
    idx = _random_indices(rng, 2, shape, 500)
    rb = ops.rulebook_conv(dev(idx), 2, shape, ksize, stride, padding, dilation, want_pairs=True, numbering="sorted")
    out_idx, pairs, pair_num, out_shape = orc.rulebook_conv_sorted(idx, 2, shape, ksize, stride, padding, dilation)
    assert rb["out_shape"] == out_shape.tolist() and rb["num_out"] == len(out_idx)
    np.testing.assert_array_equal(rb["out_indices"].cpu().numpy(), out_idx)
    lin = ((out_idx[:, 0].astype(np.int64) * out_shape[0] + out_idx[:, 1]) * out_shape[1] + out_idx[:, 2]) * out_shape[2] + out_idx[:, 3]
    assert np.all(np.diff(lin) > 0)
    np.testing.assert_array_equal(rb["pair_num"].cpu().numpy(), pair_num)
    np.testing.assert_array_equal(rb["pairs"].cpu().numpy(), pairs)
    nbr_out, nbr_in = _tables_from_pairs(pairs, pair_num, len(idx), len(out_idx))
    np.testing.assert_array_equal(rb["nbr_out"].cpu().numpy(), nbr_out)
    np.testing.assert_array_equal(rb["nbr_in"].cpu().numpy(), nbr_in)
    # the same convolution as the first-touch rulebook: identical (offset, input coordinate, output coordinate) triples
    ft = ops.rulebook_conv(dev(idx), 2, shape, ksize, stride, padding, dilation, want_pairs=True)
    def triples(r):
        oi, pr, pn = r["out_indices"].cpu().numpy(), r["pairs"].cpu().numpy(), r["pair_num"].cpu().numpy()
        return {(k, tuple(idx[pr[k, 0, i]]), tuple(oi[pr[k, 1, i]])) for k in range(len(pn)) for i in range(pn[k])}
    assert triples(rb) == triples(ft)
    # inference form + SubM layer on the outputs through the bitmap ranks == stand-alone SubM build
    lean = ops.rulebook_conv(dev(idx), 2, shape, ksize, stride, padding, dilation, want_nbr_in=False, numbering="sorted")
    assert lean["nbr_in"] is None and torch.equal(lean["nbr_out"], rb["nbr_out"])
    sub_reuse = ops.rulebook_subm(lean["out_indices"], 2, lean["out_shape"], 3, 1, site_table=lean["site_table"])
    sub_plain = ops.rulebook_subm(lean["out_indices"].clone(), 2, lean["out_shape"], 3, 1)
    assert torch.equal(sub_reuse["nbr_out"], sub_plain["nbr_out"])


@pytest.mark.parametrize("shape", [(11, 24, 19), (9, 16, 64), (5, 33, 96), (41, 40, 32)])
def test_rulebook_conv_sorted_chain_from_input_bitmap(ops, shape):
    """A chain of sorted-numbering strided layers (the SpMiddleFHD pattern: 3x3x3 s2 p1, 3x3x3 s2 p(0,1,1), (3,1,1) s(2,1,1)): from
    the second layer on the output bitmap is derived from the previous layer's bitmap (no atomics).  Row widths that are and are
    not multiples of 32 (words straddling row ends), each layer against the oracle."""
    rng = np.random.default_rng(7)
    batch = 3
    idx = _random_indices(rng, batch, shape, 900)
    layers = [(3, 2, 1), (3, 2, (0, 1, 1)), ((3, 1, 1), (2, 1, 1), 0)]
    cur, cur_shape, sites = idx, list(shape), None
    for li, (ks, st, pd) in enumerate(layers):
        if min(cur_shape) < 3:
            break
        want_idx, pairs, pair_num, out_shape = orc.rulebook_conv_sorted(cur, batch, cur_shape, ks, st, pd, 1)
        rb = ops.rulebook_conv(dev(cur), batch, cur_shape, ks, st, pd, 1, want_pairs=True, numbering="sorted", in_sites=sites)
        assert rb["num_out"] == len(want_idx), (li, rb["num_out"], len(want_idx))
        np.testing.assert_array_equal(rb["out_indices"].cpu().numpy(), want_idx, err_msg=f"layer {li}")
        np.testing.assert_array_equal(rb["pairs"].cpu().numpy(), pairs)
        # the atomics form of the same layer
        plain = ops.rulebook_conv(dev(cur), batch, cur_shape, ks, st, pd, 1, numbering="sorted")
        assert torch.equal(plain["out_indices"], rb["out_indices"]) and torch.equal(plain["nbr_out"], rb["nbr_out"])
        cur, cur_shape, sites = want_idx, [int(v) for v in out_shape], rb["site_table"]


def test_rulebook_conv_sorted_static_capacity_and_overflow(ops):
    """static-capacity form (device row counts, caller-owned tables) of the sorted numbering, with padding rows and with a
    capacity smaller than the output count (overflow reported in num_out[1], nothing written past the tables)."""
    rng = np.random.default_rng(5)
    shape = (11, 24, 19)
    idx = _random_indices(rng, 2, shape, 400)
    want_idx, pairs, pair_num, out_shape = orc.rulebook_conv_sorted(idx, 2, shape, 3, 2, 1, 1)
    m = len(want_idx)
    padded = np.concatenate([idx, np.full((100, 4), 7, np.int32)])           # rows beyond n_dev are garbage
    n_dev = dev(np.array([len(idx)], np.int32))
    rb = ops.rulebook_conv(dev(padded), 2, shape, 3, 2, 1, 1, n_dev=n_dev, out_cap=m + 50, numbering="sorted")
    cnt = rb["num_out_dev"].cpu().numpy()
    assert cnt.tolist() == [m, m]
    np.testing.assert_array_equal(rb["out_indices"].cpu().numpy()[:m], want_idx)
    nbr_out, nbr_in = _tables_from_pairs(pairs, pair_num, len(idx), m)
    np.testing.assert_array_equal(rb["nbr_out"].cpu().numpy()[:m], nbr_out)
    assert (rb["nbr_out"].cpu().numpy()[m:] == -1).all()
    np.testing.assert_array_equal(rb["nbr_in"].cpu().numpy()[:len(idx)], nbr_in)
    small = ops.rulebook_conv(dev(padded), 2, shape, 3, 2, 1, 1, n_dev=n_dev, out_cap=m - 40, numbering="sorted")
    cnt = small["num_out_dev"].cpu().numpy()
    assert cnt.tolist() == [m - 40, m]
    np.testing.assert_array_equal(small["out_indices"].cpu().numpy(), want_idx[:m - 40])
    np.testing.assert_array_equal(small["nbr_out"].cpu().numpy(), nbr_out[:m - 40])


def test_site_map_from_the_sorted_builds_bitmap(ops):
    """sec_sparse_site_map_sorted (one launch off the strided build's bitmap) == the generic fill + scatter map of the same
    outputs, eager and static-capacity, including a capacity smaller than the output count (rows past it read as absent)."""
    rng = np.random.default_rng(9)
    shape = (11, 24, 19)
    idx = _random_indices(rng, 3, shape, 500)
    rb = ops.rulebook_conv(dev(idx), 3, shape, 3, 2, 1, 1, numbering="sorted")
    m = rb["num_out"]
    want = ops.sparse_site_map(rb["out_indices"], 3, rb["out_shape"])
    got = ops.sparse_site_map_sorted(rb["site_table"][1], m, 3, rb["out_shape"])
    assert torch.equal(want, got) and int((got > 0).sum()) == m
    n_dev = dev(np.array([len(idx)], np.int32))
    for cap in (m + 30, m - 25):
        st = ops.rulebook_conv(dev(idx), 3, shape, 3, 2, 1, 1, n_dev=n_dev, out_cap=cap, numbering="sorted")
        want = ops.sparse_site_map(st["out_indices"], 3, st["out_shape"], num_dev=st["num_out_dev"])
        got = ops.sparse_site_map_sorted(st["site_table"][1], cap, 3, st["out_shape"], num_dev=st["num_out_dev"])
        assert torch.equal(want, got) and int((got > 0).sum()) == min(cap, m)


def test_rulebook_conv_stride_with_dilation_is_refused(ops):
    from second_amd.runtime import SecondHipError
    idx = _random_indices(np.random.default_rng(2), 1, (11, 24, 19), 50)
    with pytest.raises(SecondHipError):
        ops.rulebook_conv(dev(idx), 1, (11, 24, 19), 3, 2, 2, 2)


def test_rulebook_empty_and_single(ops):
    e = torch.zeros((0, 4), dtype=torch.int32, device="cuda")
    rb = ops.rulebook_subm(e, 1, (5, 5, 5), 3, 1, want_pairs=True)
    assert rb["pair_num"].sum().item() == 0
    rb = ops.rulebook_conv(e, 1, (5, 5, 5), 3, 2, 1, 1, want_pairs=True)
    assert rb["num_out"] == 0 and rb["pair_num"].sum().item() == 0
    one = dev(np.array([[0, 1, 1, 1]], np.int32))
    rb = ops.rulebook_conv(one, 1, (5, 5, 5), 3, 2, 1, 1, want_pairs=True)
    assert rb["num_out"] == 8
    assert rb["out_indices"][0].tolist() == [0, 1, 1, 1] and rb["out_indices"][-1].tolist() == [0, 0, 0, 0]


def test_rulebook_car_fhd_stack(ops, syn):
    """The four SubM + four strided rulebooks of SpMiddleFHD (middle.py:146-189) on a real-size frame pair."""
    clouds = [syn.syn_kitti_cloud(s) for s in range(2)]
    coors = []
    for b, c in enumerate(clouds):
        r = orc.points_to_voxel(c, syn.CAR_FHD_VOXEL, syn.CAR_FHD_RANGE, 5, 40000)
        coors.append(np.concatenate([np.full((r["voxel_num"], 1), b, np.int32), r["coordinates"]], 1))
    idx = np.concatenate(coors).astype(np.int32)
    shape = [41, 1600, 1408]
    for ks, st, pd in ((3, 2, 1), (3, 2, 1), (3, 2, (0, 1, 1)), ((3, 1, 1), (2, 1, 1), 0)):
        rb = ops.rulebook_subm(dev(idx), 2, shape, 3, 1, want_pairs=True)
        _, pairs, pair_num = orc.rulebook_subm(idx, 2, shape, 3)
        np.testing.assert_array_equal(rb["pair_num"].cpu().numpy(), pair_num)
        np.testing.assert_array_equal(rb["pairs"].cpu().numpy(), pairs)
        rb = ops.rulebook_conv(dev(idx), 2, shape, ks, st, pd, 1, want_pairs=True)
        out_idx, pairs, pair_num, out_shape = orc.rulebook_conv(idx, 2, shape, ks, st, pd)
        np.testing.assert_array_equal(rb["out_indices"].cpu().numpy(), out_idx)
        np.testing.assert_array_equal(rb["pairs"].cpu().numpy(), pairs)
        idx, shape = out_idx, out_shape.tolist()
    assert shape == [2, 200, 176]


def test_rulebook_nuscenes_fhd_stack_vs_oracle(ops, syn):
    """BASELINE config 5 at its stated size: the rulebooks of SpMiddleFHD on the all.fhd grid (41 x 1984 x 1984) for two
    block-filtered synthetic nuScenes frames (~85 k voxels each) -- every SubM pair list and both numberings of every strided
    layer (first touch = spconv's CPU order, sorted = its GPU order, the chain fed from the previous layer's bitmap) element by
    element against the oracle."""
    from second_amd.models import ALL_FHD_NUSC as C
    rng_, vs, bf = C["point_cloud_range"], C["voxel_size"], C["block_filtering"]
    coors = []
    for b, seed in enumerate((3, 4)):
        c = syn.syn_nusc_cloud(seed, 293000, tuple(rng_), scene="urban")
        r = orc.points_to_voxel(c, vs, rng_, 1, C["max_voxels"])
        keep = orc.block_filter(r["voxels"], r["coordinates"], r["num_points_per_voxel"], [1984, 1984], bf["block_factor"], bf["block_size"],
                                bf["height_threshold"], 3.0)
        coors.append(np.concatenate([np.full((int(keep.sum()), 1), b, np.int32), r["coordinates"][keep]], 1))
    idx = np.concatenate(coors).astype(np.int32)
    assert len(idx) > 160000
    idx_sorted, sites, shape = idx, None, [41, 1984, 1984]
    for ks, st, pd in ((3, 2, 1), (3, 2, 1), (3, 2, (0, 1, 1)), ((3, 1, 1), (2, 1, 1), 0)):
        rb = ops.rulebook_subm(dev(idx), 2, shape, 3, 1, want_pairs=True)
        _, pairs, pair_num = orc.rulebook_subm(idx, 2, shape, 3)
        np.testing.assert_array_equal(rb["pair_num"].cpu().numpy(), pair_num)
        np.testing.assert_array_equal(rb["pairs"].cpu().numpy(), pairs)
        rb = ops.rulebook_conv(dev(idx), 2, shape, ks, st, pd, 1, want_pairs=True)
        out_idx, pairs, pair_num, out_shape = orc.rulebook_conv(idx, 2, shape, ks, st, pd)
        np.testing.assert_array_equal(rb["out_indices"].cpu().numpy(), out_idx)
        np.testing.assert_array_equal(rb["pairs"].cpu().numpy(), pairs)
        rs = ops.rulebook_conv(dev(idx_sorted), 2, shape, ks, st, pd, 1, want_pairs=True, numbering="sorted", in_sites=sites)
        s_idx, s_pairs, s_num, _ = orc.rulebook_conv_sorted(idx_sorted, 2, shape, ks, st, pd)
        np.testing.assert_array_equal(rs["out_indices"].cpu().numpy(), s_idx)
        np.testing.assert_array_equal(rs["pair_num"].cpu().numpy(), s_num)
        np.testing.assert_array_equal(rs["pairs"].cpu().numpy(), s_pairs)
        idx, idx_sorted, sites, shape = out_idx, s_idx, rs["site_table"], out_shape.tolist()
    assert shape == [2, 248, 248]


# ------------------------------------------------------------------ indice_conv
CONV_SHAPES = [(4, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (5, 7), (64, 128)]


def _conv_case(rng, cin, cout, subm):
    shape = (8, 30, 28)
    idx = _random_indices(rng, 2, shape, 900)
    if subm:
        _, pairs, pair_num = orc.rulebook_subm(idx, 2, shape, 3)
        n_out = len(idx)
    else:
        out_idx, pairs, pair_num, _ = orc.rulebook_conv(idx, 2, shape, 3, 2, 1)
        n_out = len(out_idx)
    feat = rng.standard_normal((len(idx), cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    nbr_out, nbr_in = _tables_from_pairs(pairs, pair_num, len(idx), n_out)
    return feat, w, pairs, pair_num, nbr_out, nbr_in, n_out


@pytest.mark.parametrize("cin,cout", CONV_SHAPES)
@pytest.mark.parametrize("subm", [True, False])
def test_indice_conv_fp32(ops, cin, cout, subm):
    rng = np.random.default_rng(cin * 131 + cout)
    feat, w, pairs, pair_num, nbr_out, _, n_out = _conv_case(rng, cin, cout, subm)
    ref = orc.indice_conv(feat, w, pairs, pair_num, n_out, acc64=True)
    out = ops.indice_conv(dev(feat), dev(w), dev(nbr_out), n_out).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    # PER ELEMENT (the line above is relative to the tensor's maximum): the forward error bound of the arithmetic the kernel runs --
    # every product within 3 * 2^-18 of exact (fp32 operands carried as two bf16 halves, the lo * lo term dropped: DESIGN 4) plus an
    # fp32 accumulation over K = 27 * cin terms -- against the sum of the products' magnitudes of THAT output element
    mag = orc.indice_conv(np.abs(feat), np.abs(w), pairs, pair_num, n_out, acc64=True)
    bound = (4 * 2.0 ** -18 + (27 * cin + 2) * 2.0 ** -24) * mag + 1e-30
    assert np.all(np.abs(out - ref) <= bound), float((np.abs(out - ref) / bound).max())


@pytest.mark.parametrize("cin,cout", CONV_SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_indice_conv_half_mfma(ops, cin, cout, dtype):
    """bf16/f16 inputs, fp32 accumulate: vs the fp32 oracle on the SAME rounded inputs, <= 1e-4 relative."""
    rng = np.random.default_rng(cin * 17 + cout)
    feat, w, pairs, pair_num, nbr_out, _, n_out = _conv_case(rng, cin, cout, True)
    f_t, w_t = dev(feat, dtype), dev(w, dtype)
    ref = orc.indice_conv(f_t.float().cpu().numpy(), w_t.float().cpu().numpy(), pairs, pair_num, n_out, acc64=True)
    packed = ops.pack_weight(w_t)
    assert (packed is not None) == (cin % 16 == 0 or (cin, cout) == (4, 16))      # MFMA layouts: Cin multiple of 16, or the 4 -> 16 first layer
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.uniform(-0.2, 0.2, cout).astype(np.float32)
    out32 = ops.indice_conv(f_t, w_t, dev(nbr_out), n_out, packed=packed, out_dtype=torch.float32).cpu().numpy()
    np.testing.assert_allclose(out32, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    # per element: 16-bit x 16-bit products are exact in fp32, so only the fp32 accumulation of K = 27 * cin terms is left --
    # gamma_K * sum |x| |w| of that element
    mag = orc.indice_conv(np.abs(f_t.float().cpu().numpy()), np.abs(w_t.float().cpu().numpy()), pairs, pair_num, n_out, acc64=True)
    bound = (27 * cin + 2) * 2.0 ** -24 * mag + 1e-30
    assert np.all(np.abs(out32 - ref) <= bound), float((np.abs(out32 - ref) / bound).max())
    # generic path on the same inputs must agree too (cross-check of the MFMA fragment layouts)
    gen = ops.indice_conv(f_t, w_t, dev(nbr_out), n_out, packed=None, out_dtype=torch.float32).cpu().numpy()
    np.testing.assert_allclose(out32, gen, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    # fused epilogue + low-precision store: within one rounding of the fp32 result
    fused = ops.indice_conv(f_t, w_t, dev(nbr_out), n_out, packed=packed, scale=dev(scale), shift=dev(shift), relu=True)
    assert fused.dtype == dtype
    ref_f = torch.from_numpy(np.maximum(ref * scale + shift, 0)).to(dtype).float().numpy()
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
    np.testing.assert_allclose(fused.float().cpu().numpy(), ref_f, rtol=tol, atol=tol * np.abs(ref_f).max())


def test_indice_conv_asymmetric_identity(ops):
    """A = one-hot rows, asymmetric B: catches transposed MFMA operand / output layouts."""
    cin = cout = 64
    n = 70
    nbr = -np.ones((n, 27), np.int32)
    nbr[:, 13] = np.arange(n)
    feat = np.zeros((n, cin), np.float32)
    feat[np.arange(n), np.arange(n) % cin] = 1.0
    w = np.zeros((27, cin, cout), np.float32)
    w[13] = np.arange(cin * cout, dtype=np.float32).reshape(cin, cout) / 64.0  # exact in bf16? use small ints
    w[13] = (np.arange(cin)[:, None] * 2 + np.arange(cout)[None, :] % 2).astype(np.float32)
    f_t, w_t = dev(feat, torch.bfloat16), dev(w.reshape(3, 3, 3, cin, cout), torch.bfloat16)
    out = ops.indice_conv(f_t, w_t, dev(nbr), n, packed=ops.pack_weight(w_t), out_dtype=torch.float32).cpu().numpy()
    np.testing.assert_array_equal(out, w[13][np.arange(n) % cin])


@pytest.mark.parametrize("subm", [True, False])
def test_indice_conv_backward(ops, subm):
    rng = np.random.default_rng(5)
    feat, w, pairs, pair_num, nbr_out, nbr_in, n_out = _conv_case(rng, 16, 32, subm)
    dout = rng.standard_normal((n_out, 32)).astype(np.float32)
    dfeat_ref, dw_ref = orc.indice_conv_backward(feat, w, pairs, pair_num, dout)
    dfeat, dw = ops.indice_conv_backward(dev(feat), dev(w), dev(nbr_out), None if subm else dev(nbr_in), dev(dout))
    np.testing.assert_allclose(dfeat.cpu().numpy(), dfeat_ref, rtol=1e-4, atol=1e-4 * np.abs(dfeat_ref).max())
    np.testing.assert_allclose(dw.cpu().numpy(), dw_ref, rtol=1e-4, atol=1e-4 * np.abs(dw_ref).max())


@pytest.mark.parametrize("cin,cout", [(16, 16), (16, 32), (32, 64), (64, 64), (4, 16)])
@pytest.mark.parametrize("subm", [True, False])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_indice_conv_backward_shapes_and_dtypes(ops, cin, cout, subm, dtype):
    """dgrad on the MFMA forward kernels (16-bit dtypes: transposed, offset-mirrored packed weights) and the
    register-tiled wgrad, vs the fp32 oracle on the same (rounded) operands.  SubM with cin != cout included: 4 -> 16 is the first
    layer of car.fhd (middle.py:146), 16 -> 32 / 32 -> 64 the widths of the nuScenes stacks."""
    rng = np.random.default_rng(cin * 100 + cout)
    feat, w, pairs, pair_num, nbr_out, nbr_in, n_out = _conv_case(rng, cin, cout, subm)
    dout = rng.standard_normal((n_out, cout)).astype(np.float32)
    rnd = lambda a: torch.from_numpy(a).to(dtype).float().numpy()
    feat, w, dout = rnd(feat), rnd(w), rnd(dout)
    dfeat_ref, dw_ref = orc.indice_conv_backward(feat, w, pairs, pair_num, dout)
    dfeat, dw = ops.indice_conv_backward(dev(feat).to(dtype), dev(w).to(dtype), dev(nbr_out), None if subm else dev(nbr_in),
                                         dev(dout).to(dtype))
    tol = {torch.float32: 1e-4, torch.float16: 2 ** -9, torch.bfloat16: 2 ** -7}[dtype]
    np.testing.assert_allclose(dfeat.float().cpu().numpy(), dfeat_ref, rtol=tol, atol=tol * np.abs(dfeat_ref).max())
    np.testing.assert_allclose(dw.float().cpu().numpy(), dw_ref, rtol=tol, atol=tol * np.abs(dw_ref).max())


def test_indice_conv_car_fhd_layer(ops, syn):
    """Full-size subm2-like layer (64->64) on a synthetic frame: bf16 MFMA vs oracle."""
    c = syn.syn_kitti_cloud(0)
    r = orc.points_to_voxel(c, syn.CAR_FHD_VOXEL, syn.CAR_FHD_RANGE, 5, 40000)
    idx = np.concatenate([np.zeros((r["voxel_num"], 1), np.int32), r["coordinates"]], 1)
    _, pairs, pair_num = orc.rulebook_subm(idx, 1, [41, 1600, 1408], 3)
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((len(idx), 64)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, 64, 64)) / 40).astype(np.float32)
    f_t, w_t = dev(feat, torch.bfloat16), dev(w, torch.bfloat16)
    rb = ops.rulebook_subm(dev(idx), 1, [41, 1600, 1408], 3)
    out = ops.indice_conv(f_t, w_t, rb["nbr_out"], len(idx), packed=ops.pack_weight(w_t), out_dtype=torch.float32)
    ref = orc.indice_conv(f_t.float().cpu().numpy(), w_t.float().cpu().numpy(), pairs, pair_num, len(idx))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


# ------------------------------------------------------------------ scatters
def test_sparse_to_dense(ops):
    rng = np.random.default_rng(7)
    shape = (2, 20, 18)
    idx = _random_indices(rng, 3, shape, 100)
    feat = rng.standard_normal((len(idx), 64)).astype(np.float32)
    ref = orc.sparse_to_dense(feat, idx, 3, shape)
    out = ops.sparse_to_dense(dev(feat), dev(idx), 3, shape)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    cl = ops.sparse_to_dense(dev(feat, torch.bfloat16), dev(idx), 3, shape, channels_last_2d=True)
    assert cl.shape == (3, 128, 20, 18) and cl.is_contiguous(memory_format=torch.channels_last)
    ref_bf = torch.from_numpy(ref).to(torch.bfloat16).view(3, 128, 20, 18)
    assert torch.equal(cl.cpu(), ref_bf)


def test_pillar_scatter_golden(ops, golden):
    g = golden("torch_modules")
    b, c, ny, nx = g["ps_out"].shape
    out = ops.pillar_scatter(dev(g["ps_feats"]), dev(g["ps_coords"]), b, ny, nx)
    np.testing.assert_array_equal(out.cpu().numpy(), g["ps_out"])


# ------------------------------------------------------------------ IoU / NMS
def test_rotate_iou_golden(ops, golden):
    g = golden("rotate_iou")
    for crit in (-1, 0, 1, 2):
        out = ops.rotate_iou(dev(g["boxes"]), dev(g["qboxes"]), crit).cpu().numpy()
        np.testing.assert_allclose(out, g[f"iou_c{crit}"], atol=2e-5, rtol=1e-5)
    sp = np.array([ops.rotate_iou(dev(g["special_a"][i:i + 1]), dev(g["special_b"][i:i + 1]))[0, 0].item()
                   for i in range(len(g["special_a"]))])
    np.testing.assert_allclose(sp, g["special_iou"], atol=2e-5)


def test_rotate_iou_degenerate_and_tied_polygons_vs_oracle(ops):
    """The clipper keeps its vertex list in registers and replays the reference's insertion sort with predicates (csrc/nms.hip):
    the cases where the ORDER of appended vertices, ties between sort keys or a degenerate polygon decide the result must give the
    oracle's result -- identical boxes (8 coincident vertices), boxes sharing an edge or only a corner, a box inside another,
    45 / 90 degree rotations of the same box (symmetric keys), sliver and dot-sized boxes, far-apart boxes, a cross, plus a
    dense jittered cluster; all four criteria (values to 2e-5, zeros and non-finite entries exactly)."""
    base = np.array([[0, 0, 2, 4, 0.0], [0, 0, 2, 4, 0.0], [2, 0, 2, 4, 0.0], [2, 4, 2, 4, 0.0], [0, 0, 1, 1, 0.3],
                     [0, 0, 2, 4, np.pi / 2], [0, 0, 2, 4, np.pi / 4], [0, 0, 4, 2, 0.0], [0, 0, 1e-3, 4, 0.0], [7, 7, 1e-3, 1e-3, 0.0],
                     [50, 50, 2, 4, 1.0], [0, 0, 6, 1, 0.0], [0, 0, 1, 6, 0.0], [0.5, 0.25, 2, 4, 1e-3], [0, 0, 2, 4, np.pi],
                     [1, 2, 2, 4, 0.0], [0, 0, 2, 2, np.pi / 4], [0, 0, 2.8284271, 2.8284271, 0.0]], np.float32)
    rng = np.random.default_rng(21)
    k = 150
    cluster = np.stack([rng.normal(0, 0.6, k), rng.normal(0, 0.6, k), rng.uniform(1.4, 2.0, k), rng.uniform(3.4, 4.6, k),
                        rng.normal(0.3, 0.2, k)], 1).astype(np.float32)
    boxes = np.concatenate([base, cluster])
    for crit in (-1, 0, 1, 2):
        want = orc.rotate_iou(boxes, boxes, crit)
        got = ops.rotate_iou(dev(boxes), dev(boxes), crit).cpu().numpy()
        # sinf / cosf of the device and of the host libm may differ in the last bit, so rotated corners are compared to 2e-5 like the
        # golden test; everything is finite (no zero-area box: x / 0 is outside the reference's domain too) and the exact zeros coincide
        # A ROTATED box against itself is outside the comparison: all its edges are colinear with the other's, the reference
        # algorithm's crossing tests are then decided by the last bit of sin / cos, and it returns 1 or 1/3 depending on the libm
        # (device and host disagree on 5 of 150 such pairs; identical AXIS-ALIGNED boxes -- rows 0 and 1 -- are exact and compared).
        off = ~np.eye(len(boxes), dtype=bool)
        assert np.isfinite(want[off]).all() and np.isfinite(got[off]).all(), crit
        np.testing.assert_allclose(got[off], want[off], atol=2e-5, rtol=1e-5)
        assert np.array_equal(got[off] == 0.0, want[off] == 0.0), crit


def _nms_call(ops, dets_sorted_list, thr, kind, semantics, eps=1.0, post_max=0):
    b = len(dets_sorted_list)
    max_n = max(1, max(len(d) for d in dets_sorted_list))
    stride = dets_sorted_list[0].shape[1]
    buf = np.zeros((b, max_n, stride), np.float32)
    for i, d in enumerate(dets_sorted_list):
        buf[i, :len(d)] = d
    counts = np.array([len(d) for d in dets_sorted_list], np.int32)
    keep, num = ops.nms_sorted(dev(buf), dev(counts), thr, kind, semantics, eps, post_max)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    return [keep[i, :num[i]] for i in range(b)]


@pytest.mark.parametrize("thr", [0.01, 0.3])
def test_rotate_nms_golden_batched(ops, golden, thr):
    g = golden("rotate_nms")
    tags = ["a", "b", "c", "d"]
    orders = [g[f"dets_{t}"][:, 5].argsort()[::-1] for t in tags]
    sorted_dets = [g[f"dets_{t}"][o] for t, o in zip(tags, orders)]
    for sem in ("numba", "cpu"):
        keeps = _nms_call(ops, sorted_dets, thr, "rotate", sem)
        for t, o, k in zip(tags, orders, keeps):
            np.testing.assert_array_equal(o[k], g[f"keep_{t}_{thr}"])


def test_rotate_nms_1000_boxes_vs_oracle(ops):
    rng = np.random.default_rng(11)
    n = 1000
    dets = np.concatenate([rng.uniform(0, 70, (n, 1)), rng.uniform(-40, 40, (n, 1)), rng.uniform(1.4, 1.8, (n, 1)),
                           rng.uniform(3.5, 4.3, (n, 1)), rng.uniform(-3.2, 3.2, (n, 1)),
                           np.sort(rng.uniform(0.3, 1, (n, 1)), 0)[::-1]], 1).astype(np.float32)
    for sem in ("numba", "cpu"):
        for post in (0, 100):
            k = _nms_call(ops, [dets, dets[:333]], 0.01, "rotate", sem, post_max=post)
            r0 = orc.rotate_nms_sorted(dets, 0.01, sem)
            r1 = orc.rotate_nms_sorted(dets[:333], 0.01, sem)
            np.testing.assert_array_equal(k[0], r0[:post] if post else r0)
            np.testing.assert_array_equal(k[1], r1[:post] if post else r1)


@pytest.mark.parametrize("n", [700, 1100, 1500])
def test_rotate_nms_reduce_forms_vs_oracle(ops, n):
    """Candidate counts off the usual 1000 (odd and even numbers of 64-bit mask words per row, up to 24): keep lists equal the sequential
    oracle's, ragged batch (65 = one box into the second block, n - 1), with and without a post-NMS cap."""
    rng = np.random.default_rng(n)
    dets = np.concatenate([rng.uniform(0, 70, (n, 1)), rng.uniform(-40, 40, (n, 1)), rng.uniform(1.4, 1.8, (n, 1)),
                           rng.uniform(3.5, 4.3, (n, 1)), rng.uniform(-3.2, 3.2, (n, 1)),
                           np.sort(rng.uniform(0.3, 1, (n, 1)), 0)[::-1]], 1).astype(np.float32)
    for post in (0, 83):
        k = _nms_call(ops, [dets, dets[:65], dets[:n - 1]], 0.05, "rotate", "numba", post_max=post)
        for kk, d in zip(k, (dets, dets[:65], dets[:n - 1])):
            r = orc.rotate_nms_sorted(d, 0.05, "numba")
            np.testing.assert_array_equal(kk, r[:post] if post else r)


@pytest.mark.parametrize("thr", [0.01, 0.1, 0.3, 0.5])
def test_rotate_nms_clustered_candidates_vs_oracle(ops, thr):
    """The candidates of a detector cluster on the objects (a few dozen jittered boxes per object): the regime where the mask
    kernel decides most overlapping pairs with the three-inscribed-circles lower bound of the IoU instead of the polygon clipper
    (csrc/nms.hip phase A) and where one 64 x 64 tile used to queue > 1000 pairs.  Keep lists must still equal the sequential
    oracle's, for both semantics, elongated and near-square boxes, thresholds from car.fhd's 0.01 to 0.5, and a ragged batch."""
    rng = np.random.default_rng(int(thr * 1000) + 5)
    frames = []
    for n_obj, per, aspect in ((13, 32, (1.5, 1.9, 3.4, 4.6)), (6, 60, (0.6, 2.6, 0.6, 2.6)), (25, 40, (1.4, 2.0, 3.0, 12.0))):
        cx, cy = rng.uniform(0, 70, n_obj), rng.uniform(-40, 40, n_obj)
        rot = rng.uniform(-3.2, 3.2, n_obj)
        k = n_obj * per
        o = rng.integers(0, n_obj, k)
        # jitter up to a box length: same-object pairs run from nearly identical to barely touching
        x = cx[o] + rng.normal(0, 0.9, k)
        y = cy[o] + rng.normal(0, 0.9, k)
        w = rng.uniform(aspect[0], aspect[1], k)
        l = rng.uniform(aspect[2], aspect[3], k)
        r = rot[o] + rng.normal(0, 0.25, k) + (rng.random(k) < 0.1) * np.pi / 2
        sc = np.sort(rng.uniform(0.3, 1, k))[::-1]
        frames.append(np.stack([x, y, w, l, r, sc], 1).astype(np.float32))
    for sem in ("numba", "cpu"):
        keeps = _nms_call(ops, frames, thr, "rotate", sem)
        for d, k in zip(frames, keeps):
            np.testing.assert_array_equal(k, orc.rotate_nms_sorted(d, thr, sem))


@pytest.mark.parametrize("thr", [0.1, 0.5])
def test_axis_aligned_nms_golden(ops, golden, thr):
    g = golden("nms_axis_aligned")
    dets = g["dets"]
    order = dets[:, 4].argsort()[::-1]
    k = _nms_call(ops, [dets[order]], thr, "axis_aligned", "numba")[0]
    np.testing.assert_array_equal(order[k], g[f"keep_gpu_{thr}"])
    k = _nms_call(ops, [dets[order]], thr, "axis_aligned", "cpu", eps=0.0)[0]
    np.testing.assert_array_equal(order[k], g[f"keep_jit_eps0_{thr}"])
    k = _nms_call(ops, [dets[order]], thr, "axis_aligned", "cpu", eps=1.0)[0]
    np.testing.assert_array_equal(order[k], g[f"keep_jit_eps1_{thr}"])


def test_cpu_tensors_are_rejected(ops):
    from second_amd.runtime import SecondHipError
    with pytest.raises(SecondHipError):
        ops.rulebook_subm(torch.zeros((1, 4), dtype=torch.int32), 1, (3, 3, 3))


# ------------------------------------------------------------------ static-capacity (sync-free) pipeline
def test_static_capacity_rulebooks_match_eager(ops, syn):
    c = syn.syn_kitti_cloud(1, num_points=6000, num_voxels=5000)
    r = orc.points_to_voxel(c, syn.CAR_FHD_VOXEL, syn.CAR_FHD_RANGE, 5, 40000)
    idx = np.concatenate([np.zeros((r["voxel_num"], 1), np.int32), r["coordinates"]], 1)
    n = len(idx)
    cap = n + 777
    padded = np.concatenate([idx, np.full((cap - n, 4), 12345, np.int32)])  # garbage rows beyond the live count
    n_dev = dev(np.array([n], np.int32))
    shape = [41, 1600, 1408]
    e = ops.rulebook_subm(dev(idx), 1, shape, 3)
    s = ops.rulebook_subm(dev(padded), 1, shape, 3, n_dev=n_dev)
    assert torch.equal(e["nbr_out"], s["nbr_out"][:n])
    e = ops.rulebook_conv(dev(idx), 1, shape, 3, 2, 1)
    s = ops.rulebook_conv(dev(padded), 1, shape, 3, 2, 1, n_dev=n_dev, out_cap=4 * cap, out_per_in_hint=4)
    m = e["num_out"]
    assert s["num_out_dev"].tolist() == [m, m]
    assert torch.equal(e["out_indices"], s["out_indices"][:m])
    assert torch.equal(e["nbr_out"], s["nbr_out"][:m])
    assert torch.equal(e["nbr_in"], s["nbr_in"][:n])
    # capacity overflow is reported, never a hang or an out-of-bounds write
    s = ops.rulebook_conv(dev(padded), 1, shape, 3, 2, 1, n_dev=n_dev, out_cap=100, out_per_in_hint=2)
    assert s["num_out_dev"][0].item() == 100 and s["num_out_dev"][1].item() == m
    s = ops.rulebook_conv(dev(padded[:64]), 1, shape, 3, 2, 1, out_cap=64, out_per_in_hint=1)
    assert s["num_out_dev"][1].item() >= 64


def test_detector_static_and_graph_match_eager(syn):
    from second_amd.models import SecondDetector, CAR_FHD
    torch.manual_seed(0)
    det = SecondDetector(CAR_FHD).cuda().prepare_inference(torch.bfloat16)
    clouds = [syn.syn_kitti_cloud(s, num_points=9000, num_voxels=8000) for s in range(2)]
    pts, offs = syn.batch_clouds(clouds)
    pts, offs = dev(pts), dev(offs)
    with torch.no_grad():
        e = det.forward_points(pts, offs)
        s = det.forward_points(pts, offs, static=True)   # default growth-factor capacities
        det.check_overflow()
        det.calibrate(pts, offs)                          # profile-sized capacities
        replay, g = det.make_graphed(pts, offs)
        replay()
        torch.cuda.synchronize()
    assert e["valid"].any()
    for other in (s, g):
        assert torch.equal(e["valid"], other["valid"])
        m = e["valid"]
        assert torch.equal(e["scores"][m], other["scores"][m])
        assert torch.equal(e["boxes"][m], other["boxes"][m])


def test_rpn_mirror_matches_reference_golden(golden):
    """RPNV2 mirror vs outputs of the reference's own RPNV2 module (fixture, fp32)."""
    from second_amd.models import RPNV2
    g = golden("torch_modules")
    net = RPNV2(num_class=1, layer_nums=(2,), layer_strides=(1,), num_filters=(16,), upsample_strides=(1,),
                num_upsample_filters=(16,), num_input_features=16)
    sd = {k[len("rpn_sd."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("rpn_sd.")}
    net.load_state_dict(sd)
    net = net.cuda().eval()
    with torch.no_grad():
        r = net(dev(g["rpn_in"]))
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        np.testing.assert_allclose(r[k].cpu().numpy(), g["rpn_" + k], rtol=1e-4, atol=1e-5)


def test_rpn_inference_form_matches_module(golden):
    """Folded-BN + fused bias/ReLU + merged heads == the plain module (fp32, tight) and bf16 (loose)."""
    from second_amd.models import RPNV2, RPNInference
    torch.manual_seed(0)
    net = RPNV2(num_class=1, layer_nums=(3,), layer_strides=(1,), num_filters=(32,), upsample_strides=(1,),
                num_upsample_filters=(32,), num_input_features=32).cuda().eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.1, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.2, 0.2)
    x = torch.randn(2, 32, 24, 20, device="cuda")
    with torch.no_grad():
        ref = net(x)
        f32 = RPNInference(net, torch.float32)(x.contiguous(memory_format=torch.channels_last))
        bf = RPNInference(net, torch.bfloat16)(x.bfloat16().contiguous(memory_format=torch.channels_last))
    for k in ref:
        np.testing.assert_allclose(f32[k].cpu().numpy(), ref[k].cpu().numpy(), rtol=1e-3, atol=1e-4)
        err = (bf[k].float() - ref[k]).abs().max().item() / ref[k].abs().max().item()
        assert err < 3e-2, (k, err)


def test_rpn_inference_multi_block_pointpillars_shape():
    """The 3-block RPN of nuscenes/all.pp.largea (strided first convs, k = s strided "upsample" convs, 1x1 deblock, 384-channel
    concat, 228 -> 256 padded head channels) entirely on sec_conv2d_nhwc: bf16 kernels vs the fp32 module, and the fp32
    folded form (torch convs) tight against it."""
    from second_amd.models import RPNV2, RPNInference, ALL_PP_LARGEA, anchors_per_location
    torch.manual_seed(0)
    cfg = ALL_PP_LARGEA
    net = RPNV2(num_class=cfg["num_class"], num_anchor_per_loc=anchors_per_location(cfg),
                num_direction_bins=cfg["num_direction_bins"], **cfg["rpn"]).cuda().eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.1, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.2, 0.2)
    assert RPNInference.supports(net) and len(net.blocks) == 3
    x = torch.randn(2, 64, 96, 80, device="cuda")
    with torch.no_grad():
        ref = net(x)
        f32 = RPNInference(net, torch.float32)(x.contiguous(memory_format=torch.channels_last))
        inf = RPNInference(net, torch.bfloat16)
        assert inf.use_hip and not inf.chain_tail
        xb = x.bfloat16().contiguous(memory_format=torch.channels_last)
        bf = inf(xb)
        # the deblocks deposit their outputs straight into the concatenated map (sec_conv2d_nhwc_into): bit-identical to torch.cat of them
        assert inf._deblocks_write_into_the_concat()
        cat = RPNInference(net, torch.bfloat16)
        cat.concat_in_place = False
        bf_cat = cat(xb)
        assert not cat._deblocks_write_into_the_concat()
        for k in bf:
            assert torch.equal(bf[k], bf_cat[k]), k
    for k in ref:
        assert f32[k].shape == ref[k].shape == bf[k].shape
        np.testing.assert_allclose(f32[k].cpu().numpy(), ref[k].cpu().numpy(), rtol=2e-3, atol=2e-4)
        err = (bf[k].float() - ref[k]).abs().max().item() / ref[k].abs().max().item()
        assert err < 4e-2, (k, err)


def test_rpn_inference_transposed_deblocks_kitti_pointpillars_shape():
    """KITTI PointPillars RPN (configs/pointpillars/car/xyres_16.config: upsample strides 1, 2, 4 = ConvTranspose2d with
    k = s): each deblock runs as a 1x1 conv to s*s*Cout channels + depth-to-space; bf16 kernels and the fp32 folded form
    against the fp32 module."""
    from second_amd.models import RPNV2, RPNInference
    torch.manual_seed(0)
    net = RPNV2(num_class=1, layer_nums=(3, 5, 5), layer_strides=(2, 2, 2), num_filters=(64, 128, 256),
                upsample_strides=(1, 2, 4), num_upsample_filters=(128, 128, 128), num_input_features=64,
                num_anchor_per_loc=2).cuda().eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.1, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.2, 0.2)
    assert RPNInference.supports(net)
    x = torch.randn(2, 64, 64, 48, device="cuda")
    with torch.no_grad():
        ref = net(x)
        f32 = RPNInference(net, torch.float32)(x.contiguous(memory_format=torch.channels_last))
        inf = RPNInference(net, torch.bfloat16)
        assert inf.use_hip and inf.ups.count(1) == len(inf.ups) - 2 and sorted(u for u in inf.ups if u > 1) == [2, 4]
        bf = inf(x.bfloat16().contiguous(memory_format=torch.channels_last))
    for k in ref:
        assert f32[k].shape == ref[k].shape == bf[k].shape
        np.testing.assert_allclose(f32[k].cpu().numpy(), ref[k].cpu().numpy(), rtol=2e-3, atol=2e-4)
        err = (bf[k].float() - ref[k]).abs().max().item() / ref[k].abs().max().item()
        assert err < 4e-2, (k, err)


# ------------------------------------------------------------------ PointPillars front end / block filter
def test_pfn_kernel_matches_reference_module_and_oracle(ops, golden):
    g = golden("torch_modules")
    W = g["pfn_sd.pfn_layers.0.linear.weight"]
    bw, bb, rm, rv = [g["pfn_sd.pfn_layers.0.norm." + k] for k in ("weight", "bias", "running_mean", "running_var")]
    scale = bw / np.sqrt(rv + 1e-3)
    shift = bb - rm * scale
    out = ops.pfn_forward(dev(g["pfn_voxels"]), dev(g["pfn_num_points"]), dev(g["pfn_coords"]), dev(W.T.copy()),
                          dev(scale), dev(shift), 0.25, 0.25, 0.125 - 50, 0.125 - 50)
    np.testing.assert_allclose(out.cpu().numpy(), g["pfn_out"], rtol=1e-4, atol=1e-5)   # the reference's own module
    rng = np.random.default_rng(0)
    P, T, C = 3000, 60, 64
    vox = rng.uniform(-3, 3, (P, T, 4)).astype(np.float32)
    n = rng.integers(1, T + 1, P).astype(np.int32)
    n[:10] = T
    for i in range(P):
        vox[i, n[i]:] = 0
    coords = np.stack([rng.integers(0, 4, P), np.zeros(P, int), rng.integers(0, 400, P), rng.integers(0, 400, P)], 1).astype(np.int32)
    Wt = (rng.standard_normal((9, C)) / 3).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.uniform(-0.3, 0.3, C).astype(np.float32)
    ref = orc.pfn_forward(vox, n, coords, Wt, sc, sh, 0.25, 0.25, -49.875, -49.875)
    out = ops.pfn_forward(dev(vox), dev(n), dev(coords), dev(Wt), dev(sc), dev(sh), 0.25, 0.25, -49.875, -49.875)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


def test_pointpillars_front_end_at_config4_size_vs_oracle(ops, syn):
    """BASELINE config 4's front end at its stated size: a synthetic 10-sweep nuScenes cloud (~293 k points) on the all.pp.largea
    grid (0.25 m pillars, 60 points per pillar, cap 30 000; all.pp.largea.config:6-15): pillars, slot order and contents bit-exact vs
    the sequential oracle (25-27 k pillars, near ones holding hundreds of points); the PillarFeatureNet kernel on those pillars vs the
    oracle (1e-4); the pseudo image bit-exact."""
    from second_amd.models import ALL_PP_LARGEA as C
    rng_, vs = C["point_cloud_range"], C["voxel_size"]
    cloud = syn.syn_nusc_cloud(5, 293000, tuple(rng_), scene="urban")
    res = _check_voxelize(ops, [cloud], vs, rng_, C["max_points_per_voxel"], C["max_voxels"], "break")
    p = res["voxel_num"]
    assert 20000 < p <= C["max_voxels"] and int(res["num_points_per_voxel"].max()) == C["max_points_per_voxel"]
    vox, npts, coords = res["voxels"][:p].contiguous(), res["num_points_per_voxel"][:p].contiguous(), res["coordinates"][:p].contiguous()
    rng = np.random.default_rng(3)
    wt = (rng.standard_normal((9, 64)) / 3).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, 64).astype(np.float32), rng.uniform(-0.3, 0.3, 64).astype(np.float32)
    xo, yo = vs[0] / 2 + rng_[0], vs[1] / 2 + rng_[1]
    ref = orc.pfn_forward(vox.cpu().numpy(), npts.cpu().numpy(), coords.cpu().numpy(), wt, sc, sh, vs[0], vs[1], xo, yo)
    out = ops.pfn_forward(vox, npts, coords, dev(wt), dev(sc), dev(sh), vs[0], vs[1], xo, yo)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    # the same from the voxeliser's point lists (no pillar tensor): bit-identical, with and without the tensor having been written
    pts, offs = syn.batch_clouds([cloud])
    pts, offs = dev(pts), dev(offs)
    for fill in (False, True):
        v2 = ops.voxelize(pts, offs, rng_, vs, C["max_points_per_voxel"], C["max_voxels"], "break", fill=fill)
        assert (v2["voxels"] is None) == (not fill) and v2["voxel_num"] == p
        assert torch.equal(v2["num_points_per_voxel"], npts) and torch.equal(v2["coordinates"], coords)
        assert torch.equal(ops.pfn_forward_slots(pts, v2, dev(wt), dev(sc), dev(sh), vs[0], vs[1], xo, yo), out)
    ny, nx = int(round((rng_[4] - rng_[1]) / vs[1])), int(round((rng_[3] - rng_[0]) / vs[0]))
    img = ops.pillar_scatter(out, coords, 1, ny, nx)
    np.testing.assert_array_equal(img.cpu().numpy(), orc.pillar_scatter(out.cpu().numpy(), coords.cpu().numpy(), 1, ny, nx))


def test_block_filter_vs_oracle(ops):
    rng = np.random.default_rng(1)
    rng_, vs = [-10, -10, -3, 10, 10, 1], [0.1, 0.1, 0.2]
    clouds = []
    for b in range(3):   # flat ground + a few tall clusters
        ground = np.concatenate([rng.uniform(-10, 10, (4000, 2)), rng.normal(-1.7, 0.02, (4000, 1))], 1)
        tall = np.concatenate([rng.normal(0, 0.4, (1500, 2)) + rng.uniform(-8, 8, (1, 2)), rng.uniform(-1.7, 0.5, (1500, 1))], 1)
        pts = np.concatenate([ground, tall])
        rng.shuffle(pts)
        clouds.append(np.concatenate([pts, rng.uniform(0, 1, (len(pts), 1))], 1).astype(np.float32))
    from second_amd.synthetic import batch_clouds
    pts, offs = batch_clouds(clouds)
    vox = ops.voxelize(dev(pts), dev(offs), rng_, vs, 2, 20000, sync=False)
    out = ops.voxel_block_filter(vox, [200, 200], 1, 8, 0.2, 3.0)
    ooff = out["voxel_offsets"].cpu().numpy()
    kept_any = False
    for b, c in enumerate(clouds):
        r = orc.points_to_voxel(c, vs, rng_, 2, 20000)
        keep = orc.block_filter(r["voxels"], r["coordinates"], r["num_points_per_voxel"], [200, 200], 1, 8, 0.2, 3.0)
        lo, hi = ooff[b], ooff[b + 1]
        assert hi - lo == keep.sum(), (b, hi - lo, keep.sum())
        np.testing.assert_array_equal(out["coordinates"][lo:hi, 1:].cpu().numpy(), r["coordinates"][keep])
        np.testing.assert_array_equal(out["voxels"][lo:hi].cpu().numpy(), r["voxels"][keep])
        np.testing.assert_array_equal(out["num_points_per_voxel"][lo:hi].cpu().numpy(), r["num_points_per_voxel"][keep])
        kept_any |= 0 < keep.sum() < len(keep)
    assert kept_any


def test_voxelize_and_block_filter_at_nuscenes_fhd_size_vs_oracle(ops, syn):
    """BASELINE config 5's front end at its stated size -- two synthetic 10-sweep nuScenes clouds of ~293 k points on the
    all.fhd grid (0.05 x 0.05 x 0.2 m, one point per voxel, cap 90 000 voxels per frame, hit by both frames; nuscenes/all.fhd.config:6-12)
    -- voxel order, coordinates, contents and the block-filtered subset bit-exact against the sequential oracle, both cap modes."""
    from second_amd.models import ALL_FHD_NUSC as C
    rng_, vs = C["point_cloud_range"], C["voxel_size"]
    clouds = [syn.syn_nusc_cloud(s, 293000, tuple(rng_), scene="urban") for s in (3, 4)]
    for mode in ("break", "continue"):
        res = _check_voxelize(ops, clouds, vs, rng_, 1, C["max_voxels"], mode)
        assert res["voxel_num"] == 2 * C["max_voxels"]
    pts, offs = syn.batch_clouds(clouds)
    bf = C["block_filtering"]
    grid_xy = [int(round((rng_[3] - rng_[0]) / vs[0])), int(round((rng_[4] - rng_[1]) / vs[1]))]
    vox = ops.voxelize(dev(pts), dev(offs), rng_, vs, 1, C["max_voxels"], sync=False)
    out = ops.voxel_block_filter(vox, grid_xy, bf["block_factor"], bf["block_size"], bf["height_threshold"], 3.0)
    ooff = out["voxel_offsets"].cpu().numpy()
    for b, c in enumerate(clouds):
        r = orc.points_to_voxel(c, vs, rng_, 1, C["max_voxels"])
        keep = orc.block_filter(r["voxels"], r["coordinates"], r["num_points_per_voxel"], grid_xy, bf["block_factor"], bf["block_size"],
                                bf["height_threshold"], 3.0)
        lo, hi = ooff[b], ooff[b + 1]
        assert hi - lo == keep.sum() and 0.5 * len(keep) < keep.sum() < len(keep)
        np.testing.assert_array_equal(out["coordinates"][lo:hi, 1:].cpu().numpy(), r["coordinates"][keep])
        np.testing.assert_array_equal(out["voxels"][lo:hi].cpu().numpy(), r["voxels"][keep])


def test_pointpillars_detector_runs_fused_equals_module_path(syn):
    from second_amd.models import SecondDetector, ALL_PP_LARGEA
    torch.manual_seed(0)
    det = SecondDetector(ALL_PP_LARGEA).cuda().eval()
    cloud = syn.syn_nusc_cloud(0, num_points=60000, point_cloud_range=(-50, -50, -5, 50, 50, 3))
    pts, offs = syn.batch_clouds([cloud, cloud[::3]])
    pts, offs = dev(pts), dev(offs)
    with torch.no_grad():
        out = det.forward_points(pts, offs)                    # fused PFN kernel + scatter kernel, fp32
        vox = det.voxel_generator.generate_device(pts, offs)
    with torch.enable_grad():                                  # torch formulation of the PFN (training path)
        feats_t = det.voxel_feature_extractor(vox["voxels"], vox["num_points_per_voxel"], vox["coordinates"])
    with torch.no_grad():
        feats_k = det.voxel_feature_extractor(vox["voxels"], vox["num_points_per_voxel"], vox["coordinates"])
    np.testing.assert_allclose(feats_k.cpu().numpy(), feats_t.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    assert out["boxes"].shape[0] == 2 and out["valid"].any()
    assert int(out["labels"].max()) < 10
    # static-capacity pillars (device-side pillar count) and its hipGraph replay == the eager path
    with torch.no_grad():
        st = det.forward_points(pts, offs, static=True)
        replay, g = det.make_graphed(pts, offs)
        replay()
        torch.cuda.synchronize()
    for other in (st, g):
        assert torch.equal(out["valid"], other["valid"])
        m = out["valid"]
        assert torch.equal(out["scores"][m], other["scores"][m]) and torch.equal(out["boxes"][m], other["boxes"][m])


@pytest.mark.parametrize("cin,cout,k,stride,pad,hw", [(128, 128, 3, 1, 1, (200, 176)), (64, 64, 3, 2, 1, (37, 29)),
                                                     (128, 256, 3, 2, 1, (50, 50)), (128, 64, 1, 1, 0, (33, 17)),
                                                     (128, 128, 1, 1, 0, (61, 43)),
                                                     (64, 128, 3, 1, 1, (9, 7)),
                                                     # the halo kernel's border handling: ragged right / bottom edges, a single tile, one pixel
                                                     # over a tile, two output-channel blocks
                                                     (128, 128, 3, 1, 1, (37, 29)), (128, 128, 3, 1, 1, (8, 16)), (128, 128, 3, 1, 1, (9, 17)),
                                                     (128, 256, 3, 1, 1, (23, 40)), (128, 128, 3, 1, 1, (1, 1)),
                                                     (256, 256, 3, 1, 1, (50, 50)), (256, 128, 3, 1, 1, (7, 19)),    # 4 x 16 tiles, 256 input channels
                                                     (64, 64, 3, 1, 1, (200, 200)), (64, 64, 3, 1, 1, (13, 21)),      # 64 output channels per workgroup
                                                     # k_conv2d_patch (PointPillars RPN: stride-2 first convs, k == stride deblocks, 1x1 over 256 / 384 channels):
                                                     # full-size and ragged maps, odd sizes (the last tap row / column falls into the padding or is dropped)
                                                     (64, 64, 3, 2, 1, (400, 400)), (64, 64, 3, 2, 1, (101, 77)), (64, 128, 3, 2, 1, (200, 200)),
                                                     (64, 128, 3, 2, 1, (50, 37)), (128, 256, 3, 2, 1, (100, 100)), (128, 128, 3, 2, 1, (23, 18)),
                                                     (64, 128, 4, 4, 0, (200, 200)), (64, 128, 4, 4, 0, (37, 53)), (128, 128, 2, 2, 0, (100, 100)),
                                                     (128, 128, 2, 2, 0, (31, 45)), (256, 128, 1, 1, 0, (50, 50)), (384, 256, 1, 1, 0, (50, 50)),
                                                     (384, 128, 1, 1, 0, (7, 5)), (256, 256, 1, 1, 0, (3, 33))])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv2d_nhwc_mfma_vs_torch(ops, cin, cout, k, stride, pad, hw, dtype):
    torch.manual_seed(cin + cout + k)
    b = 2
    x = torch.randn(b, cin, *hw, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5).to(dtype)
    bias = torch.randn(cout, device="cuda")
    ref = torch.relu(torch.nn.functional.conv2d(x.float(), w.float(), bias, stride, pad))
    out = ops.conv2d_nhwc(x, ops.conv2d_pack_weight(w), bias, cout, k, stride, pad, relu=True)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -9
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.cpu().numpy(), rtol=tol, atol=tol * ref.abs().max().item())
    nob = ops.conv2d_nhwc(x, ops.conv2d_pack_weight(w), None, cout, k, stride, pad, relu=False)
    ref2 = torch.nn.functional.conv2d(x.float(), w.float(), None, stride, pad)
    np.testing.assert_allclose(nob.float().cpu().numpy(), ref2.cpu().numpy(), rtol=tol, atol=tol * ref2.abs().max().item())
    # PER ELEMENT (the lines above are relative to the tensor's maximum): exact 16-bit products, fp32 accumulation of K = cin k^2 terms
    # on both sides (gamma_K * sum |x| |w| of that output element each), one rounding to the output dtype
    mag = torch.nn.functional.conv2d(x.float().abs(), w.float().abs(), None, stride, pad)
    u_out = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    bound = u_out * ref2.abs() + 2 * (cin * k * k + 2) * 2.0 ** -24 * mag + 1e-30
    assert bool(((nob.float() - ref2).abs() <= bound).all()), float(((nob.float() - ref2).abs() / bound).max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("fill", ["clusters", "empty", "negzero"])
def test_conv2d_sparse_input_tile_skip_is_bit_identical(ops, dtype, fill):
    """Flag bit 1 of sec_conv2d_nhwc (first RPN layer, input = SparseConvTensor.dense()): all-zero 10x18 input halos write
    act(bias) without the MFMA loop.  Must equal the unflagged launch bit for bit, also across tile borders, for an
    entirely empty input and for -0.0 inputs (treated as live)."""
    torch.manual_seed(5)
    b, c, h, w = 2, 128, 50, 70                       # ragged against the 8 x 16 tile
    x = torch.zeros(b, c, h, w, device="cuda")
    if fill != "empty":
        for (bi, y0, x0) in [(0, 0, 0), (0, 7, 15), (0, 8, 16), (1, 23, 47), (1, 49, 69), (1, 31, 32)]:   # tile corners / borders
            x[bi, :, y0:y0 + 2, x0:x0 + 1] = torch.randn(c, min(2, h - y0), 1, device="cuda")
    if fill == "negzero":
        x[0, 3, 40, 5] = -0.0
    x = x.to(dtype).contiguous(memory_format=torch.channels_last)
    wgt = (torch.randn(128, c, 3, 3, device="cuda") / 34).to(dtype)
    bias = torch.randn(128, device="cuda")
    pk = ops.conv2d_pack_weight(wgt)
    for relu in (True, False):
        plain = ops.conv2d_nhwc(x, pk, bias, 128, 3, 1, 1, relu=relu)
        skip = ops.conv2d_nhwc(x, pk, bias, 128, 3, 1, 1, relu=relu, sparse_input=True)
        assert torch.equal(plain.view(torch.int16), skip.view(torch.int16))
    ref = torch.relu(torch.nn.functional.conv2d(x.float(), wgt.float(), bias, 1, 1))
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -9
    out = ops.conv2d_nhwc(x, pk, bias, 128, 3, 1, 1, relu=True, sparse_input=True)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.cpu().numpy(), rtol=tol, atol=tol * ref.abs().max().item())


@pytest.mark.parametrize("cfg_name", ["car.fhd", "pp", "nusc.fhd"])
def test_fused_predict_matches_torch_formulation(cfg_name):
    """select / decode / NMS / finalize kernels vs the torch restatement of voxelnet.py:377-645 (distinct scores)."""
    from second_amd.models import SecondDetector, CAR_FHD, ALL_PP_LARGEA, ALL_FHD_NUSC
    cfg = {"car.fhd": CAR_FHD, "pp": ALL_PP_LARGEA, "nusc.fhd": ALL_FHD_NUSC}[cfg_name]
    det = SecondDetector(cfg).cuda().eval()
    a = det.rpn._num_anchor_per_loc
    _, h, w = det.feature_map_size
    b, nc = 3, cfg["num_class"]
    g = torch.Generator(device="cuda").manual_seed(0)
    preds = {"cls_preds": torch.randn(b, a, h, w, nc, device="cuda", generator=g) * 0.8 - (1.0 if nc == 1 else 3.0),
             "box_preds": torch.randn(b, a, h, w, 7, device="cuda", generator=g) * 0.2,
             "dir_cls_preds": torch.randn(b, a, h, w, 2, device="cuda", generator=g)}
    with torch.no_grad():
        det.fused_predict = False
        ref = det.predict_device({k: v.clone() for k, v in preds.items()}, b)
        det.fused_predict = True
        out = det.predict_device(preds, b)
        # strided (non-contiguous) views of a packed head tensor, bf16
        packed = torch.cat([preds["box_preds"].permute(0, 1, 4, 2, 3).reshape(b, a * 7, h, w),
                            preds["cls_preds"].permute(0, 1, 4, 2, 3).reshape(b, a * nc, h, w),
                            preds["dir_cls_preds"].permute(0, 1, 4, 2, 3).reshape(b, a * 2, h, w)], 1)
        packed = packed.bfloat16().contiguous(memory_format=torch.channels_last)
        c0, views = 0, {}
        for name, code in (("box_preds", 7), ("cls_preds", nc), ("dir_cls_preds", 2)):
            views[name] = packed[:, c0:c0 + a * code].reshape(b, a, code, h, w).permute(0, 1, 3, 4, 2)
            c0 += a * code
        out_v = det.predict_device(views, b)
        det.fused_predict = False
        ref_v = det.predict_device({k: v.float().contiguous() for k, v in views.items()}, b)
    for o, r in ((out, ref), (out_v, ref_v)):
        assert torch.equal(o["valid"], r["valid"]) and o["valid"].any()
        m = r["valid"]
        np.testing.assert_allclose(o["scores"][m].cpu().numpy(), r["scores"][m].cpu().numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(o["boxes"][m].cpu().numpy(), r["boxes"][m].cpu().numpy(), rtol=1e-5, atol=1e-5)
        assert torch.equal(o["labels"][m].long(), r["labels"][m].long())


def test_nuscenes_fhd_detector_static_and_graph_match_eager(syn):
    """BASELINE config 5 on the device path: block-filtered voxels -> SpMiddleFHD (1984x1984x40) -> RPN with a
    stride-2 "upsample" conv -> 10-class axis-aligned NMS; eager == static-capacity == hipGraph replay."""
    from second_amd.models import SecondDetector, ALL_FHD_NUSC
    torch.manual_seed(0)
    det = SecondDetector(ALL_FHD_NUSC).cuda().prepare_inference(torch.bfloat16)
    assert type(det.rpn).__name__ == "RPNInference" and det.rpn.use_hip
    clouds = [syn.syn_nusc_cloud(s, num_points=30000, point_cloud_range=(-49.6, -49.6, -5, 49.6, 49.6, 3)) for s in range(2)]
    pts, offs = syn.batch_clouds(clouds)
    pts, offs = dev(pts), dev(offs)
    with torch.no_grad():
        e = det.forward_points(pts, offs)
        det.calibrate(pts, offs)
        s = det.forward_points(pts, offs, static=True)
        det.check_overflow()
        replay, g = det.make_graphed(pts, offs)
        replay()
        torch.cuda.synchronize()
    assert e["boxes"].shape[:2] == (2, 300)
    for other in (s, g):
        assert torch.equal(e["valid"], other["valid"])
        m = e["valid"]
        assert torch.equal(e["scores"][m], other["scores"][m])
        assert torch.equal(e["boxes"][m], other["boxes"][m])
        assert torch.equal(e["labels"][m], other["labels"][m])


def test_sparse_sequential_training_step_vs_oracle(ops):
    """Training path through the drop-in modules (train-mode BatchNorm1d, autograd through IndiceConvFunction,
    rulebook reuse via indice_key): loss.backward() on the GPU vs the same modules over the CPU oracle."""
    import copy
    import spconv
    import oracle_backend
    torch.manual_seed(0)
    rng = np.random.RandomState(0)
    shape = [9, 24, 20]
    n = 700
    flat = rng.choice(2 * shape[0] * shape[1] * shape[2], n, replace=False)
    b, r = np.divmod(flat, shape[0] * shape[1] * shape[2])
    z, r = np.divmod(r, shape[1] * shape[2])
    y, x = np.divmod(r, shape[2])
    idx = np.stack([b, z, y, x], 1).astype(np.int32)
    idx = idx[np.argsort(b, kind="stable")]
    feats = rng.randn(n, 16).astype(np.float32)

    def bn(c):
        return torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)
    net = spconv.SparseSequential(
        spconv.SubMConv3d(16, 32, 3, bias=False, indice_key="subm0"), bn(32), torch.nn.ReLU(),
        spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="subm0"), bn(32), torch.nn.ReLU(),
        spconv.SparseConv3d(32, 64, 3, 2, padding=1, bias=False), bn(64), torch.nn.ReLU(),
        spconv.SparseConv3d(64, 64, (3, 1, 1), (2, 1, 1), bias=True), torch.nn.ReLU())
    net_cpu = copy.deepcopy(net)

    def step(model, device):
        model.train()
        f = torch.from_numpy(feats).to(device).requires_grad_(True)
        t = spconv.SparseConvTensor(f, torch.from_numpy(idx).to(device), shape, 2)
        out = model(t).dense()
        w = torch.linspace(-1, 1, out.numel(), device=device).view_as(out)
        loss = (out * w).sum() + (out ** 2).mean()
        loss.backward()
        grads = {k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters()}
        return loss.item(), f.grad.cpu().numpy(), grads, {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}

    loss_g, df_g, grads_g, state_g = step(net.cuda(), "cuda")
    with oracle_backend.installed():
        loss_c, df_c, grads_c, state_c = step(net_cpu, "cpu")
    assert abs(loss_g - loss_c) <= 1e-4 * max(1.0, abs(loss_c))
    np.testing.assert_allclose(df_g, df_c, rtol=1e-3, atol=1e-4 * np.abs(df_c).max())
    for k in grads_c:
        np.testing.assert_allclose(grads_g[k], grads_c[k], rtol=1e-3, atol=2e-4 * np.abs(grads_c[k]).max(), err_msg=k)
    for k in state_c:   # BatchNorm running statistics updated identically
        np.testing.assert_allclose(state_g[k], state_c[k], rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("cout2", [64, 128])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv1x1_chain_vs_torch(ops, cout2, dtype):
    """Fused deblock + heads kernel vs two torch 1x1 convs (the intermediate rounded to the 16-bit type, as the
    unfused path stores it)."""
    torch.manual_seed(cout2)
    b, hw = 2, (37, 29)                                  # 2146 pixels: not a multiple of the 128-pixel tile
    x = torch.randn(b, 128, *hw, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    w1 = (torch.randn(128, 128, 1, 1, device="cuda") / 128 ** 0.5).to(dtype)
    w2 = (torch.randn(cout2, 128, 1, 1, device="cuda") / 128 ** 0.5).to(dtype)
    b1, b2 = torch.randn(128, device="cuda"), torch.randn(cout2, device="cuda")
    h = torch.relu(torch.nn.functional.conv2d(x.float(), w1.float(), b1)).to(dtype).float()
    ref = torch.nn.functional.conv2d(h, w2.float(), b2)
    out = ops.conv1x1_chain(x, ops.conv2d_pack_weight(w1), b1, ops.conv2d_pack_weight(w2), b2, cout2)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -9
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.cpu().numpy(), rtol=tol, atol=tol * ref.abs().max().item())
    sep = ops.conv2d_nhwc(ops.conv2d_nhwc(x, ops.conv2d_pack_weight(w1), b1, 128, 1, 1, 0, relu=True),
                          ops.conv2d_pack_weight(w2), b2, cout2, 1, 1, 0, relu=False)
    np.testing.assert_allclose(out.float().cpu().numpy(), sep.float().cpu().numpy(), rtol=tol, atol=tol * ref.abs().max().item())


def test_detector_empty_and_tiny_frames(syn):
    """Ragged batch: an EMPTY frame between two real ones, and a frame with three points -- eager and static paths;
    the real frames' detections do not depend on their neighbours (frames are independent)."""
    from second_amd.models import SecondDetector, CAR_FHD
    torch.manual_seed(0)
    det = SecondDetector(CAR_FHD).cuda().prepare_inference(torch.bfloat16)
    a, c = syn.syn_kitti_cloud(0, num_points=9000, num_voxels=8000), syn.syn_kitti_cloud(1, num_points=9000, num_voxels=8000)
    empty = np.zeros((0, 4), np.float32)
    tiny = a[:3].copy()

    def run(clouds, static):
        pts, offs = syn.batch_clouds(clouds)
        with torch.no_grad():
            out = det.forward_points(dev(pts), dev(offs), static=static)
            if static:
                det.check_overflow()
        return {k: v.cpu() for k, v in out.items()}

    ref = run([a, c], False)
    for static in (False, True):
        out = run([a, empty, c, tiny], static)
        # (the empty frame may well "detect" something: an all-zero RPN input still passes through the biases)
        for src, dst in ((0, 0), (1, 2)):
            m = ref["valid"][src]
            assert torch.equal(m, out["valid"][dst])
            assert torch.equal(ref["scores"][src][m], out["scores"][dst][m])
            assert torch.equal(ref["boxes"][src][m], out["boxes"][dst][m])
    only_empty = run([empty], False)                         # must simply run
    assert only_empty["valid"].shape[0] == 1


def test_detector_branched_graph_matches_eager(syn):
    """make_graphed(branches=2): two half-batch chains on two streams of one hipGraph == the eager per-frame results."""
    from second_amd.models import SecondDetector, CAR_FHD
    torch.manual_seed(0)
    det = SecondDetector(CAR_FHD).cuda().prepare_inference(torch.bfloat16)
    clouds = [syn.syn_kitti_cloud(s, num_points=6000 + 700 * s, num_voxels=5000 + 500 * s) for s in range(5)]
    pts, offs = syn.batch_clouds(clouds)
    pts, offs = dev(pts), dev(offs)
    with torch.no_grad():
        e = det.forward_points(pts, offs)
        replay, outs, parts = det.make_graphed(pts, offs, branches=2)
        replay()
        replay()
        torch.cuda.synchronize()
        det.check_overflow()
    assert len(outs) == 2 and sum(o["valid"].shape[0] for o in outs) == 5
    got = {k: torch.cat([o[k] for o in outs]) for k in ("valid", "scores", "boxes")}
    assert torch.equal(e["valid"], got["valid"])
    m = e["valid"]
    assert torch.equal(e["scores"][m], got["scores"][m]) and torch.equal(e["boxes"][m], got["boxes"][m])


def test_detector_steps_in_flight(syn):
    """InFlightRunner: captured forwards with their own activation buffers replayed concurrently on separate streams (the
    bench's --inflight pipelining) leave each other's results intact and equal the eager path."""
    from second_amd.models import SecondDetector, CAR_FHD, InFlightRunner
    torch.manual_seed(0)
    det = SecondDetector(CAR_FHD).cuda().prepare_inference(torch.bfloat16)
    pts, offs = syn.batch_clouds([syn.syn_kitti_cloud(s, num_points=7000, num_voxels=6000) for s in range(3)])
    pts, offs = dev(pts), dev(offs)
    with torch.no_grad():
        e = det.forward_points(pts, offs)
        det.calibrate(pts, offs)
        runner = InFlightRunner(det, pts, offs, inflight=3)
        seen = []
        for _ in range(7):
            out, stream = runner.step()
            seen.append(out)
        runner.synchronize()
    assert len({id(o) for o in seen}) == 3
    m = e["valid"]
    for o in runner.outputs:
        assert torch.equal(m, o["valid"])
        assert torch.equal(e["scores"][m], o["scores"][m]) and torch.equal(e["boxes"][m], o["boxes"][m])


def test_detector_steps_in_flight_with_serialised_rpn_segments(syn):
    """InFlightRunner(serialize_rpn=True): a step is three graphs sharing one memory pool (sparse front / RPN / predict) and the
    lanes pass an event from RPN segment to RPN segment -- the bench's default serving form.  Every lane's detections equal the
    eager path's bit for bit, with the clouds fed through per-lane input buffers too."""
    from second_amd.models import SecondDetector, CAR_FHD, InFlightRunner
    torch.manual_seed(0)
    det = SecondDetector(CAR_FHD).cuda().prepare_inference(torch.bfloat16)
    pts, offs = syn.batch_clouds([syn.syn_kitti_cloud(s, num_points=7000, num_voxels=6000) for s in range(3)])
    pts, offs = dev(pts), dev(offs)
    with torch.no_grad():
        e = det.forward_points(pts, offs)
        det.calibrate(pts, offs)
        for private in (False, True):
            runner = InFlightRunner(det, pts, offs, inflight=4, serialize_rpn=True, private_inputs=private)
            assert runner.serialize_rpn and isinstance(runner.replays[0], tuple) and len(runner.replays[0]) == 3
            # the runner's token rule: two RPN segments at a time up to 60 % live tiles in the last conv of the calibration scene, else one
            share = float(det.rpn.last_live_counts[det.rpn.background_convs - 1].sum()) / (det.rpn.last_tiles_per_frame * 3)
            assert runner.rpn_tokens == (2 if share <= 0.6 else 1), (runner.rpn_tokens, share)
            hp, ho = (pts.cpu().pin_memory(), offs.cpu().pin_memory()) if private else (None, None)
            for _ in range(11):
                runner.step(hp, ho, fetch=private)
            runner.synchronize()
            m = e["valid"]
            for o in runner.outputs:
                assert torch.equal(m, o["valid"])
                assert torch.equal(e["scores"][m], o["scores"][m]) and torch.equal(e["boxes"][m], o["boxes"][m])
            if private:
                for h in runner.host_outputs:
                    assert torch.equal(h["valid"], m.cpu()) and torch.equal(h["boxes"][m.cpu()], e["boxes"][m].cpu())
        assert InFlightRunner(det, pts, offs, inflight=2, serialize_rpn=True, rpn_tokens=3).rpn_tokens == 3        # an explicit count is kept


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("levels,k", [(3, 1000), (40, 1000), (700, 1000), (5, 64), (1, 1000), (100000, 1000)])
def test_predict_select_tie_ranking(levels, k, dtype):
    """Heavily tied 16-bit logits (an untrained or saturated head): the selection is the k largest keys, ties by ascending
    anchor index (the order torch.topk of voxelnet.py:551-570 is replaced by, documented in DESIGN), identical run to run.
    levels = number of distinct logit values per frame (1 = every anchor ties; 100000 = practically distinct)."""
    from second_amd import ops
    b, a, h, w = 3, 2, 200, 176
    n = a * h * w
    g = torch.Generator(device="cuda").manual_seed(levels)
    vals = (torch.randn(levels, device="cuda", generator=g) * 2.0 - 1.0).to(dtype)
    pick = torch.randint(0, levels, (b, n), device="cuda", generator=g)
    cls = vals[pick].reshape(b, a, h, w, 1).contiguous()
    top_idx, top_score, _, counts = ops.predict_select(cls, k, 0.3)
    again = ops.predict_select(cls, k, 0.3)
    assert torch.equal(counts, again[3])
    flat = cls.reshape(b, n).float()
    # reference: stable sort by descending logit -> ties keep ascending index; voxelnet.py:551-570 masks by the score threshold
    # BEFORE its topk, so the selection is the first counts[b] entries (what lies behind them is unspecified)
    order = torch.sort(flat, dim=1, descending=True, stable=True).indices[:, :k]
    ref_scores = torch.sigmoid(torch.gather(flat, 1, order))
    assert torch.equal(counts.long(), (ref_scores >= 0.3).sum(1))
    for f in range(b):
        c = int(counts[f])
        assert torch.equal(top_idx[f, :c], again[0][f, :c]) and torch.equal(top_score[f, :c], again[1][f, :c])
        assert torch.equal(top_idx[f, :c].long(), order[f, :c]), (levels, k, f)
        np.testing.assert_allclose(top_score[f, :c].cpu().numpy(), ref_scores[f, :c].cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("dtype,a,h,w", [(torch.bfloat16, 2, 200, 176), (torch.float16, 2, 200, 176), (torch.float16, 20, 248, 248),
                                         (torch.bfloat16, 12, 200, 200), (torch.float32, 2, 200, 176)])
@pytest.mark.parametrize("thr", [0.05, 0.3, 0.5, 0.9])
def test_predict_select_threshold_shortcut(thr, dtype, a, h, w):
    """A trained-like head: a few hundred anchors per frame above the score threshold, logits spread around it (some within one
    bf16 step of the threshold's logit, some exactly on it).  The selection with the threshold shortcut (no bisection when at most
    k keys reach the threshold's conservative 16-bit key; csrc/predict.hip) equals the stable-sort reference entry for entry up
    to counts[b], for a frame with fewer than k candidates, one with none and one with far more than k -- for bf16 and fp16 heads
    (fp16 keys are the half's own bits), for heads large enough to take the 73 728-anchor chunks, and for fp32 heads (the 32-bit
    radix select's shortcut on a conservative 32-bit threshold key, round 5)."""
    from second_amd import ops
    b = 3                                           # (a, h, w): car.fhd's head, nuScenes all.fhd's (1.23 M anchors: 73 728-anchor
    n = a * h * w                                   # chunks), a 480 k-anchor head (59 chunks of 8192); bf16 and fp16 keys
    g = torch.Generator(device="cuda").manual_seed(int(thr * 100))
    lt = float(np.log(thr / (1 - thr)))
    cls = torch.full((b, n), lt - 6.0, device="cuda")
    cls += torch.rand(b, n, device="cuda", generator=g)                      # background: well below the threshold, distinct-ish
    hot = torch.randperm(n, device="cuda", generator=g)[:700]
    cls[0, hot] = lt + torch.randn(700, device="cuda", generator=g) * 0.05    # ~350 above, many within a bf16 step of the threshold
    cls[0, hot[:40]] = lt                                                    # exactly the threshold's logit (before bf16 rounding)
    many = torch.randperm(n, device="cuda", generator=g)[:5000]
    cls[2, many] = lt + 0.5 + torch.rand(5000, device="cuda", generator=g) * 4   # more than k candidates: bisection path
    cls = cls.to(dtype).reshape(b, a, h, w, 1).contiguous()
    k = 1000
    top_idx, top_score, _, counts = ops.predict_select(cls, k, thr)
    flat = cls.reshape(b, n).float()
    order = torch.sort(flat, dim=1, descending=True, stable=True).indices[:, :k]
    ref_scores = torch.sigmoid(torch.gather(flat, 1, order))
    ref_counts = (ref_scores >= thr).sum(1)
    assert torch.equal(counts.long(), ref_counts), (counts, ref_counts)
    assert 100 < int(counts[0]) < 700 and int(counts[1]) == 0 and int(counts[2]) == k
    for f in range(b):
        c = int(counts[f])
        assert torch.equal(top_idx[f, :c].long(), order[f, :c])
        np.testing.assert_allclose(top_score[f, :c].cpu().numpy(), ref_scores[f, :c].cpu().numpy(), rtol=1e-6)


def test_detector_sorted_and_first_touch_numbering_give_identical_results(syn):
    """The device fast path numbers the strided layers' outputs the spconv-GPU way ("sorted", no hash table); the row order is
    internal: the dense RPN input and the detections are bit-identical to the first-touch (spconv-CPU / oracle) numbering, in
    eager, static-capacity and bf16 inference form."""
    from second_amd import ops
    from second_amd.models import SecondDetector, CAR_FHD
    torch.manual_seed(0)
    det = SecondDetector(CAR_FHD).cuda().eval()
    pts, offs = syn.batch_clouds([syn.syn_kitti_cloud(s, num_points=9000, num_voxels=8000) for s in range(3)])
    pts, offs = dev(pts), dev(offs)
    res = {}
    with torch.no_grad():
        vox = det.voxel_generator.generate_device(pts, offs, mean_features=4)
        for mode in ("first_touch", "sorted"):
            prev = ops.set_rulebook_numbering(mode)
            try:
                res["dense", mode] = det.middle_feature_extractor(vox["mean"], vox["coordinates"], 3)
            finally:
                ops.set_rulebook_numbering(prev)
            det.rulebook_numbering = mode
            res["eager", mode] = det.forward_points(pts, offs)
            res["static", mode] = det.forward_points(pts, offs, static=True)
        det.prepare_inference(torch.bfloat16)
        for mode in ("first_touch", "sorted"):
            det.rulebook_numbering = mode
            res["bf16", mode] = det.forward_points(pts, offs, static=True)
    assert torch.equal(res["dense", "first_touch"], res["dense", "sorted"])
    for form in ("eager", "static", "bf16"):
        a, b = res[form, "first_touch"], res[form, "sorted"]
        assert a["valid"].any() and torch.equal(a["valid"], b["valid"])
        m = a["valid"]
        assert torch.equal(a["scores"][m], b["scores"][m]) and torch.equal(a["boxes"][m], b["boxes"][m])


def test_rulebook_subm_after_voxelize_reuses_the_voxel_hash_table(ops, syn):
    """First SubM layer of SpMiddleFHD: site lookup in the hash table the voxeliser left in its workspace (grid z = 40, sparse
    shape z = 41) == the stand-alone build; eager (sliced rows) and static-capacity (device row count) forms, two frames."""
    pts, offs = syn.batch_clouds([syn.syn_kitti_cloud(s, num_points=9000, num_voxels=8000) for s in range(2)])
    for sync in (True, False):
        vox = ops.voxelize(dev(pts), dev(offs), syn.CAR_FHD_RANGE, syn.CAR_FHD_VOXEL, 5, 40000, mean_features=4, sync=sync)
        assert vox["site_table"][0] == "vox" and vox["site_table"][5] == [40, 1600, 1408]
        idx = vox["coordinates"]
        nd = None if sync else vox["voxel_offsets"][2:]
        plain = ops.rulebook_subm(idx.clone(), 2, [41, 1600, 1408], 3, 1, n_dev=nd)
        reuse = ops.rulebook_subm(idx, 2, [41, 1600, 1408], 3, 1, n_dev=nd, site_table=vox["site_table"])
        live = int(vox["voxel_offsets"][2].item())
        assert live == 16000 and torch.equal(reuse["nbr_out"][:live], plain["nbr_out"][:live])
        assert (reuse["nbr_out"][:live, 13] == torch.arange(live, device="cuda", dtype=torch.int32)).all()


@pytest.mark.parametrize("cout,hw,npil", [(64, (400, 400), 60000), (64, (37, 53), 300), (128, (50, 70), 0), (128, (96, 40), 900)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv2d_nhwc_rows_equals_pillar_scatter_then_conv(ops, cout, hw, npil, dtype):
    """sec_conv2d_nhwc_rows (PointPillarsScatter + the first RPN conv without the canvas; pointpillars.py:444-476, rpn.py:484-486): bit-identical
    to sec_pillar_scatter followed by sec_conv2d_nhwc, including empty tiles (act(bias)), an empty frame, rows past the live count (static
    capacity: the device-side pillar count) and ragged map sizes."""
    torch.manual_seed(cout + npil)
    b, (h, w), cap = (4 if hw == (400, 400) else 3), hw, npil + 37     # 4 x 400 x 400: the several-rounds launch (piece table form of the kernel)
    cells = torch.randperm(2 * h * w, device="cuda")[:npil]            # frames 0 and 1 hold pillars, frame 2 is empty
    if npil > 400:                                                       # clustered: leave whole tiles without pillars
        cells = cells[(cells % w) < w // 2]
    n = cells.numel()
    coords = torch.zeros(cap, 4, dtype=torch.int32, device="cuda")
    coords[:n, 0], coords[:n, 2], coords[:n, 3] = (cells // (h * w)).int(), ((cells // w) % h).int(), (cells % w).int()
    coords[n:] = torch.tensor([1, 0, 2, 3], dtype=torch.int32, device="cuda")          # stale rows past the live count: must not be read
    feats = torch.randn(cap, 64, device="cuda").to(dtype)
    num_dev = torch.tensor([n], dtype=torch.int32, device="cuda")
    wgt = (torch.randn(cout, 64, 3, 3, device="cuda") / 24).to(dtype)
    bias = torch.randn(cout, device="cuda")
    pk = ops.conv2d_pack_weight(wgt)
    canvas = ops.pillar_scatter(feats, coords, b, h, w, channels_last=True, num_dev=num_dev)
    smap = ops.pillar_site_map(coords, b, h, w, num_dev=num_dev)
    assert int((smap > 0).sum()) == n and int(smap.max()) <= n
    for relu in (True, False):
        ref = ops.conv2d_nhwc(canvas, pk, bias, cout, 3, 2, 1, relu=relu)
        out = ops.conv2d_nhwc_rows(feats, smap, pk, bias, cout, 3, 2, 1, relu=relu)
        assert out.shape == ref.shape and torch.equal(out.view(torch.int16), ref.view(torch.int16))
    tref = torch.relu(torch.nn.functional.conv2d(canvas.float(), wgt.float(), bias, 2, 1))
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -9
    np.testing.assert_allclose(ops.conv2d_nhwc_rows(feats, smap, pk, bias, cout, 3, 2, 1, relu=True).float().cpu().numpy(),
                               tref.cpu().numpy(), rtol=tol, atol=tol * tref.abs().max().item())


@pytest.mark.parametrize("sizes", [(150000, 0, 210000), (60000, 0, 30000)])
def test_voxelize_pillars_dense_numbering_and_staged_passes(ops, syn, sizes):
    """The pillar path of sec_voxelize_f32 (csrc/voxelize.hip: voxels == NULL, max_points > 8): dense slot numbering (k_vox_cell_first) and,
    from 256 k points, the staged first-point pass (early eighth of every cloud first, the others look before their atomicMin).  A ragged batch with an empty cloud, above and below the staging threshold, both cap modes, the voxel cap hit
    and not hit: counts and coordinates equal the oracle-checked tensor path, and the PillarFeatureNet computed through the point lists
    equals the one computed from the oracle-checked voxel tensor bit for bit (slot contents and slot order)."""
    rng_, vs = [-50, -50, -10, 50, 50, 10], [0.25, 0.25, 20]
    clouds = [syn.syn_nusc_cloud(7 + k, n, tuple(rng_), scene="urban") if n else np.zeros((0, 4), np.float32) for k, n in enumerate(sizes)]
    pts, offs = syn.batch_clouds(clouds)
    pts, offs = dev(pts), dev(offs)
    rng = np.random.default_rng(11)
    wt, sc, sh = dev((rng.standard_normal((9, 64)) / 3).astype(np.float32)), dev(rng.uniform(0.5, 1.5, 64).astype(np.float32)), dev(rng.uniform(-0.3, 0.3, 64).astype(np.float32))
    xo, yo = vs[0] / 2 + rng_[0], vs[1] / 2 + rng_[1]
    for cap_mode in ("break", "continue"):
        for max_points, max_voxels in ((60, 30000), (60, 5000), (100, 30000)):
            ref = _check_voxelize(ops, clouds, vs, rng_, max_points, max_voxels, cap_mode)       # tensor path (hash numbering) vs the oracle
            p = ref["voxel_num"]
            out = ops.pfn_forward(ref["voxels"][:p].contiguous(), ref["num_points_per_voxel"][:p].contiguous(), ref["coordinates"][:p].contiguous(),
                                  wt, sc, sh, vs[0], vs[1], xo, yo)
            v2 = ops.voxelize(pts, offs, rng_, vs, max_points, max_voxels, cap_mode, fill=False)
            assert v2["voxels"] is None and v2["voxel_num"] == p and torch.equal(v2["voxel_offsets"], ref["voxel_offsets"])
            assert torch.equal(v2["num_points_per_voxel"][:p], ref["num_points_per_voxel"][:p]) and torch.equal(v2["coordinates"][:p], ref["coordinates"][:p])
            assert torch.equal(ops.pfn_forward_slots(pts, v2, wt, sc, sh, vs[0], vs[1], xo, yo)[:p], out)
