"""Property tests of SURVEY 8(c)(ii): the only extra pins available for the spconv boundary, whose source is absent
(rulebooks / indice_conv: "parity unpinned", anchored on dense conv3d in tests/test_oracle_conv.py).

CPU part (hypothesis, small random geometries): the ORACLE satisfies the structural laws any correct spconv v1.x rulebook
obeys.  GPU part (-m gpu, BASELINE sizes): the HIP path satisfies the same laws at batch 8 x 16 000 voxels, where an
element-by-element oracle comparison of every table would be the slow part -- size-independent properties instead."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import oracle as orc

SETTINGS = dict(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


@st.composite
def sparse_sites(draw):
    shape = (draw(st.integers(2, 7)), draw(st.integers(3, 11)), draw(st.integers(3, 11)))
    batch = draw(st.integers(1, 3))
    cells = batch * shape[0] * shape[1] * shape[2]
    n = draw(st.integers(1, min(cells, 120)))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    lin = rng.choice(cells, n, replace=False)
    b, rem = np.divmod(lin, shape[0] * shape[1] * shape[2])
    z, rem = np.divmod(rem, shape[1] * shape[2])
    y, x = np.divmod(rem, shape[2])
    return np.stack([b, z, y, x], 1).astype(np.int32), batch, list(shape)


@settings(**SETTINGS)
@given(sparse_sites())
def test_subm_rulebook_laws(case):
    idx, batch, shape = case
    n = len(idx)
    out_idx, pairs, num = orc.rulebook_subm(idx, batch, shape, 3)
    np.testing.assert_array_equal(out_idx, idx)                       # submanifold: outputs ARE the inputs, same order
    assert num[13] == n                                               # the centre offset pairs every site with itself
    np.testing.assert_array_equal(pairs[13, 0, :n], pairs[13, 1, :n])
    sets = [set(map(tuple, pairs[k, :, :num[k]].T)) for k in range(27)]
    for k in range(27):                                               # (i, o) at offset k  <=>  (o, i) at offset 26 - k
        assert sets[26 - k] == {(o, i) for (i, o) in sets[k]}
        p = pairs[k, :, :num[k]]
        assert np.all(np.diff(p[0]) > 0)                              # canonical order: ascending input row within an offset
        # geometry: output = input shifted by the offset, same frame
        d = idx[p[1], 1:] - idx[p[0], 1:]
        kz, ky, kx = k // 9, (k // 3) % 3, k % 3
        assert np.all(idx[p[0], 0] == idx[p[1], 0])
        assert np.all(d == -(np.array([kz, ky, kx]) - 1)) or np.all(d == np.array([kz, ky, kx]) - 1)


@settings(**SETTINGS)
@given(sparse_sites(), st.sampled_from([(3, 2, 1), (3, 2, 0), ((3, 1, 1), (2, 1, 1), 0), (2, 2, 0), (3, 1, 1)]))
def test_strided_rulebook_laws(case, geom):
    idx, batch, shape = case
    ks, stv, pad = geom
    out_shape = orc.conv_output_size(shape, ks, stv, pad, 1)
    if min(out_shape) <= 0:
        return
    out_idx, pairs, num, oshape = orc.rulebook_conv(idx, batch, shape, ks, stv, pad)
    assert list(oshape) == list(out_shape)
    m = len(out_idx)
    assert len({tuple(r) for r in out_idx}) == m                      # every active output site appears once
    assert np.all(out_idx[:, 1:] >= 0) and np.all(out_idx[:, 1:] < np.array(out_shape))
    kvol = pairs.shape[0]
    used_out, first_touch = set(), {}
    for k in range(kvol):
        p = pairs[k, :, :num[k]]
        assert np.all(np.diff(p[0]) > 0) and (num[k] == 0 or (p[0].max() < len(idx) and p[1].max() < m))
        assert len(set(p[1])) == num[k]                               # one input per (output, offset)
        used_out |= set(p[1])
        for i, o in p.T:
            first_touch[o] = min(first_touch.get(o, (1 << 60, 0)), (int(i), k))
    assert used_out == set(range(m))                                  # no output without a contributing input
    # first-touch numbering (spconv CPU order): outputs are numbered by their earliest (input row, offset) token
    order = sorted(range(m), key=lambda o: first_touch[o])
    assert order == list(range(m))


@settings(**SETTINGS)
@given(sparse_sites(), st.sampled_from([(3, 2, 1), (3, 2, 0), ((3, 1, 1), (2, 1, 1), 0), (2, 2, 0), (3, 1, 1), (3, 2, (0, 1, 1))]))
def test_sorted_numbering_is_a_relabelling_of_first_touch(case, geom):
    """spconv's GPU numbering (ascending linear cell index) vs its CPU numbering (first touch): the same set of output sites,
    the same (offset, input row, output cell) triples, rows in strictly ascending cell order, pairs in ascending input row."""
    idx, batch, shape = case
    ks, stv, pad = geom
    if min(orc.conv_output_size(shape, ks, stv, pad, 1)) <= 0:
        return
    o1, p1, n1, oshape = orc.rulebook_conv(idx, batch, shape, ks, stv, pad)
    o2, p2, n2, _ = orc.rulebook_conv_sorted(idx, batch, shape, ks, stv, pad)
    np.testing.assert_array_equal(n1, n2)
    lin = ((o2[:, 0].astype(np.int64) * oshape[0] + o2[:, 1]) * oshape[1] + o2[:, 2]) * oshape[2] + o2[:, 3]
    assert np.all(np.diff(lin) > 0)
    assert {tuple(r) for r in o1} == {tuple(r) for r in o2}
    for k in range(p1.shape[0]):
        c = n1[k]
        np.testing.assert_array_equal(p1[k, 0, :c], p2[k, 0, :c])                      # same inputs, same (ascending) order
        np.testing.assert_array_equal(o1[p1[k, 1, :c]], o2[p2[k, 1, :c]])              # reaching the same output cells


@settings(max_examples=15, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 60))
def test_voxel_count_is_monotone_in_max_voxels(seed, cap):
    rng = np.random.default_rng(seed)
    pts = np.concatenate([rng.uniform([0, -4, -3], [8, 4, 1], (300, 3)), rng.uniform(0, 1, (300, 1))], 1).astype(np.float32)
    vs, rg = [0.5, 0.5, 0.5], [0, -4, -3, 8, 4, 1]
    for mode in ("break", "continue"):
        a = orc.points_to_voxel(pts, vs, rg, 3, cap, mode)
        b = orc.points_to_voxel(pts, vs, rg, 3, cap + 7, mode)
        assert a["voxel_num"] <= b["voxel_num"] <= cap + 7 and a["voxel_num"] <= cap
        np.testing.assert_array_equal(a["coordinates"], b["coordinates"][:a["voxel_num"]])      # a prefix: first-occurrence order
        assert np.all(a["num_points_per_voxel"] <= b["num_points_per_voxel"][:a["voxel_num"]])
        assert np.all(a["num_points_per_voxel"] >= 1) and np.all(a["num_points_per_voxel"] <= 3)
    brk = orc.points_to_voxel(pts, vs, rg, 3, cap, "break")
    cont = orc.points_to_voxel(pts, vs, rg, 3, cap, "continue")
    np.testing.assert_array_equal(brk["coordinates"], cont["coordinates"])
    assert np.all(brk["num_points_per_voxel"] <= cont["num_points_per_voxel"])     # `continue` keeps filling existing voxels


# ------------------------------------------------------------------------------------------------ the HIP path, BASELINE sizes
@pytest.mark.gpu
def test_gpu_rulebook_laws_at_bench_size():
    import torch
    from second_amd import ops, synthetic as syn
    clouds = [syn.syn_kitti_cloud(s) for s in range(8)]
    pts, offs = syn.batch_clouds(clouds)
    vox = ops.voxelize(torch.from_numpy(pts).cuda(), torch.from_numpy(offs).cuda(), syn.CAR_FHD_RANGE, syn.CAR_FHD_VOXEL, 5, 40000)
    idx, shape = vox["coordinates"].contiguous(), [41, 1600, 1408]
    assert idx.shape[0] == 128000
    for ks, stv, pad in [(3, 2, 1), (3, 2, 1), (3, 2, (0, 1, 1)), ((3, 1, 1), (2, 1, 1), 0)]:
        # SubM laws on this level
        rb = ops.rulebook_subm(idx, 8, shape, 3, want_pairs=True)
        n = idx.shape[0]
        nbr = rb["nbr_out"]
        assert torch.equal(nbr[:, 13], torch.arange(n, device="cuda", dtype=torch.int32))          # centre offset = identity
        assert rb["pair_num"][13].item() == n
        for k in (0, 5, 12):                                                                       # mirror symmetry of the table
            o = torch.nonzero(nbr[:, k] >= 0).squeeze(1)
            i = nbr[o, k].long()
            assert torch.equal(nbr[i, 26 - k].long(), o)
            assert rb["pair_num"][k].item() == o.numel() == rb["pair_num"][26 - k].item()
        # strided laws
        r = ops.rulebook_conv(idx, 8, shape, ks, stv, pad, want_pairs=True)
        out, m = r["out_indices"], r["num_out"]
        oshape = r["out_shape"]
        lin = ((out[:, 0].long() * oshape[0] + out[:, 1]) * oshape[1] + out[:, 2]) * oshape[2] + out[:, 3]
        assert torch.unique(lin).numel() == m                                                      # unique output sites
        no, ni = r["nbr_out"], r["nbr_in"]
        assert (no >= 0).any(1).all()                                                              # every output has an input
        assert int((no >= 0).sum()) == int((ni >= 0).sum()) == int(r["pair_num"].sum())            # both tables hold the same pairs
        k = 4 if no.shape[1] == 27 else 1
        o = torch.nonzero(no[:, k] >= 0).squeeze(1)
        assert torch.equal(ni[no[o, k].long(), k].long(), o)
        # first-touch numbering: the earliest (input row, offset) token of the outputs is increasing in the output row
        tok = torch.where(no >= 0, no.long() * no.shape[1] + torch.arange(no.shape[1], device="cuda"), torch.full_like(no, 1 << 40, dtype=torch.long))
        first = tok.min(1).values
        assert torch.all(first[1:] > first[:-1])
        idx, shape = out.contiguous(), oshape


@pytest.mark.gpu
def test_gpu_sorted_rulebook_chain_laws_at_bench_size():
    """The device fast path's rulebook chain at batch 8 x 16 000 voxels (bitmap builds, layers 2-4 derived from the previous
    layer's bitmap, SubM lookups by bitmap rank): ascending-cell output order, the same output sites and the same pair counts as
    the first-touch (hash) build, tables consistent with each other, SubM tables equal to the stand-alone build."""
    import torch
    from second_amd import ops, synthetic as syn
    pts, offs = syn.batch_clouds([syn.syn_kitti_cloud(s) for s in range(8)])
    vox = ops.voxelize(torch.from_numpy(pts).cuda(), torch.from_numpy(offs).cuda(), syn.CAR_FHD_RANGE, syn.CAR_FHD_VOXEL, 5, 40000)
    idx, shape, sites = vox["coordinates"].contiguous(), [41, 1600, 1408], None
    idx_ft = idx
    for li, (ks, stv, pad) in enumerate([(3, 2, 1), (3, 2, 1), (3, 2, (0, 1, 1)), ((3, 1, 1), (2, 1, 1), 0)]):
        r = ops.rulebook_conv(idx, 8, shape, ks, stv, pad, want_pairs=True, numbering="sorted", in_sites=sites)
        ft = ops.rulebook_conv(idx_ft, 8, shape, ks, stv, pad, want_pairs=True)
        out, m, oshape = r["out_indices"], r["num_out"], r["out_shape"]
        assert m == ft["num_out"] and torch.equal(r["pair_num"], ft["pair_num"]), li
        lin = ((out[:, 0].long() * oshape[0] + out[:, 1]) * oshape[1] + out[:, 2]) * oshape[2] + out[:, 3]
        assert torch.all(lin[1:] > lin[:-1])                                                       # ascending cell order
        fo = ft["out_indices"]
        lin_ft = ((fo[:, 0].long() * oshape[0] + fo[:, 1]) * oshape[1] + fo[:, 2]) * oshape[2] + fo[:, 3]
        assert torch.equal(torch.sort(lin_ft).values, lin)                                         # the same output sites
        no, ni = r["nbr_out"], r["nbr_in"]
        assert (no >= 0).any(1).all() and int((no >= 0).sum()) == int((ni >= 0).sum()) == int(r["pair_num"].sum())
        k = 4 if no.shape[1] == 27 else 1
        o = torch.nonzero(no[:, k] >= 0).squeeze(1)
        assert torch.equal(ni[no[o, k].long(), k].long(), o)
        sub = ops.rulebook_subm(out, 8, oshape, 3, site_table=r["site_table"])                    # bitmap-rank site lookup
        plain = ops.rulebook_subm(out.clone(), 8, oshape, 3)
        assert torch.equal(sub["nbr_out"], plain["nbr_out"])
        idx, idx_ft, shape, sites = out.contiguous(), fo.contiguous(), oshape, r["site_table"]


@pytest.mark.gpu
def test_gpu_voxel_count_monotone_and_prefix_at_bench_size():
    import torch
    from second_amd import ops, synthetic as syn
    cloud = syn.syn_kitti_cloud(3)
    pts, offs = syn.batch_clouds([cloud])
    pts, offs = torch.from_numpy(pts).cuda(), torch.from_numpy(offs).cuda()
    prev = None
    for cap in (1000, 5000, 15999, 16000, 40000):
        for mode in ("break", "continue"):
            v = ops.voxelize(pts, offs, syn.CAR_FHD_RANGE, syn.CAR_FHD_VOXEL, 5, cap, mode)
            assert v["voxel_num"] == min(cap, 16000)
            if mode == "break" and prev is not None:
                assert torch.equal(prev["coordinates"], v["coordinates"][:prev["voxel_num"]])
                assert torch.all(prev["num_points_per_voxel"] <= v["num_points_per_voxel"][:prev["voxel_num"]])
            if mode == "break":
                prev = v


@pytest.mark.gpu
def test_gpu_rulebook_and_conv_laws_at_nuscenes_size():
    """BASELINE configs 4 / 5 sizes (4 synthetic 10-sweep nuScenes clouds, ~293 k points each; all.fhd grid 40 x 1984 x 1984, one
    point per voxel, cap 90 000 per frame): where an element-by-element oracle comparison would take minutes, the size-independent
    laws -- SubM table symmetric with identity centre, the sorted strided chain ascending / unique / consistent in both tables, its
    site-map equal to the generic one -- and for indice_conv at that size: linearity in the features and in the weights, a
    permutation of the input rows (with the table relabelled) leaves the output unchanged, fp32 and bf16."""
    import torch
    from second_amd import ops, synthetic as syn
    from second_amd.models import ALL_FHD_NUSC as C
    clouds = [syn.syn_nusc_cloud(s, 293000, tuple(C["point_cloud_range"]), scene="urban") for s in range(4)]
    pts, offs = syn.batch_clouds(clouds)
    vox = ops.voxelize(torch.from_numpy(pts).cuda(), torch.from_numpy(offs).cuda(), C["point_cloud_range"], C["voxel_size"], 1, C["max_voxels"])
    idx = vox["coordinates"].contiguous()
    n = idx.shape[0]
    assert n == 4 * 90000                                                              # every frame hits its cap
    shape = [41, 1984, 1984]
    rb = ops.rulebook_subm(idx, 4, shape, 3, want_pairs=True)
    nbr = rb["nbr_out"]
    assert torch.equal(nbr[:, 13], torch.arange(n, device="cuda", dtype=torch.int32)) and rb["pair_num"][13].item() == n
    for k in (1, 9, 12):
        o = torch.nonzero(nbr[:, k] >= 0).squeeze(1)
        assert torch.equal(nbr[nbr[o, k].long(), 26 - k].long(), o) and rb["pair_num"][k].item() == o.numel()
    # sorted strided chain (the device fast path): two levels down
    cur, cshape, sites = idx, shape, None
    for ks, stv, pad in [(3, 2, 1), (3, 2, 1)]:
        r = ops.rulebook_conv(cur, 4, cshape, ks, stv, pad, want_pairs=True, numbering="sorted", in_sites=sites)
        out, m, oshape = r["out_indices"], r["num_out"], r["out_shape"]
        lin = ((out[:, 0].long() * oshape[0] + out[:, 1]) * oshape[1] + out[:, 2]) * oshape[2] + out[:, 3]
        assert torch.all(lin[1:] > lin[:-1])
        no, ni = r["nbr_out"], r["nbr_in"]
        assert (no >= 0).any(1).all() and int((no >= 0).sum()) == int((ni >= 0).sum()) == int(r["pair_num"].sum())
        o = torch.nonzero(no[:, 4] >= 0).squeeze(1)
        assert torch.equal(ni[no[o, 4].long(), 4].long(), o)
        cur, cshape, sites = out.contiguous(), oshape, r["site_table"]
    assert torch.equal(ops.sparse_site_map_sorted(sites[1], cur.shape[0], 4, cshape), ops.sparse_site_map(cur, 4, cshape))
    # indice_conv laws on the first level's SubM rulebook (360 000 rows, 1.5 M pairs)
    g = torch.Generator(device="cuda").manual_seed(7)
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        cin, cout = 16, 32
        x = torch.randn(n, cin, device="cuda", generator=g).to(dtype)
        y = torch.randn(n, cin, device="cuda", generator=g).to(dtype)
        w = (torch.randn(3, 3, 3, cin, cout, device="cuda", generator=g) / 20).to(dtype)
        v = (torch.randn(3, 3, 3, cin, cout, device="cuda", generator=g) / 20).to(dtype)

        def conv(f, wt, table=nbr):
            return ops.indice_conv(f, wt, table, n, packed=ops.pack_weight(wt)).float()
        fx, fy = conv(x, w), conv(y, w)
        scale = fx.abs().max().item()
        assert (conv((x.float() + y.float()).to(dtype), w) - (fx + fy)).abs().max().item() <= tol * 4 * scale       # linear in the features
        assert (conv(x, (w.float() + v.float()).to(dtype)) - (fx + conv(x, v))).abs().max().item() <= tol * 4 * scale  # ... and in the weights
        perm = torch.randperm(n, device="cuda", generator=g)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(n, device="cuda")
        relabelled = torch.where(nbr >= 0, inv[nbr.clamp(min=0).long()].int(), nbr)    # row r now lives at position inv[r]
        xp = torch.empty_like(x)
        xp[inv] = x
        assert torch.equal(conv(xp, w, relabelled.contiguous()), fx)                    # same sums in the same order: same bits


@pytest.mark.gpu
def test_gpu_nms_and_iou_laws_at_full_size():
    """1000 candidates per frame (nms_pre_max_size) x 8 frames: rotated IoU is symmetric and 1 on the diagonal of distinct
    axis-aligned boxes; NMS is idempotent (the kept boxes survive a second pass untouched), its keep list is ascending (score
    order), no kept pair overlaps beyond the threshold, and every dropped box overlaps an earlier kept one."""
    import torch
    from second_amd import ops
    rng = np.random.default_rng(17)
    b, n, thr = 8, 1000, 0.3
    dets = np.zeros((b, n, 6), np.float32)
    for f in range(b):
        c = rng.uniform(-35, 35, (40, 2))
        o = rng.integers(0, 40, n)
        dets[f, :, 0:2] = c[o] + rng.normal(0, 0.8, (n, 2))
        dets[f, :, 2] = rng.uniform(1.4, 2.0, n)
        dets[f, :, 3] = rng.uniform(3.4, 4.6, n)
        dets[f, :, 4] = rng.uniform(-3.2, 3.2, n)
        dets[f, :, 5] = np.sort(rng.uniform(0.3, 1, n))[::-1]
    d = torch.from_numpy(dets).cuda()
    counts = torch.full((b,), n, dtype=torch.int32, device="cuda")
    keep, num = ops.nms_sorted(d, counts, thr, "rotate", "numba", 1.0, 0)
    for f in range(b):
        k = keep[f, :int(num[f])].long()
        assert torch.all(k[1:] > k[:-1])
        kept = d[f, k][:, :5].contiguous()
        iou = ops.rotate_iou(kept, kept)
        assert torch.allclose(iou, iou.t(), atol=2e-5)
        off = iou - torch.diag(torch.diag(iou))
        assert off.max().item() <= thr + 2e-5                                           # no kept pair overlaps beyond the threshold
        dropped = torch.ones(n, dtype=torch.bool, device="cuda")
        dropped[k] = False
        di = torch.nonzero(dropped).squeeze(1)
        cross = ops.rotate_iou(d[f, di][:, :5].contiguous(), kept)                      # [dropped, kept]
        earlier = k.view(1, -1) < di.view(-1, 1)
        assert torch.all(((cross > thr - 2e-5) & earlier).any(1))                       # every dropped box has an earlier kept suppressor
        again, num2 = ops.nms_sorted(d[f:f + 1, k].contiguous(), torch.tensor([k.numel()], dtype=torch.int32, device="cuda"), thr,
                                     "rotate", "numba", 1.0, 0)
        assert int(num2[0]) == k.numel() and torch.equal(again[0, :k.numel()].long(), torch.arange(k.numel(), device="cuda"))
