"""-m gpu parity tests of the row-split sparse-conv kernels (k_conv_rows*, csrc/indice_conv.hip) against the CPU oracle.

These are the kernels the bench's `roofline` object is quoted on: the 64->64 SubMConv3d layers of car.fhd's subm2 group
(second/pytorch/models/middle.py:166-174).  Every case here
is built at the size of the bench launch (the automatic dispatch takes the row-split kernel from 40 000 rows on) -- batch 8 of the SURVEY 8(d) clouds through the ORACLE's voxeliser and rulebooks
(56 298 rows / 594 482 pairs) -- and `sec_indice_conv_fwd_plan` proves which kernel each comparison exercised.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402  (test infrastructure only)

PLAN_ROWS_BUF = 11
# (variant number, expected plan id): None = the automatic choice, 22 = the same kernel forced.  A library built with
# -DSEC_CONV_EXPERIMENTS also carries the superseded row-split forms (9-15: LDS-DMA / register-direct gathers) and the A/B forms of
# the buffer-load kernel (16-21, 23, 27, 28); they are run through the same comparisons when present.
ROW_VARIANTS = [(None, 11), (22, 11)]
# Round 3 added 41 = two row tiles per wave (k_conv_rows_m2, plan 13) and 46 = input planes staged in LDS windows (k_conv_rows_lds,
# plan 14: on the first-touch-ordered rows of the `layer` fixture its windows mostly MISS -- the exact gather fallback).
# Round 6 added 50 = the offsets of a row tile split over three wave groups (k_conv_rows_ks, plan 15; measured slower, not shipped).
EXPERIMENT_VARIANTS = [(9, 6), (10, 7), (11, 8), (12, 9), (13, 10), (14, 10), (15, 10), (16, 11), (17, 11), (18, 11), (19, 11),
                       (20, 11), (21, 11), (23, 11), (27, 11), (28, 11), (41, 13), (46, 14), (50, 15)]


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


@pytest.fixture(scope="module")
def ops():
    from second_amd import ops
    ops.indice_conv_set_variant(9)                       # plan 6 only exists in experiment builds
    if ops.indice_conv_plan(64, 64, 27, 50000, torch.bfloat16) == 6:
        ROW_VARIANTS.extend(EXPERIMENT_VARIANTS)
    ops.indice_conv_set_variant(-1)
    yield ops
    ops.indice_conv_set_variant(-1)


@pytest.fixture(scope="module")
def layer():
    """The exact subm2 launch of bench.py: rulebook from the oracle, output-major gather table derived from its pairs."""
    from second_amd import synthetic as syn
    idx = []
    for b in range(8):
        r = orc.points_to_voxel(syn.syn_kitti_cloud(b), syn.CAR_FHD_VOXEL, syn.CAR_FHD_RANGE, 5, 40000)
        idx.append(np.concatenate([np.full((r["voxel_num"], 1), b, np.int32), r["coordinates"]], 1))
    idx, shape = np.concatenate(idx), [41, 1600, 1408]
    for _ in range(2):                                   # the two stride-2 convs in front of the subm2 group
        idx, _, _, shape = orc.rulebook_conv(idx, 8, shape, 3, 2, 1)
        shape = [int(s) for s in shape]
    _, pairs, pair_num = orc.rulebook_subm(idx, 8, shape, 3)
    n = len(idx)
    nbr = -np.ones((n, 27), np.int32)
    for k in range(27):
        p = pairs[k, :, :pair_num[k]]
        nbr[p[1], k] = p[0]
    assert n >= 32768 and n % 128 not in (0, 1, 127)
    return {"idx": idx, "shape": shape, "pairs": pairs, "pair_num": pair_num, "nbr": nbr, "n": n}


def _operands(rng, n, dtype):
    feat = rng.standard_normal((n, 64)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, 64, 64)) / 40).astype(np.float32)
    f_t, w_t = dev(feat, dtype), dev(w, dtype)
    return f_t, w_t, f_t.float().cpu().numpy(), w_t.float().cpu().numpy()


def _tol(dtype):
    return 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10    # one rounding of the 16-bit store (tests/test_gpu_parity.py)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv_rows_bench_launch_vs_oracle(ops, layer, dtype):
    """(i) the bench launch itself + (v) the fused scale / shift / ReLU epilogue, every kernel form, vs the fp64-accumulating
    oracle on the same rounded operands."""
    rng = np.random.default_rng(11)
    n = layer["n"]
    f_t, w_t, f_np, w_np = _operands(rng, n, dtype)
    ref = orc.indice_conv(f_np, w_np, layer["pairs"], layer["pair_num"], n, acc64=True)
    scale = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    shift = rng.uniform(-0.2, 0.2, 64).astype(np.float32)
    ref_plain = torch.from_numpy(ref).to(dtype).float().numpy()
    ref_fused = torch.from_numpy(np.maximum(ref * scale + shift, 0)).to(dtype).float().numpy()
    packed, nbr = ops.pack_weight(w_t), dev(layer["nbr"])
    tol = _tol(dtype)
    for variant, plan in ROW_VARIANTS:
        ops.indice_conv_set_variant(-1 if variant is None else variant)
        assert ops.indice_conv_plan(64, 64, 27, n, dtype) == plan, (variant, plan)
        out = ops.indice_conv(f_t, w_t, nbr, n, packed=packed)
        assert out.dtype == dtype
        np.testing.assert_allclose(out.float().cpu().numpy(), ref_plain, rtol=tol, atol=tol * np.abs(ref).max(), err_msg=f"variant {variant}")
        out = ops.indice_conv(f_t, w_t, nbr, n, packed=packed, scale=dev(scale), shift=dev(shift), relu=True)
        np.testing.assert_allclose(out.float().cpu().numpy(), ref_fused, rtol=tol, atol=tol * np.abs(ref_fused).max(), err_msg=f"variant {variant}")
    ops.indice_conv_set_variant(-1)
    # mid-size layers (8 k .. 40 k rows) take the four-wave form of the same kernel, small ones split-K: a 23 k-row and a 6 k-row prefix
    # of the layer must agree with the oracle too
    assert ops.indice_conv_plan(64, 64, 27, 22834, dtype) == PLAN_ROWS_BUF
    assert ops.indice_conv_plan(64, 64, 27, 6000, dtype) != PLAN_ROWS_BUF
    for m in (22834, 6000):
        sub = layer["nbr"][:m]
        out = ops.indice_conv(f_t, w_t, dev(sub), m, packed=packed, scale=dev(scale), shift=dev(shift), relu=True)
        np.testing.assert_allclose(out.float().cpu().numpy(), ref_fused[:m], rtol=tol, atol=tol * np.abs(ref_fused).max(), err_msg=f"{m} rows")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv_rows_integer_operands_are_bit_exact(ops, layer, dtype):
    """Small-integer features and weights: every partial sum is an exactly representable integer, so the 16-bit output must
    equal the oracle BIT FOR BIT (no tolerance) -- catches any dropped, duplicated or misplaced (row, offset) contribution."""
    rng = np.random.default_rng(5)
    n = layer["n"]
    feat = rng.integers(-1, 2, (n, 64)).astype(np.float32)
    dens = 0.04 if dtype == torch.bfloat16 else 0.5
    w = (rng.integers(-1, 2, (3, 3, 3, 64, 64)) * (rng.random((3, 3, 3, 64, 64)) < dens)).astype(np.float32)
    ref = orc.indice_conv(feat, w, layer["pairs"], layer["pair_num"], n, acc64=True)
    limit = 256 if dtype == torch.bfloat16 else 2048
    assert np.abs(ref).max() <= limit and np.abs(ref).max() >= 8, np.abs(ref).max()
    f_t, w_t = dev(feat, dtype), dev(w, dtype)
    packed, nbr = ops.pack_weight(w_t), dev(layer["nbr"])
    for variant, plan in ROW_VARIANTS:
        ops.indice_conv_set_variant(-1 if variant is None else variant)
        assert ops.indice_conv_plan(64, 64, 27, n, dtype) == plan
        out = ops.indice_conv(f_t, w_t, nbr, n, packed=packed)
        np.testing.assert_array_equal(out.float().cpu().numpy(), ref, err_msg=f"variant {variant}")
    ops.indice_conv_set_variant(-1)


def test_conv_rows_ragged_tail_device_count_and_empty_rows(ops, layer):
    """(ii) n_out % 128 in {1, 127}, (iii) a device-side row count below the table capacity with garbage rows behind it,
    (iv) rows without any neighbour -- all bit-exact on integer operands, every kernel form."""
    dtype = torch.bfloat16
    rng = np.random.default_rng(9)
    n = layer["n"]
    feat = rng.integers(-1, 2, (n, 64)).astype(np.float32)
    w = (rng.integers(-1, 2, (3, 3, 3, 64, 64)) * (rng.random((3, 3, 3, 64, 64)) < 0.04)).astype(np.float32)
    nbr = layer["nbr"].copy()
    lonely = rng.choice(n, 3000, replace=False)
    nbr[lonely] = -1                                         # (iv) these rows must come out as act(shift)
    nbr[100:228] = -1                                        # a whole workgroup's worth of them
    pairs = -np.ones((27, 2, n), np.int32)
    pair_num = np.zeros(27, np.int32)
    for k in range(27):
        o = np.nonzero(nbr[:, k] >= 0)[0]
        pairs[k, 0, :len(o)], pairs[k, 1, :len(o)], pair_num[k] = nbr[o, k], o, len(o)
    shift = rng.integers(-2, 3, 64).astype(np.float32)
    ref = np.maximum(orc.indice_conv(feat, w, pairs, pair_num, n, acc64=True) + shift, 0)
    assert np.abs(ref).max() <= 256
    f_t, w_t = dev(feat, dtype), dev(w, dtype)
    packed = ops.pack_weight(w_t)
    sizes = [n - (n % 128) + 1 - 128, n - (n % 128) + 127 - 128, n]
    assert [m % 128 for m in sizes[:2]] == [1, 127] and min(sizes) >= 32768
    garbage = np.full((512, 27), 0x3fffffff, np.int32)       # rows past the live count: never dereferenced
    for variant, plan in ROW_VARIANTS:
        ops.indice_conv_set_variant(-1 if variant is None else variant)
        for m in sizes:
            assert ops.indice_conv_plan(64, 64, 27, m, dtype) == plan
            out = ops.indice_conv(f_t, w_t, dev(nbr[:m]), m, packed=packed, shift=dev(shift), relu=True)
            np.testing.assert_array_equal(out.float().cpu().numpy(), ref[:m], err_msg=f"variant {variant} n_out {m}")
            # static-capacity form: capacity m + 512, live count on the device
            table = dev(np.concatenate([nbr[:m], garbage]))
            out = ops.indice_conv(f_t, w_t, table, m + 512, packed=packed, shift=dev(shift), relu=True,
                                  num_out_dev=dev(np.array([m], np.int32)))
            np.testing.assert_array_equal(out[:m].float().cpu().numpy(), ref[:m], err_msg=f"variant {variant} static n_out {m}")
    ops.indice_conv_set_variant(-1)
    # the mid-size form (four-wave workgroups, rows per wave chosen on the device): ragged counts, static capacity
    for m in (8193, 20001, 24127, 39000):
        assert ops.indice_conv_plan(64, 64, 27, m, dtype) == PLAN_ROWS_BUF
        out = ops.indice_conv(f_t, w_t, dev(nbr[:m]), m, packed=packed, shift=dev(shift), relu=True)
        np.testing.assert_array_equal(out.float().cpu().numpy(), ref[:m], err_msg=f"mid-size n_out {m}")
        table = dev(np.concatenate([nbr[:m], garbage]))
        out = ops.indice_conv(f_t, w_t, table, m + 512, packed=packed, shift=dev(shift), relu=True,
                              num_out_dev=dev(np.array([m], np.int32)))
        np.testing.assert_array_equal(out[:m].float().cpu().numpy(), ref[:m], err_msg=f"mid-size static n_out {m}")
    torch.cuda.synchronize()


# ------------------------------------------------------------------ the buffer-load kernel on every layer shape of SpMiddleFHD
BUF_SHAPES = [(16, 16, 3), (16, 32, 3), (32, 32, 3), (32, 64, 3), (64, 64, 3), (64, 64, (3, 1, 1))]


@pytest.mark.parametrize("cin,cout,ksize", BUF_SHAPES)
@pytest.mark.parametrize("subm", [True, False])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv_rows_buf_all_layer_shapes(ops, cin, cout, ksize, subm, dtype):
    """SEC_CONV_VARIANT 22 forces k_conv_rows_buf for every (Cin, Cout, kernel) of middle.py:146-189, SubM and strided tables,
    ragged row counts; integer operands -> bit-exact against the oracle, Gaussian operands -> within one 16-bit rounding."""
    from test_gpu_parity import _random_indices, _tables_from_pairs
    if subm and ksize != 3:
        pytest.skip("the (3,1,1) kernel only occurs as a strided conv (middle.py:188)")
    rng = np.random.default_rng(cin * 1000 + cout * 10 + int(subm))
    shape = (9, 34, 30)
    idx = _random_indices(rng, 3, shape, 2500)
    if subm:
        _, pairs, pair_num = orc.rulebook_subm(idx, 3, shape, ksize)
        n_out = len(idx)
    else:
        stride = 2 if ksize == 3 else (2, 1, 1)
        out_idx, pairs, pair_num, _ = orc.rulebook_conv(idx, 3, shape, ksize, stride, 1 if ksize == 3 else 0)
        n_out = len(out_idx)
    kvol = 27 if ksize == 3 else 3
    nbr_out, _ = _tables_from_pairs(pairs, pair_num, len(idx), n_out)
    assert n_out % 256 not in (0,) and n_out > 600
    ops.indice_conv_set_variant(22)
    try:
        assert ops.indice_conv_plan(cin, cout, kvol, n_out, dtype) == 11
        # integer operands: exact
        feat = rng.integers(-1, 2, (len(idx), cin)).astype(np.float32)
        dens = 0.08 if dtype == torch.bfloat16 else 0.5
        w = (rng.integers(-1, 2, (kvol, cin, cout)) * (rng.random((kvol, cin, cout)) < dens)).astype(np.float32)
        wk = w.reshape((3, 3, 3, cin, cout) if kvol == 27 else (3, 1, 1, cin, cout))
        ref = orc.indice_conv(feat, wk, pairs, pair_num, n_out, acc64=True)
        assert np.abs(ref).max() <= (256 if dtype == torch.bfloat16 else 2048)
        f_t, w_t = dev(feat, dtype), dev(wk, dtype)
        out = ops.indice_conv(f_t, w_t, dev(nbr_out), n_out, packed=ops.pack_weight(w_t))
        assert out.dtype == dtype and out.shape == (n_out, cout)
        np.testing.assert_array_equal(out.float().cpu().numpy(), ref)
        # Gaussian operands + fused epilogue
        feat = rng.standard_normal((len(idx), cin)).astype(np.float32)
        wk = (rng.standard_normal(wk.shape) / np.sqrt(kvol * cin)).astype(np.float32)
        f_t, w_t = dev(feat, dtype), dev(wk, dtype)
        ref = orc.indice_conv(f_t.float().cpu().numpy(), w_t.float().cpu().numpy(), pairs, pair_num, n_out, acc64=True)
        scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        shift = rng.uniform(-0.2, 0.2, cout).astype(np.float32)
        ref_f = torch.from_numpy(np.maximum(ref * scale + shift, 0)).to(dtype).float().numpy()
        out = ops.indice_conv(f_t, w_t, dev(nbr_out), n_out, packed=ops.pack_weight(w_t), scale=dev(scale), shift=dev(shift), relu=True)
        tol = _tol(dtype)
        np.testing.assert_allclose(out.float().cpu().numpy(), ref_f, rtol=tol, atol=tol * np.abs(ref_f).max())
    finally:
        ops.indice_conv_set_variant(-1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_first_layer_c4_mfma_kernel(ops, dtype):
    """middle.py:146 SubMConv3d(4, 16, 3): K = 27 x 4 folded into seven MFMA steps (k_conv_c4_mfma).  Bench-size row count (ragged),
    integer operands bit-exact, Gaussian operands + fused epilogue within one 16-bit rounding, and agreement with the VALU kernel."""
    from test_gpu_parity import _random_indices, _tables_from_pairs
    rng = np.random.default_rng(4)
    shape = (12, 60, 56)
    idx = _random_indices(rng, 4, shape, 9000)                      # 36 000 rows: 282 workgroups, ragged last tile
    _, pairs, pair_num = orc.rulebook_subm(idx, 4, shape, 3)
    n = len(idx)
    assert n % 128 != 0
    nbr, _ = _tables_from_pairs(pairs, pair_num, n, n)
    assert ops.indice_conv_plan(4, 16, 27, n, dtype) == 12
    feat = rng.integers(-2, 3, (n, 4)).astype(np.float32)
    w = rng.integers(-1, 2, (3, 3, 3, 4, 16)).astype(np.float32)
    ref = orc.indice_conv(feat, w, pairs, pair_num, n, acc64=True)
    assert np.abs(ref).max() <= 256
    f_t, w_t = dev(feat, dtype), dev(w, dtype)
    packed = ops.pack_weight(w_t)
    assert packed is not None
    out = ops.indice_conv(f_t, w_t, dev(nbr), n, packed=packed)
    np.testing.assert_array_equal(out.float().cpu().numpy(), ref)
    # static-capacity form
    table = dev(np.concatenate([nbr, np.full((300, 27), 0x3fffffff, np.int32)]))
    out = ops.indice_conv(f_t, w_t, table, n + 300, packed=packed, num_out_dev=dev(np.array([n], np.int32)))
    np.testing.assert_array_equal(out[:n].float().cpu().numpy(), ref)
    # Gaussian operands, fused epilogue; and the thread-per-row VALU kernel (no packed weights) on the same inputs
    feat = rng.standard_normal((n, 4)).astype(np.float32) * 10
    w = (rng.standard_normal((3, 3, 3, 4, 16)) / 10).astype(np.float32)
    f_t, w_t = dev(feat, dtype), dev(w, dtype)
    ref = orc.indice_conv(f_t.float().cpu().numpy(), w_t.float().cpu().numpy(), pairs, pair_num, n, acc64=True)
    scale, shift = rng.uniform(0.5, 1.5, 16).astype(np.float32), rng.uniform(-0.2, 0.2, 16).astype(np.float32)
    ref_f = torch.from_numpy(np.maximum(ref * scale + shift, 0)).to(dtype).float().numpy()
    tol = _tol(dtype)
    a = ops.indice_conv(f_t, w_t, dev(nbr), n, packed=ops.pack_weight(w_t), scale=dev(scale), shift=dev(shift), relu=True)
    b = ops.indice_conv(f_t, w_t, dev(nbr), n, packed=None, scale=dev(scale), shift=dev(shift), relu=True)
    np.testing.assert_allclose(a.float().cpu().numpy(), ref_f, rtol=tol, atol=tol * np.abs(ref_f).max())
    np.testing.assert_allclose(b.float().cpu().numpy(), ref_f, rtol=tol, atol=tol * np.abs(ref_f).max())


def test_conv_rows_are_deterministic_beside_the_rpn_conv(ops, layer):
    """k_conv_rows_buf / k_conv_rows_m2 keep the packed-fp32 target feature (their fused scale / shift epilogue compiles to
    v_pk_mul_f32 / v_pk_add_f32) although the rest of the library is built without it: round 2 saw packed fp32 VALU results go
    wrong in a VALU-heavy kernel (the rotated-NMS clipper) while another wave of the CU ran the RPN conv's MFMA loop
    (tests/test_gpu_round2.py::test_nms_is_deterministic_beside_the_rpn_conv, tools/pkfp32_repro.py).  The exemption is therefore
    part of the stress-tested set: the bench launch (fused epilogue, every row-split form) on one stream while two other streams
    run the RPN conv back to back must return the same bits every time."""
    dtype = torch.bfloat16
    rng = np.random.default_rng(21)
    n = layer["n"]
    f_t, w_t, _, _ = _operands(rng, n, dtype)
    scale, shift = dev(rng.uniform(0.5, 1.5, 64).astype(np.float32)), dev(rng.uniform(-0.2, 0.2, 64).astype(np.float32))
    packed, nbr = ops.pack_weight(w_t), dev(layer["nbr"])
    x = torch.relu(torch.randn(8, 128, 200, 176, device="cuda")).bfloat16().contiguous(memory_format=torch.channels_last)
    wc = (torch.randn(128, 128, 3, 3, device="cuda") / 34).bfloat16()
    pk, bias = ops.conv2d_pack_weight(wc), torch.randn(128, device="cuda")
    load = [torch.cuda.Stream(), torch.cuda.Stream()]
    s_conv = torch.cuda.Stream()
    m_small = 22834                                          # the four-wave form of the mid-size layers
    for variant, plan in ROW_VARIANTS[:2] + [v for v in ROW_VARIANTS if v[0] == 41]:
        ops.indice_conv_set_variant(-1 if variant is None else variant)
        first, first_small = None, None
        for it in range(60):
            for s in load:
                with torch.cuda.stream(s):
                    for _ in range(2):
                        ops.conv2d_nhwc(x, pk, bias, 128, 3, 1, 1, relu=True)
            with torch.cuda.stream(s_conv):
                out = ops.indice_conv(f_t, w_t, nbr, n, packed=packed, scale=scale, shift=shift, relu=True)
                out_small = ops.indice_conv(f_t, w_t, nbr[:m_small], m_small, packed=packed, scale=scale, shift=shift, relu=True)
            torch.cuda.synchronize()
            if first is None:
                first, first_small = out.clone(), out_small.clone()
            else:
                assert torch.equal(out, first), (variant, it)
                assert torch.equal(out_small, first_small), (variant, it)
    ops.indice_conv_set_variant(-1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv_rows_lds_windows_hit_on_sorted_rows(ops, layer, dtype):
    """k_conv_rows_lds takes its operands from LDS windows of consecutive input rows; they only fill when the rows are in ascending
    cell order (the device fast path's sorted numbering).  The bench layer relabelled into that order -- same sites, same pairs --
    must give the oracle's result bit for bit on integer operands (window hits, window overflows and global fallbacks all mix
    in one launch), within one rounding on Gaussian operands with the fused epilogue, and the same bits with the ragged tail cut."""
    idx, shape, n = layer["idx"], layer["shape"], layer["n"]
    lin = ((idx[:, 0].astype(np.int64) * shape[0] + idx[:, 1]) * shape[1] + idx[:, 2]) * shape[2] + idx[:, 3]
    perm = np.argsort(lin, kind="stable")                    # new row j = old row perm[j]
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    old = layer["nbr"][perm]
    nbr = np.where(old >= 0, inv[np.maximum(old, 0)], -1).astype(np.int32)
    assert np.all(np.diff(lin[perm]) > 0)
    pairs = -np.ones((27, 2, n), np.int32)
    pair_num = np.zeros(27, np.int32)
    for k in range(27):
        o = np.nonzero(nbr[:, k] >= 0)[0]
        pairs[k, 0, :len(o)], pairs[k, 1, :len(o)], pair_num[k] = nbr[o, k], o, len(o)
    rng = np.random.default_rng(17)
    feat = rng.integers(-1, 2, (n, 64)).astype(np.float32)
    dens = 0.04 if dtype == torch.bfloat16 else 0.5
    w = (rng.integers(-1, 2, (3, 3, 3, 64, 64)) * (rng.random((3, 3, 3, 64, 64)) < dens)).astype(np.float32)
    ref = orc.indice_conv(feat, w, pairs, pair_num, n, acc64=True)
    f_t, w_t = dev(feat, dtype), dev(w, dtype)
    packed = ops.pack_weight(w_t)
    ops.indice_conv_set_variant(46)
    if ops.indice_conv_plan(64, 64, 27, n, dtype) != 14:
        ops.indice_conv_set_variant(-1)
        pytest.skip("k_conv_rows_lds is an A/B form: built with -DSEC_CONV_EXPERIMENTS only")
    try:
        assert ops.indice_conv_plan(64, 64, 27, n, dtype) == 14
        out = ops.indice_conv(f_t, w_t, dev(nbr), n, packed=packed)
        np.testing.assert_array_equal(out.float().cpu().numpy(), ref)
        m = n - (n % 256) - 255                              # ragged: the last workgroup holds one row
        out = ops.indice_conv(f_t, w_t, dev(nbr[:m]), m, packed=packed)
        np.testing.assert_array_equal(out.float().cpu().numpy(), ref[:m])
        table = dev(np.concatenate([nbr[:m], np.full((512, 27), 0x3fffffff, np.int32)]))
        out = ops.indice_conv(f_t, w_t, table, m + 512, packed=packed, num_out_dev=dev(np.array([m], np.int32)))
        np.testing.assert_array_equal(out[:m].float().cpu().numpy(), ref[:m])
        # Gaussian operands + fused epilogue
        f_g, w_g, f_np, w_np = _operands(rng, n, dtype)
        refg = orc.indice_conv(f_np, w_np, pairs, pair_num, n, acc64=True)
        scale, shift = rng.uniform(0.5, 1.5, 64).astype(np.float32), rng.uniform(-0.2, 0.2, 64).astype(np.float32)
        ref_f = torch.from_numpy(np.maximum(refg * scale + shift, 0)).to(dtype).float().numpy()
        out = ops.indice_conv(f_g, w_g, dev(nbr), n, packed=ops.pack_weight(w_g), scale=dev(scale), shift=dev(shift), relu=True)
        tol = _tol(dtype)
        np.testing.assert_allclose(out.float().cpu().numpy(), ref_f, rtol=tol, atol=tol * np.abs(ref_f).max())
    finally:
        ops.indice_conv_set_variant(-1)


def test_indice_conv_fp32_16_to_16_packed_split_form(ops):
    """The 16 -> 16 layer (subm0 of SpMiddleFHD) has no on-the-fly split kernel; with a pre-split weight it runs the pipelined
    split-operand kernel on a half-used 32-column tile."""
    from test_gpu_parity import _random_indices, _tables_from_pairs
    rng = np.random.default_rng(5)
    shape = (9, 34, 30)
    idx = _random_indices(rng, 3, shape, 2500)
    _, pairs, pair_num = orc.rulebook_subm(idx, 3, shape, 3)
    nbr, _ = _tables_from_pairs(pairs, pair_num, len(idx), len(idx))
    feat = rng.standard_normal((len(idx), 16)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, 16, 16)) / np.sqrt(27 * 16)).astype(np.float32)
    ref = orc.indice_conv(feat, w, pairs, pair_num, len(idx), acc64=True)
    pk = ops.pack_weight(dev(w))
    assert pk is not None and tuple(pk.shape[:2]) == (27, 2)
    scale, shift = rng.uniform(0.5, 1.5, 16).astype(np.float32), rng.uniform(-0.2, 0.2, 16).astype(np.float32)
    out = ops.indice_conv(dev(feat), dev(w), dev(nbr), len(idx), packed=pk, scale=dev(scale), shift=dev(shift), relu=True)
    assert "k_conv_rows_x3p_f32<16, 16" in ops.last_kernel_name(), ops.last_kernel_name()
    want = np.maximum(ref * scale + shift, 0)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-4, atol=1e-5 * np.abs(ref).max())


@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 32), (32, 64), (64, 32), (64, 64)])
@pytest.mark.parametrize("subm", [True, False])
def test_indice_conv_fp32_on_the_matrix_cores(ops, cin, cout, subm):
    """fp32 features (the reference's default precision) on the matrix cores: the split-operand form on the bf16 pipe
    (k_conv_rows_x3_f32, default since round 5: v = bf16(v) + bf16(v - bf16(v)), three MFMAs per product term, fp32 accumulation) and
    the fp32-MFMA form (k_conv_mfma_f32, variant 31), each against the fp64-accumulating oracle within 1e-4 of the range
    (BASELINE.json's tolerance) and against the VALU form (variant 30); ragged row counts, device-side row count below the capacity
    with garbage table rows behind it, fused scale / shift / ReLU."""
    from test_gpu_parity import _random_indices, _tables_from_pairs
    rng = np.random.default_rng(cin * 7 + cout + int(subm))
    shape = (9, 34, 30)
    idx = _random_indices(rng, 3, shape, 2500)
    if subm:
        _, pairs, pair_num = orc.rulebook_subm(idx, 3, shape, 3)
        n_out = len(idx)
    else:
        out_idx, pairs, pair_num, _ = orc.rulebook_conv(idx, 3, shape, 3, 2, 1)
        n_out = len(out_idx)
    nbr, _ = _tables_from_pairs(pairs, pair_num, len(idx), n_out)
    feat = rng.standard_normal((len(idx), cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    ref = orc.indice_conv(feat, w, pairs, pair_num, n_out, acc64=True)
    f_t, w_t = dev(feat), dev(w)
    out = ops.indice_conv(f_t, w_t, dev(nbr), n_out)
    assert "k_conv_rows_x3_f32" in ops.last_kernel_name(), ops.last_kernel_name()
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
    assert float(np.abs(out.cpu().numpy() - ref).max()) <= 2e-5 * float(np.abs(ref).max())      # an order of magnitude inside the bound
    # the inference form: weights pre-split and pre-packed once (ops.pack_weight of an fp32 weight), loop pipelined three offsets deep:
    # the same products in the same order -> bit-identical to the on-the-fly form
    pk = ops.pack_weight(w_t)
    assert pk is not None and pk.dtype == torch.bfloat16 and pk.shape[:2] == (27, 2)
    out_p = ops.indice_conv(f_t, w_t, dev(nbr), n_out, packed=pk)
    assert "k_conv_rows_x3p_f32" in ops.last_kernel_name(), ops.last_kernel_name()
    assert torch.equal(out_p, out)
    ops.indice_conv_set_variant(31)
    try:
        mf = ops.indice_conv(f_t, w_t, dev(nbr), n_out)
        assert "k_conv_mfma_f32" in ops.last_kernel_name(), ops.last_kernel_name()
    finally:
        ops.indice_conv_set_variant(-1)
    np.testing.assert_allclose(mf.cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
    ops.indice_conv_set_variant(30)
    try:
        valu = ops.indice_conv(f_t, w_t, dev(nbr), n_out)
    finally:
        ops.indice_conv_set_variant(-1)
    torch.testing.assert_close(out, valu, rtol=1e-4, atol=2e-5 * float(np.abs(ref).max()))
    torch.testing.assert_close(mf, valu, rtol=1e-5, atol=1e-5 * float(np.abs(ref).max()))
    # static capacity: garbage rows behind the live count, fused epilogue
    cap = n_out + 77
    nbr_pad = np.concatenate([nbr, rng.integers(0, len(idx), (77, 27)).astype(np.int32)])
    scale, shift = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.uniform(-0.2, 0.2, cout).astype(np.float32)
    n_dev = dev(np.array([n_out], np.int32))
    got = ops.indice_conv(f_t, w_t, dev(nbr_pad), cap, scale=dev(scale), shift=dev(shift), relu=True, num_out_dev=n_dev)
    want = np.maximum(ref * scale + shift, 0)
    np.testing.assert_allclose(got[:n_out].cpu().numpy(), want, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
    got_p = ops.indice_conv(f_t, w_t, dev(nbr_pad), cap, packed=pk, scale=dev(scale), shift=dev(shift), relu=True, num_out_dev=n_dev)
    assert torch.equal(got_p[:n_out], got[:n_out])
