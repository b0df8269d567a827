"""Independent anchor for the (otherwise parity-unpinned) spconv rulebook + indice_conv restatement:
a SubMConv3d / SparseConv3d must equal a dense torch conv3d restricted to the active sites
(upstream spconv test/test_conv.py uses the same contract).  CPU-only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as orc


def random_sparse(rng, batch, shape, n_per_batch, cin):
    idx = []
    for b in range(batch):
        lin = rng.choice(int(np.prod(shape)), size=n_per_batch, replace=False)
        z, y, x = np.unravel_index(lin, shape)
        idx.append(np.stack([np.full_like(z, b), z, y, x], 1))
    idx = np.concatenate(idx).astype(np.int32)
    rng.shuffle(idx)  # rows in arbitrary order, batches interleaved
    feat = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    return idx, feat


def dense_of(idx, feat, batch, shape):
    d = torch.zeros(batch, feat.shape[1], *shape)
    d[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = torch.from_numpy(feat)
    return d


@pytest.mark.parametrize("ksize", [3, (3, 1, 1), (1, 3, 3)])
def test_subm_equals_dense_conv_at_active_sites(ksize):
    rng = np.random.default_rng(0)
    batch, shape, cin, cout = 2, (7, 9, 8), 5, 6
    idx, feat = random_sparse(rng, batch, shape, 90, cin)
    ks = (ksize,) * 3 if np.isscalar(ksize) else ksize
    w = rng.standard_normal((*ks, cin, cout)).astype(np.float32)
    out_idx, pairs, pair_num = orc.rulebook_subm(idx, batch, shape, ks)
    out = orc.indice_conv(feat, w, pairs, pair_num, idx.shape[0])
    wt = torch.from_numpy(w).permute(4, 3, 0, 1, 2).contiguous()
    ref = F.conv3d(dense_of(idx, feat, batch, shape), wt, padding=tuple(k // 2 for k in ks))
    ref_rows = ref[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]].numpy()
    np.testing.assert_allclose(out, ref_rows, rtol=1e-4, atol=1e-5)
    # structural properties: centre offset pairs every site with itself; mirror symmetry
    K = int(np.prod(ks))
    c = K // 2
    assert pair_num[c] == idx.shape[0]
    np.testing.assert_array_equal(pairs[c, 0, :pair_num[c]], np.arange(idx.shape[0]))
    np.testing.assert_array_equal(pairs[c, 1, :pair_num[c]], np.arange(idx.shape[0]))
    for k in range(K):
        assert pair_num[k] == pair_num[K - 1 - k]
        a = set(map(tuple, pairs[k, :, :pair_num[k]].T))
        b = set((o, i) for i, o in pairs[K - 1 - k, :, :pair_num[K - 1 - k]].T)
        assert a == b
        # canonical order: ascending input row within an offset
        assert (np.diff(pairs[k, 0, :pair_num[k]]) > 0).all()


@pytest.mark.parametrize("ksize,stride,padding", [
    (3, 2, 1), (3, 2, (0, 1, 1)), ((3, 1, 1), (2, 1, 1), 0), (3, 1, 0), (2, 2, 0), (3, 1, 1)])
def test_sparse_conv_equals_dense_conv(ksize, stride, padding):
    rng = np.random.default_rng(1)
    batch, shape, cin, cout = 2, (9, 10, 11), 4, 7
    idx, feat = random_sparse(rng, batch, shape, 60, cin)
    t = lambda v: (v,) * 3 if np.isscalar(v) else tuple(v)
    ks, st, pd = t(ksize), t(stride), t(padding)
    w = rng.standard_normal((*ks, cin, cout)).astype(np.float32)
    out_idx, pairs, pair_num, out_shape = orc.rulebook_conv(idx, batch, shape, ks, st, pd)
    out = orc.indice_conv(feat, w, pairs, pair_num, out_idx.shape[0])
    wt = torch.from_numpy(w).permute(4, 3, 0, 1, 2).contiguous()
    ref = F.conv3d(dense_of(idx, feat, batch, shape), wt, stride=st, padding=pd)
    assert tuple(ref.shape[2:]) == tuple(out_shape)
    got = torch.zeros_like(ref)
    got[out_idx[:, 0], :, out_idx[:, 1], out_idx[:, 2], out_idx[:, 3]] = torch.from_numpy(out)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)
    # active outputs = exactly the sites whose receptive field holds an active input
    occ = F.conv3d(dense_of(idx, np.ones((idx.shape[0], 1), np.float32), batch, shape),
                   torch.ones(1, 1, *ks), stride=st, padding=pd)
    assert int((occ > 0).sum()) == out_idx.shape[0]
    assert len(set(map(tuple, out_idx))) == out_idx.shape[0]
    # first-touch numbering: the first pair (in ascending input order) that mentions an output id
    # appears in increasing id order
    first = {}
    for k in range(pairs.shape[0]):
        for i, o in pairs[k, :, :pair_num[k]].T:
            first[o] = min(first.get(o, (1 << 60, 0)), (int(i), k))
    keys = [first[o] for o in range(out_idx.shape[0])]
    assert keys == sorted(keys)


def test_rulebook_micro_cases():
    """Hand-computed: two x-adjacent voxels under 3x3x3 SubM; one voxel under stride 2."""
    idx = np.array([[0, 2, 2, 2], [0, 2, 2, 3]], np.int32)
    _, pairs, num = orc.rulebook_subm(idx, 1, (5, 5, 5), 3)
    assert num.tolist() == [0] * 12 + [1, 2, 1] + [0] * 12
    # offset k = 9*kz+3*ky+kx with out = in - (k-1): k=12 -> out.x = in.x+1 ; k=14 -> out.x = in.x-1
    assert pairs[12, :, 0].tolist() == [0, 1]
    assert pairs[14, :, 0].tolist() == [1, 0]
    assert pairs[13, :, :2].tolist() == [[0, 1], [0, 1]]
    # stride 2, k 3, pad 1: even coordinate -> offsets {1} per dim... in+pad odd -> 1 candidate
    out_idx, pairs, num, oshape = orc.rulebook_conv(np.array([[0, 2, 2, 2]], np.int32), 1, (5, 5, 5), 3, 2, 1)
    assert oshape.tolist() == [3, 3, 3]
    assert out_idx.tolist() == [[0, 1, 1, 1]] and num.sum() == 1 and num[13] == 1
    out_idx, pairs, num, _ = orc.rulebook_conv(np.array([[0, 1, 1, 1]], np.int32), 1, (5, 5, 5), 3, 2, 1)
    # in+pad = 2 (even) -> two candidates per dim: out 1 (offset 0) and out 0 (offset 2); descending out
    assert out_idx.shape[0] == 8 and out_idx[0].tolist() == [0, 1, 1, 1] and out_idx[-1].tolist() == [0, 0, 0, 0]
    assert num.sum() == 8 and num[0] == 1 and num[26] == 1


def test_backward_matches_autograd():
    rng = np.random.default_rng(2)
    batch, shape, cin, cout = 1, (6, 6, 6), 3, 4
    idx, feat = random_sparse(rng, batch, shape, 40, cin)
    w = rng.standard_normal((3, 3, 3, cin, cout)).astype(np.float32)
    out_idx, pairs, pair_num, _ = orc.rulebook_conv(idx, batch, shape, 3, 2, 1)
    dout = rng.standard_normal((out_idx.shape[0], cout)).astype(np.float32)
    dfeat, dw = orc.indice_conv_backward(feat, w, pairs, pair_num, dout)
    d = dense_of(idx, feat, batch, shape).requires_grad_(True)
    wt = torch.from_numpy(w).requires_grad_(True)
    y = F.conv3d(d, wt.permute(4, 3, 0, 1, 2), stride=2, padding=1)
    g = torch.zeros_like(y)
    g[out_idx[:, 0], :, out_idx[:, 1], out_idx[:, 2], out_idx[:, 3]] = torch.from_numpy(dout)
    y.backward(g)
    np.testing.assert_allclose(dw, wt.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dfeat, d.grad[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]].numpy(),
                               rtol=1e-4, atol=1e-5)


def test_dense_layout():
    rng = np.random.default_rng(3)
    idx, feat = random_sparse(rng, 2, (2, 5, 4), 10, 3)
    d = orc.sparse_to_dense(feat, idx, 2, (2, 5, 4))
    np.testing.assert_array_equal(d, dense_of(idx, feat, 2, (2, 5, 4)).numpy())
