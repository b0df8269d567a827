"""-m gpu: sec_conv2d_nhwc_x3 -- the RPN's 3x3 convolution for fp32 networks (the reference's default precision,
second/pytorch/train.py:232-235; rpn.py:468-497) on the bf16 matrix pipe with split operands: v = bf16(v) + bf16(v - bf16(v)),
three passes (x_hi w_hi, x_hi w_lo, x_lo w_hi) accumulated in fp32.  Checked against torch's convolution in float64 on the same
fp32 inputs; BASELINE.json's tolerance for fp32 features is 1e-4 relative (here: of the layer's range), the split form is
expected an order of magnitude inside it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_split_and_merge_round_trip_to_16_significant_bits():
    from second_amd import ops
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(3, 128, 20, 24, generator=g) * torch.logspace(-6, 6, 128).view(1, -1, 1, 1)).cuda().contiguous(memory_format=torch.channels_last)
    x[0, :, 3, 4] = 0.0
    hi, lo = ops.split_bf16x2(x)
    assert hi.dtype == lo.dtype == torch.bfloat16 and hi.stride() == x.stride()
    assert torch.equal(hi, x.to(torch.bfloat16)) and torch.equal(lo, (x - hi.float()).to(torch.bfloat16))
    y = ops.merge_bf16x2(hi, lo)
    assert y.dtype == torch.float32 and y.stride() == x.stride()
    assert torch.equal(y, hi.float() + lo.float())
    rel = ((y - x).abs() / x.abs().clamp_min(1e-30)).max().item()
    assert rel <= 2.0 ** -16, rel
    assert torch.all(y[0, :, 3, 4] == 0)


@pytest.mark.parametrize("batch,h,w,cout,relu", [(2, 200, 176, 128, True), (1, 37, 45, 128, False), (3, 8, 16, 256, True)])
def test_conv2d_x3_matches_fp64_convolution(batch, h, w, cout, relu):
    from second_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(batch, 128, h, w, generator=g).mul_(3.0)
    x = torch.where(torch.rand(x.shape, generator=g) < 0.3, torch.zeros(()), x)       # ReLU-like inputs
    wt = torch.randn(cout, 128, 3, 3, generator=g) * 0.05
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), wt.double(), bias.double(), 1, 1)
    if relu:
        ref = ref.clamp_min(0)
    xc = x.cuda().contiguous(memory_format=torch.channels_last)
    hi, lo = ops.split_bf16x2(xc)
    pk = ops.conv2d_pack_weight_x3(wt.cuda())
    yh, yl = ops.conv2d_nhwc_x3(hi, lo, pk, bias.cuda(), cout, relu=relu)
    assert yh.is_contiguous(memory_format=torch.channels_last) and yh.shape == (batch, cout, h, w)
    y = ops.merge_bf16x2(yh, yl).cpu().double()
    scale = float(ref.abs().max())
    err = float((y - ref).abs().max()) / scale
    assert err <= 2e-5, err                                   # BASELINE's bound is 1e-4
    # the residual plane is at most half a unit in the last place of the leading plane
    assert float((yl.float().abs() - yh.float().abs() * 2.0 ** -8).max()) <= 0.0
    # a one-pass bf16 convolution of the same data is two orders of magnitude further away (the test can tell the forms apart)
    y16 = F.conv2d(x.to(torch.bfloat16).double(), wt.to(torch.bfloat16).double(), bias.double(), 1, 1)
    if relu:
        y16 = y16.clamp_min(0)
    assert float((y16 - ref).abs().max()) / scale > 20 * err


def test_conv2d_x3_all_zero_tiles_write_the_bias_and_match_the_full_form():
    """flag bit 1 (the first RPN layer reads the scattered sparse-middle image): all-zero input tiles skip the three passes."""
    from second_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.zeros(2, 128, 64, 96)
    x[:, :, 10:14, 20:30] = torch.randn(2, 128, 4, 10, generator=g)
    x[1, :, 60:, 90:] = torch.randn(128, 4, 6, generator=g)
    wt = torch.randn(128, 128, 3, 3, generator=g) * 0.05
    bias = torch.randn(128, generator=g)
    xc = x.cuda().contiguous(memory_format=torch.channels_last)
    hi, lo = ops.split_bf16x2(xc)
    pk = ops.conv2d_pack_weight_x3(wt.cuda())
    a = ops.conv2d_nhwc_x3(hi, lo, pk, bias.cuda(), 128, relu=True, sparse_input=True)
    b = ops.conv2d_nhwc_x3(hi, lo, pk, bias.cuda(), 128, relu=True, sparse_input=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    ref = F.conv2d(x.double(), wt.double(), bias.double(), 1, 1).clamp_min(0)
    y = ops.merge_bf16x2(*a).cpu().double()
    assert float((y - ref).abs().max()) / float(ref.abs().max()) <= 2e-5


def test_fp32_rpn_inference_on_split_convs_matches_the_torch_block():
    """RPNInference(dtype=float32): six x3 convs + torch fp32 1x1 tail vs the module graph (Conv2d / BatchNorm2d / ReLU, rpn.py:468-497)."""
    from second_amd.models import RPNV2, RPNInference
    torch.manual_seed(0)
    rpn = RPNV2().eval()
    g = torch.Generator().manual_seed(3)
    for m in rpn.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.empty_like(m.running_mean).uniform_(-0.1, 0.1, generator=g))
            m.running_var.copy_(torch.empty_like(m.running_var).uniform_(0.5, 1.5, generator=g))
            m.weight.data.uniform_(0.8, 1.6, generator=g)
    rpn = rpn.cuda()
    x = torch.randn(2, 128, 200, 176, generator=g).clamp_min(0).cuda()
    with torch.no_grad():
        want = rpn(x)
        inf = RPNInference(rpn, torch.float32)
        assert inf.packed_x3 is not None and len(inf.packed_x3) == 6
        got = inf(x.contiguous(memory_format=torch.channels_last))
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        a, b = got[k].float().cpu().numpy(), want[k].float().cpu().numpy()
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-4 * float(np.abs(b).max()), err_msg=k)


def _sites(batch, h, w, n, seed):
    g = torch.Generator().manual_seed(seed)
    idx = torch.stack([torch.randint(0, batch, (n,), generator=g), torch.randint(0, 2, (n,), generator=g),
                       torch.randint(0, max(h // 3, 1), (n,), generator=g), torch.randint(0, w, (n,), generator=g)], 1).int()
    idx[:4, 2] = torch.tensor([0, 0, h - 1, h - 1])                      # the four corners too
    idx[:4, 3] = torch.tensor([0, w - 1, 0, w - 1])
    if batch > 1:
        idx = idx[idx[:, 0] != batch - 1]                                # the last frame stays empty
    return torch.unique(idx, dim=0)


@pytest.mark.parametrize("batch,h,w,n", [(3, 200, 176, 700), (2, 37, 50, 12), (2, 120, 97, 4000)])
def test_fp32_rpn_on_live_tiles_is_bit_identical_to_the_full_convolutions(batch, h, w, n):
    """RPNInference(float32) fed the sparse middle's rows (SparseBEV): conv j convolves the tiles within j + 1 steps of a site only
    (sec_conv2d_nhwc_x3_tiles), reads halo pixels of unwritten tiles from the empty frame's planes, the last conv fills in its
    background -- the head outputs must equal the full split-operand convolutions BIT FOR BIT on a network whose background is not
    zero, with the unwritten tiles poisoned with NaN."""
    import spconv
    from second_amd import ops
    from second_amd.models import RPNV2, RPNInference, SparseBEV
    torch.manual_seed(0)
    rpn = RPNV2().eval()
    g = torch.Generator().manual_seed(4)
    for m in rpn.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.empty_like(m.running_mean).uniform_(-0.3, 0.1, generator=g))
            m.running_var.copy_(torch.empty_like(m.running_var).uniform_(0.5, 1.5, generator=g))
            m.bias.data.uniform_(-0.1, 0.3, generator=g)                   # a non-zero background and border imprint
    inf = RPNInference(rpn.cuda(), torch.float32)
    assert inf.packed_x3 is not None and inf.background_convs == 6
    idx = _sites(batch, h, w, n, seed=h + n).cuda()
    feats = torch.randn(idx.shape[0], 64, generator=g).abs().cuda()
    sp = spconv.SparseConvTensor(feats, idx, [2, h, w], batch)
    merged = []          # the (hi, lo) planes behind the last 3x3 conv (what the fused 1x1 tail reads), as fp32: compared bit for bit

    def run(x):
        ops.set_op_hook(lambda name, fn, a, kw, res: merged.append(a[0].float() + a[1].float()) if name == "conv1x1_chain_x3" else None)
        try:
            with torch.no_grad():
                return {k: v.float().clone() for k, v in inf(x).items()}
        finally:
            ops.set_op_hook(None)
    inf.skip_background = False
    want = run(SparseBEV(sp))
    dense = run(sp.dense_channels_last_2d())
    inf.skip_background = True
    ops.POISON_LAZY_OUTPUTS = True
    try:
        got = run(SparseBEV(sp))
    finally:
        ops.POISON_LAZY_OUTPUTS = False
    live = inf.last_live_counts.cpu().numpy()
    tiles = ((h + 7) // 8) * ((w + 15) // 16)
    assert live.shape == (6, batch) and (batch == 1 or (live[:, -1] == 0).all())
    assert live[0].sum() < batch * tiles or n > 3000
    assert len(merged) == 3 and torch.equal(merged[0], merged[1])
    assert torch.isfinite(merged[2]).all(), "an unwritten (NaN-poisoned) tile reached the last conv's output"
    assert torch.equal(merged[2], merged[0])
    for k in want:                                   # the tail is one deterministic launch on identical planes
        assert torch.equal(dense[k], want[k]), k
        assert torch.equal(got[k], want[k]), k


def test_conv1x1_chain_x3_matches_fp64():
    """sec_conv1x1_chain_x3: y = W2 relu(W1 x + b1) + b2 on (hi, lo) planes vs float64 on the same fp32 inputs (pixel count not a
    multiple of the 128-pixel workgroup tile)."""
    from second_amd import ops
    g = torch.Generator().manual_seed(7)
    b, h, w = 2, 37, 45
    x = torch.randn(b, 128, h, w, generator=g).clamp_min(0) * 2.0
    w1 = torch.randn(128, 128, 1, 1, generator=g) * 0.1
    b1 = torch.randn(128, generator=g) * 0.3
    w2 = torch.randn(64, 128, 1, 1, generator=g) * 0.1
    w2[20:] = 0                                                   # padded head channels
    b2 = torch.randn(64, generator=g)
    ref = F.conv2d(F.conv2d(x.double(), w1.double(), b1.double()).clamp_min(0), w2.double(), b2.double())
    hi, lo = ops.split_bf16x2(x.cuda().contiguous(memory_format=torch.channels_last))
    y = ops.conv1x1_chain_x3(hi, lo, ops.conv2d_pack_weight_x3(w1.cuda()), b1.cuda(), ops.conv2d_pack_weight_x3(w2.cuda()), b2.cuda(), 64)
    assert y.dtype == torch.float32 and y.shape == (b, 64, h, w) and y.is_contiguous(memory_format=torch.channels_last)
    err = float((y.cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 2e-5, err
