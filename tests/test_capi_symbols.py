"""CPU-only: libsecond_hip.so builds/loads and exports every symbol include/second_hip.h declares
(no compute calls without a GPU), and the product fails loudly instead of falling back to a CPU path."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def header_functions():
    text = open(os.path.join(ROOT, "include", "second_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sec_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from second_amd import runtime as rt
    import importlib.util
    spec = importlib.util.spec_from_file_location("sec_build", os.path.join(ROOT, "second.pytorch_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build(verbose=False)
    names = header_functions()
    assert names == sorted(rt.SYMBOLS)
    lib = ctypes.CDLL(rt.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.sec_abi_version() == rt.ABI_VERSION == 9


def test_workspace_queries_are_host_only():
    from second_amd import runtime as rt
    l = rt.lib()
    assert l.sec_voxelize_workspace_bytes(17000, 1, 40000, 5) > 0
    assert l.sec_rulebook_workspace_bytes(16000, 27, 8) > 0
    assert l.sec_nms_workspace_bytes(8, 1000) == 8 * 1000 * 16 * 8
    assert l.sec_packed_weight_bytes(27, 64, 64, rt.SEC_BF16) == 27 * 64 * 64 * 2
    assert l.sec_packed_weight_bytes(27, 16, 16, rt.SEC_BF16) == 27 * 16 * 32 * 2  # Cout padded to 32 columns
    assert l.sec_packed_weight_bytes(27, 4, 16, rt.SEC_BF16) == 7 * 64 * 8 * 2     # Cin=4 first layer: K = 27 x 4 padded to 7 MFMA steps
    assert l.sec_packed_weight_bytes(27, 5, 7, rt.SEC_BF16) == 0                   # no MFMA layout: generic path
    out = (ctypes.c_int * 3)()
    l.sec_conv_output_shape(rt.i3([41, 1600, 1408]), rt.i3(3), rt.i3(2), rt.i3(1), rt.i3(1), out)
    assert list(out) == [21, 800, 704]


def test_lazy_background_entry_points_validate_before_any_launch():
    """The lazy forms refuse what cannot be right before touching the device (status codes of include/second_hip.h; no GPU needed):
    lists, masks and the producer's empty-frame map are mandatory for sec_conv2d_nhwc_tiles_lazy, the masks for
    sec_rpn_tile_live_masks, and sec_conv2d_nhwc_tiles still requires the background it copies from."""
    from second_amd import runtime as rt
    l = rt.lib()
    one = ctypes.c_void_p(16)            # never dereferenced: validation fails first
    inval = -1
    assert l.sec_conv2d_nhwc_tiles_lazy(one, 1, 8, 16, one, None, 128, 1, None, one, None, one, one, one, rt.SEC_BF16, None) == inval     # no lists
    assert l.sec_conv2d_nhwc_tiles_lazy(one, 1, 8, 16, one, None, 128, 1, one, one, None, None, one, one, rt.SEC_BF16, None) == inval      # no masks
    assert l.sec_conv2d_nhwc_tiles_lazy(one, 1, 8, 16, one, None, 128, 1, one, one, None, one, None, one, rt.SEC_BF16, None) == inval      # no empty-frame map
    assert l.sec_conv2d_nhwc_tiles_lazy(one, 1, 8, 16, one, None, 128, 1, one, None, None, one, one, one, rt.SEC_BF16, None) == inval      # lists without counts
    assert l.sec_conv2d_nhwc_tiles_lazy(one, 1, 8, 16, one, None, 96, 1, one, one, None, one, one, one, rt.SEC_BF16, None) == -3           # cout: unsupported
    assert l.sec_conv2d_nhwc_tiles(one, 1, 8, 16, one, None, 128, 1, one, one, None, one, rt.SEC_BF16, None) == inval                       # copying form without background
    assert l.sec_rpn_tile_live_masks(one, 1, 8, 16, 2, one, one, None, one, 1 << 20, None) == inval
    assert l.sec_rpn_tile_live_masks(one, 1, 8, 16, 9, one, one, one, one, 1 << 20, None) == -3                                             # more than 8 layers
    assert l.sec_rpn_tile_live_masks(one, 1, 8, 16, 2, one, one, one, one, 0, None) == -2                                                   # workspace too small


def test_round5_training_entry_points_validate_before_any_launch():
    """sec_heads_loss_fwd / _bwd, the 64-channel form of sec_conv2d_wgrad_nhwc and the pre-zeroed accumulator arguments: shape
    support and argument errors are decided on the host, before any launch (no GPU needed)."""
    import ctypes
    from second_amd import runtime as rt
    l = rt.lib()
    one = ctypes.c_void_p(4096)                       # a non-NULL pointer that is never dereferenced on these paths
    f17 = rt.f_arr([0.25, 2.0, 3.0, 1, 1, 1, 2, 0.2, 0, 1] + [1.0] * 7)
    assert l.sec_heads_loss_supported(64, 2, 1, 2, rt.SEC_BF16) == 1 and l.sec_heads_loss_supported(64, 2, 1, 0, rt.SEC_F16) == 1
    assert l.sec_heads_loss_supported(64, 2, 1, 2, rt.SEC_F32) == 0                       # 16-bit head tensors only
    assert l.sec_heads_loss_supported(64, 20, 10, 2, rt.SEC_BF16) == 0                    # the nuScenes multi-class head does not fit 64 channels
    assert l.sec_heads_loss_supported(128, 2, 1, 2, rt.SEC_BF16) == 0
    assert l.sec_heads_loss_workspace_bytes(4, 200, 176, 2) > 4 * 138 * 64 * 4 and l.sec_heads_loss_workspace_bytes(0, 200, 176, 2) == 0
    args = (rt.SEC_BF16, 1, 8, 16, 64, 2, 1, 2, one, one, one, one, f17)
    assert l.sec_heads_loss_fwd(None, *args, one, one, 1 << 20, None) == -1
    assert l.sec_heads_loss_fwd(one, *args, None, one, 1 << 20, None) == -1               # no out6
    assert l.sec_heads_loss_fwd(one, *args, one, one, 16, None) == -2                                   # workspace too small
    assert l.sec_heads_loss_fwd(one, rt.SEC_BF16, 1, 8, 16, 64, 4, 2, 2, one, one, one, one, f17, one, one, 1 << 20, None) == -3      # head shape not instantiated
    assert l.sec_heads_loss_bwd(one, *args, None, None, one, one, 1 << 20, 0, None) == -1  # no d_heads
    assert l.sec_heads_loss_bwd(one, *args, None, one, None, one, 1 << 20, 0, None) == -1  # no d_bias
    # one-tap weight gradient with the 64-channel gradient of the stacked heads: supported; a 3x3 with 64 output channels is not
    assert l.sec_conv2d_wgrad_workspace_bytes(4, 200, 176, 128, 64, 1) > 0 and l.sec_conv2d_wgrad_workspace_bytes(4, 200, 176, 128, 64, 3) == 0
    assert l.sec_conv2d_wgrad_nhwc(one, one, 4, 200, 176, 128, 64, 3, 1, 1, one, one, 1 << 30, rt.SEC_BF16, None) == -3
    assert l.sec_conv2d_wgrad_nhwc(one, one, 4, 200, 176, 128, 64, 1, 1, 0, one, one, 16, rt.SEC_BF16, None) == -2     # workspace too small


def test_round6_entry_points_validate_before_any_launch():
    """sec_voxelize_f32 refuses max_points > 256 up front (and reports no workspace size for it); sec_set_fp32_mode takes the two
    documented modes only; sec_heads_loss_fwd_terms validates like sec_heads_loss_fwd.  Host-side decisions: no GPU needed."""
    import ctypes
    from second_amd import runtime as rt
    l = rt.lib()
    one = ctypes.c_void_p(4096)
    assert l.sec_voxelize_workspace_bytes(1000, 1, 100, 256) > 0 and l.sec_voxelize_workspace_bytes(1000, 1, 100, 257) == 0
    rng, vs = rt.f_arr([0, -40, -3, 70.4, 40, 1]), rt.f_arr([0.05, 0.05, 0.1])
    assert l.sec_voxelize_f32(one, one, 1000, 4, 1, rng, vs, 300, 100, 0, one, one, one, one, None, 0, 0, one, 1 << 30, None) == -3
    prev = l.sec_get_fp32_mode()
    assert prev in (0, 1) and l.sec_set_fp32_mode(2) == -1 and l.sec_get_fp32_mode() == prev
    assert l.sec_set_fp32_mode(1) == 0 and l.sec_get_fp32_mode() == 1 and l.sec_set_fp32_mode(prev) == 0
    f17 = rt.f_arr([0.25, 2.0, 3.0, 1, 1, 1, 2, 0.2, 0, 1] + [1.0] * 7)
    args = (rt.SEC_BF16, 1, 8, 16, 64, 2, 1, 2, one, one, one, one, f17)
    assert l.sec_heads_loss_fwd_terms(None, *args, one, one, one, one, one, 1 << 20, None) == -1
    assert l.sec_heads_loss_fwd_terms(one, *args, one, None, None, None, one, 16, None) == -2            # workspace too small
    assert l.sec_heads_loss_fwd_terms(one, rt.SEC_F32, *args[1:], one, one, one, one, one, 1 << 20, None) == -3


def test_no_cpu_fallback():
    from second_amd import ops
    from second_amd.runtime import SecondHipError
    with pytest.raises(SecondHipError):
        ops.rulebook_subm(torch.zeros((2, 4), dtype=torch.int32), 1, (4, 4, 4))
    with pytest.raises(SecondHipError):
        ops.rotate_iou(torch.zeros((1, 5)), torch.zeros((1, 5)))


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under second.pytorch_amd/ may import, link or dlopen it."""
    pkg = os.path.join(ROOT, "second.pytorch_amd")
    bad = re.compile(r"(^|\n)\s*(from|import)\s+oracle\b|libsecond_oracle|second_oracle|orc_[a-z_]+\(")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not bad.search(src), os.path.join(dirpath, f)


def test_gather_channel_perm_is_the_plane_major_view_of_dense_view():
    """ops.gather_channel_perm: the weights sec_conv2d_nhwc_gather is packed from, `w[:, perm]`, applied to the plane-major channel
    order (z * C + c) give what `w` gives on the reference's `dense().view(N, C * D, H, W)` order (c * D + z; middle.py:206-210)."""
    import sys, os
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "second.pytorch_amd"))
    from second_amd import ops
    torch.manual_seed(0)
    c, d = 64, 2
    dense = torch.randn(2, c, d, 6, 5)
    ref_in = dense.view(2, c * d, 6, 5)                                   # channel c * D + z
    plane_major = dense.permute(0, 2, 1, 3, 4).reshape(2, d * c, 6, 5)    # channel z * C + c
    w = torch.randn(8, c * d, 3, 3)
    perm = ops.gather_channel_perm(c, d)
    assert sorted(perm.tolist()) == list(range(c * d))
    a = torch.nn.functional.conv2d(ref_in, w, None, 1, 1)
    b = torch.nn.functional.conv2d(plane_major, w[:, perm], None, 1, 1)
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-3)      # same products, another summation order
