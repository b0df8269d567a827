"""ctypes binding of libsecond_hip.so (C ABI: include/second_hip.h).

Fails loudly: no library -> RuntimeError at first use; no silent CPU path anywhere.
"""
import contextlib
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libsecond_hip.so")
LIB_PATH = os.environ.get("SEC_HIP_LIB", LIB_PATH)   # A/B builds: point at another libsecond_hip.so

SEC_F32, SEC_F16, SEC_BF16 = 0, 1, 2
ABI_VERSION = 9          # include/second_hip.h SEC_ABI_VERSION the argtypes in lib() were written for
_DTYPES = {torch.float32: SEC_F32, torch.float16: SEC_F16, torch.bfloat16: SEC_BF16}
_ERRORS = {-1: "SEC_E_INVALID (bad argument)", -2: "SEC_E_WORKSPACE (workspace too small)",
           -3: "SEC_E_UNSUPPORTED", -4: "SEC_E_LAUNCH (HIP error)"}

# every symbol include/second_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "sec_abi_version", "sec_last_error", "sec_last_kernel_name", "sec_tensors_checksum", "sec_voxelize_workspace_bytes", "sec_voxelize_f32", "sec_simple_voxel_f32", "sec_rows_differ_f32",
    "sec_rulebook_workspace_bytes", "sec_rulebook_subm3d", "sec_rulebook_subm3d_after_conv", "sec_rulebook_subm3d_after_voxelize",
    "sec_rulebook_conv3d_build",
    "sec_rulebook_conv3d_tables", "sec_rulebook_sorted_workspace_bytes", "sec_rulebook_conv3d_build_sorted",
    "sec_rulebook_conv3d_tables_sorted", "sec_rulebook_subm3d_after_conv_sorted", "sec_rulebook_chain_workspace_bytes",
    "sec_rulebook_chain_sorted", "sec_conv_output_shape", "sec_packed_weight_bytes", "sec_packed_weight_x3_bytes",
    "sec_pack_conv_weight", "sec_indice_conv_fwd", "sec_indice_conv_fwd_plan", "sec_indice_conv_set_variant", "sec_indice_conv_bwd_workspace_bytes", "sec_indice_conv_bwd", "sec_pack_conv_weight_train", "sec_sparse_to_dense", "sec_dense_to_sparse", "sec_sparse_site_map", "sec_sparse_site_map_sorted", "sec_conv2d_nhwc_gather", "sec_conv2d_nhwc_rows", "sec_conv2d_nhwc_into", "sec_rpn_tile_live_workspace_bytes", "sec_rpn_tile_live", "sec_rpn_tile_live_masks", "sec_conv2d_nhwc_tiles", "sec_conv2d_nhwc_tiles_lazy", "sec_conv2d_nhwc_tiles_tail", "sec_conv1x1_chain_nhwc_tiles",
    "sec_pillar_scatter", "sec_pfn_fwd", "sec_pfn_fwd_slots", "sec_pfn_train_workspace_bytes", "sec_pfn_train_fwd", "sec_pfn_train_bwd", "sec_block_filter_workspace_bytes",
    "sec_voxel_block_filter_f32", "sec_bias_act_nhwc", "sec_conv2d_packed_weight_bytes",
    "sec_conv2d_pack_weight", "sec_conv2d_nhwc", "sec_split_f32_bf16x2", "sec_merge_bf16x2_f32", "sec_conv2d_nhwc_x3", "sec_conv2d_nhwc_x3_tiles", "sec_conv1x1_chain_x3", "sec_conv1x1_chain_nhwc", "sec_rotate_iou_f32", "sec_nms_workspace_bytes", "sec_nms_sorted_f32",
    "sec_predict_select", "sec_predict_select_lazy", "sec_predict_decode", "sec_predict_decode_lazy", "sec_predict_finalize",
    "sec_assign_targets_workspace_bytes", "sec_assign_targets_f32", "sec_assign_targets_per_class_f32",
    "sec_second_loss_workspace_bytes", "sec_second_loss_f32", "sec_heads_loss_supported", "sec_heads_loss_workspace_bytes",
    "sec_heads_loss_fwd", "sec_heads_loss_fwd_terms", "sec_heads_loss_bwd", "sec_set_fp32_mode", "sec_get_fp32_mode",
    "sec_conv2d_pack_weight_train", "sec_conv2d_pack_weight_train_multi", "sec_pack_conv_weight_train_multi", "sec_conv2d_wgrad_workspace_bytes", "sec_conv2d_wgrad_nhwc", "sec_bn_train_workspace_bytes", "sec_bn_relu_fwd_nhwc",
    "sec_bn_relu_bwd_nhwc", "sec_flat_adamw_workspace_bytes", "sec_flat_adamw_f32", "sec_flat_adamw_dev_f32",
]

_lib = None


class SecondHipError(RuntimeError):
    pass


# ---- fork guard ---------------------------------------------------------------------------------------------------------
# The reference forks its DataLoader workers (second/pytorch/train.py:262-277) and they call VoxelGeneratorV2.generate
# (second/data/preprocess.py:301-316), which runs on the GPU here.  A HIP context does not survive fork(): a child of a
# process that already initialised HIP can neither use the parent's context nor create its own (the runtime's state is
# inherited half-initialised; the first call hangs or fails with "Cannot re-initialize CUDA in forked subprocess").  Such a
# child is marked at fork time and every entry into the library raises with instructions instead.
_forked_after_gpu_init = False
_parent_used_gpu = False


def _gpu_in_use():
    return _lib is not None or torch.cuda.is_initialized()


def _before_fork():
    global _parent_used_gpu
    _parent_used_gpu = _gpu_in_use()


def _after_fork_in_child():
    global _forked_after_gpu_init
    if _parent_used_gpu:
        _forked_after_gpu_init = True


os.register_at_fork(before=_before_fork, after_in_child=_after_fork_in_child)

FORK_MESSAGE = (
    "second.pytorch_amd: this process was fork()ed from a parent that had already initialised the GPU; a HIP context does not "
    "survive fork(), so spconv ops (VoxelGeneratorV2.generate, rulebooks, NMS) cannot run here. `import spconv` makes "
    "torch.utils.data.DataLoader(num_workers > 0) default to the 'spawn' start method -- construct the DataLoader through "
    "torch.utils.data.DataLoader after importing spconv (the unmodified second/pytorch/train.py does), pass "
    "multiprocessing_context='spawn' yourself, use num_workers=0, or voxelise in the main process with "
    "VoxelGeneratorV2.generate_device / SecondDetector.forward_points.")


def check_not_forked():
    if _forked_after_gpu_init:
        raise SecondHipError(FORK_MESSAGE)


def lib():
    """The loaded library.  Raises if it has not been built (python second.pytorch_amd/build.py)."""
    global _lib
    check_not_forked()
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SecondHipError(
                f"{LIB_PATH} is missing: build it with `python second.pytorch_amd/build.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        l = ctypes.CDLL(LIB_PATH)
        if l.sec_abi_version() != ABI_VERSION:       # the argtypes below are written for exactly this header revision
            raise SecondHipError(f"{LIB_PATH} has ABI version {l.sec_abi_version()}, these bindings expect {ABI_VERSION}: "
                                 "rebuild it (python second.pytorch_amd/build.py --force)")
        for name in ("sec_voxelize_workspace_bytes", "sec_rulebook_workspace_bytes", "sec_rulebook_sorted_workspace_bytes",
                     "sec_rulebook_chain_workspace_bytes",
                     "sec_packed_weight_bytes", "sec_nms_workspace_bytes", "sec_block_filter_workspace_bytes",
                     "sec_conv2d_packed_weight_bytes", "sec_indice_conv_bwd_workspace_bytes",
                     "sec_assign_targets_workspace_bytes", "sec_second_loss_workspace_bytes", "sec_heads_loss_workspace_bytes",
                     "sec_conv2d_wgrad_workspace_bytes", "sec_bn_train_workspace_bytes", "sec_pfn_train_workspace_bytes",
                     "sec_flat_adamw_workspace_bytes"):
            getattr(l, name).restype = ctypes.c_size_t
        if os.environ.get("SEC_FP32_MODE", "").lower() == "exact":      # process default of ops.set_fp32_mode
            l.sec_set_fp32_mode(1)
        l.sec_last_error.restype = ctypes.c_char_p
        l.sec_last_kernel_name.restype = ctypes.c_char_p
        l.sec_conv_output_shape.restype = None
        vp, ci, cf, sz, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_int64
        l.sec_voxelize_workspace_bytes.argtypes = [ci, ci, ci, ci]
        l.sec_voxelize_f32.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, ci, ci, vp, sz, vp]
        l.sec_tensors_checksum.argtypes = [vp, vp, ci, vp, vp]
        l.sec_simple_voxel_f32.argtypes = [vp, vp, ci, vp, ci, ci, ci, vp, ci, vp]
        l.sec_rows_differ_f32.argtypes = [vp, ctypes.c_longlong, vp, ctypes.c_longlong, vp, vp]
        l.sec_rulebook_workspace_bytes.argtypes = [ci, ci, ci]
        l.sec_rulebook_subm3d.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        l.sec_rulebook_subm3d_after_conv.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, sz, ci, vp, vp, vp, ci, vp]
        l.sec_rulebook_subm3d_after_voxelize.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, sz, ci, ci, ci, vp, vp]
        l.sec_rulebook_conv3d_build.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, vp, ci, vp, ci, vp, vp, sz, vp]
        l.sec_rulebook_conv3d_tables.argtypes = [ci, vp, vp, vp, ci, vp, ci, vp, ci, vp, vp, vp, sz, vp]
        l.sec_rulebook_sorted_workspace_bytes.argtypes = [ci, ci, ci, vp]
        l.sec_rulebook_conv3d_build_sorted.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp, ci, vp, vp, ctypes.c_longlong,
                                                       vp, sz, vp, sz, vp]
        l.sec_rulebook_conv3d_tables_sorted.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, vp, ci, vp, ci, vp, vp, vp, sz, vp]
        l.sec_rulebook_subm3d_after_conv_sorted.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, ci, vp, sz, vp]
        l.sec_rulebook_chain_workspace_bytes.argtypes = [ci, ci, vp]
        l.sec_rulebook_chain_sorted.argtypes = [vp, ci, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, ci, ci, ci, vp, vp, vp, sz, vp]
        l.sec_conv_output_shape.argtypes = [vp] * 6
        l.sec_packed_weight_bytes.argtypes = [ci, ci, ci, ci]
        l.sec_pack_conv_weight.argtypes = [vp, ci, ci, ci, ci, vp, vp]
        l.sec_indice_conv_fwd.argtypes = [vp, ci, ci, vp, vp, ci, ci, vp, ci, vp, vp, vp, ci, vp, ci, ci, vp]
        l.sec_packed_weight_x3_bytes.argtypes = [ci, ci, ci]
        l.sec_packed_weight_x3_bytes.restype = sz
        l.sec_indice_conv_fwd_plan.argtypes = [ci] * 7
        l.sec_indice_conv_set_variant.argtypes = [ci]
        l.sec_indice_conv_bwd.argtypes = [vp, ci, ci, vp, ci, ci, vp, vp, ci, vp, vp, vp, ci, vp, sz, vp, ci, vp]
        l.sec_pack_conv_weight_train.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp]
        l.sec_indice_conv_bwd_workspace_bytes.argtypes = [ci, ci, ci, ci]
        l.sec_sparse_to_dense.argtypes = [vp, vp, ci, ci, vp, vp, sz, i64, i64, i64, i64, i64, ci, vp]
        l.sec_dense_to_sparse.argtypes = [vp, vp, ci, ci, vp, vp, i64, i64, i64, i64, i64, ci, vp]
        l.sec_pillar_scatter.argtypes = [vp, vp, ci, ci, vp, vp, sz, i64, i64, i64, i64, ci, vp]
        l.sec_pfn_fwd.argtypes = [vp, vp, vp, ci, vp, ci, ci, vp, vp, vp, ci, cf, cf, cf, cf, vp, ci, vp]
        l.sec_pfn_fwd_slots.argtypes = [vp, vp, sz, ci, ci, ci, ci, ci, vp, vp, ci, vp, vp, vp, vp, ci, cf, cf, cf, cf, vp, ci, vp]
        l.sec_pfn_train_workspace_bytes.argtypes = [ci, ci]
        l.sec_pfn_train_fwd.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp, cf, cf, vp, vp, ci, cf, cf, cf, cf, vp, vp, vp, vp, sz, vp]
        l.sec_pfn_train_bwd.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp, ci, cf, cf, cf, cf, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        l.sec_block_filter_workspace_bytes.argtypes = [ci, ci, ci, ci, ci]
        l.sec_voxel_block_filter_f32.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, cf, cf, vp, vp, vp, vp, vp, sz, vp]
        l.sec_bias_act_nhwc.argtypes = [vp, vp, sz, ci, ci, ci, vp]
        l.sec_conv2d_packed_weight_bytes.argtypes = [ci, ci, ci, ci]
        l.sec_conv2d_pack_weight.argtypes = [vp, ci, ci, ci, ci, vp, vp]
        l.sec_conv2d_nhwc.argtypes = [vp, ci, ci, ci, ci, vp, vp, ci, ci, ci, ci, ci, vp, ci, vp]
        l.sec_split_f32_bf16x2.argtypes = [vp, ctypes.c_longlong, vp, vp, vp]
        l.sec_merge_bf16x2_f32.argtypes = [vp, vp, ctypes.c_longlong, vp, vp]
        l.sec_conv2d_nhwc_x3.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, ci, vp, vp, vp]
        l.sec_conv1x1_chain_x3.argtypes = [vp, vp, ctypes.c_longlong, vp, vp, ci, vp, vp, ci, vp, vp]
        l.sec_conv2d_nhwc_x3_tiles.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        l.sec_sparse_site_map.argtypes = [vp, ci, vp, ci, ci, ci, ci, vp, vp]
        l.sec_sparse_site_map_sorted.argtypes = [vp, ctypes.c_size_t, vp, ci, ci, ci, ci, ci, vp, vp]
        l.sec_conv2d_nhwc_gather.argtypes = [vp, i64, vp, ci, ci, ci, vp, vp, ci, ci, vp, vp, vp, vp, ci, vp]
        l.sec_conv2d_nhwc_rows.argtypes = [vp, i64, vp, ci, ci, ci, ci, vp, vp, ci, ci, ci, ci, ci, vp, ci, vp]
        l.sec_conv2d_nhwc_into.argtypes = [vp, ci, ci, ci, ci, vp, vp, ci, ci, ci, ci, ci, vp, ci, ci, ci, vp]
        l.sec_conv1x1_chain_nhwc.argtypes = [vp, ctypes.c_longlong, vp, vp, ci, vp, vp, ci, vp, ci, vp]
        l.sec_conv1x1_chain_nhwc_tiles.argtypes = [vp, ci, ci, ci, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp, ci, vp]
        l.sec_rpn_tile_live_workspace_bytes.argtypes = [ci, ci, ci]
        l.sec_rpn_tile_live_workspace_bytes.restype = sz
        l.sec_rpn_tile_live.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, sz, vp]
        l.sec_conv2d_nhwc_tiles.argtypes = [vp, ci, ci, ci, vp, vp, ci, ci, vp, vp, vp, vp, ci, vp]
        l.sec_rpn_tile_live_masks.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp, sz, vp]
        l.sec_conv2d_nhwc_tiles_lazy.argtypes = [vp, ci, ci, ci, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, ci, vp]
        l.sec_conv2d_nhwc_tiles_tail.argtypes = [vp, ci, ci, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp, ci, vp, ci, vp]
        l.sec_rotate_iou_f32.argtypes = [vp, ci, vp, ci, ci, vp, vp]
        l.sec_nms_workspace_bytes.argtypes = [ci, ci]
        l.sec_nms_sorted_f32.argtypes = [vp, vp, ci, ci, ci, cf, ci, ci, cf, ci, vp, vp, vp, sz, vp]
        l.sec_predict_select.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, cf, vp, vp, vp, vp, vp, ci, vp]
        l.sec_predict_decode.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, ci, vp, vp, vp, ci, vp]
        l.sec_predict_select_lazy.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, cf, vp, vp, vp, vp, vp, ci, vp, vp, vp]
        l.sec_predict_decode_lazy.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, vp]
        l.sec_predict_finalize.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, cf, ci, vp, vp, vp, vp, vp, vp]
        l.sec_assign_targets_workspace_bytes.argtypes = [ci, ci, ci]
        l.sec_assign_targets_f32.argtypes = [vp, ci, vp, vp, vp, vp, ci, ci, cf, cf, vp, vp, vp, vp, sz, vp]
        l.sec_assign_targets_per_class_f32.argtypes = [vp, ci, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        l.sec_second_loss_workspace_bytes.argtypes = [ci, ci]
        l.sec_second_loss_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, sz, vp]
        l.sec_heads_loss_supported.argtypes = [ci] * 5
        l.sec_heads_loss_workspace_bytes.argtypes = [ci] * 4
        l.sec_heads_loss_fwd.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        l.sec_set_fp32_mode.argtypes = [ci]
        l.sec_get_fp32_mode.argtypes = []
        l.sec_heads_loss_fwd_terms.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        l.sec_heads_loss_bwd.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, ci, vp]
        ll = ctypes.c_longlong
        l.sec_conv2d_pack_weight_train.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp]
        l.sec_conv2d_pack_weight_train_multi.argtypes = [ci, vp, vp, vp, vp, ci, vp, vp, vp]
        l.sec_pack_conv_weight_train_multi.argtypes = [ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp]
        l.sec_conv2d_wgrad_workspace_bytes.argtypes = [ci] * 6
        l.sec_conv2d_wgrad_nhwc.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, sz, ci, vp]
        l.sec_bn_train_workspace_bytes.argtypes = [ci]
        l.sec_bn_relu_fwd_nhwc.argtypes = [vp, ll, ci, vp, vp, cf, cf, vp, vp, ci, vp, vp, vp, vp, sz, ci, vp, vp]
        l.sec_bn_relu_bwd_nhwc.argtypes = [vp, vp, ll, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, sz, ci, vp, vp]
        l.sec_flat_adamw_workspace_bytes.argtypes = []
        l.sec_flat_adamw_f32.argtypes = [vp, vp, vp, vp, ll, cf, cf, cf, cf, cf, cf, vp, vp, vp, sz, vp]
        l.sec_flat_adamw_dev_f32.argtypes = [vp, vp, vp, vp, ll, vp, vp, vp, vp, sz, vp]
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        detail = lib().sec_last_error().decode() if rc == -4 else ""
        raise SecondHipError(f"{what} failed: {_ERRORS.get(rc, rc)} {detail}")


def dtype_code(dt):
    try:
        return _DTYPES[dt]
    except KeyError:
        raise SecondHipError(f"unsupported feature dtype {dt}") from None


def require_gpu(*tensors):
    check_not_forked()
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SecondHipError(
                "second_amd ops are GPU only (got a CPU tensor): there is no CPU path by decision (INTEGRATION.md section 1, CPU tensors); "
                "move the tensors to the MI355X -- the reference does with example_convert_to_torch(..., device), train.py:24-55")


@contextlib.contextmanager
def capture_guard():
    """No Python garbage collection inside a stream capture.  Destroying a hipGraph (``torch.cuda.CUDAGraph.__del__`` of an object an earlier
    capture left in a reference cycle) while a stream of the process is capturing fails with hipErrorStreamCaptureUnsupported -- raised from
    a destructor, i.e. ``terminate`` -- and the collector runs whenever an allocation crosses its threshold (torch >= 2.9 no longer collects
    before a capture unless ``torch.compiler.config.force_cudagraph_gc``).  Collect first, then keep the collector off until the capture ends."""
    import gc
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def ptr(t):
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def i3(v):
    if isinstance(v, int):
        v = (v, v, v)
    v = [int(x) for x in v]
    assert len(v) == 3, v
    return (ctypes.c_int * 3)(*v)


def f_arr(v):
    v = [float(x) for x in v]
    return (ctypes.c_float * len(v))(*v)


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
