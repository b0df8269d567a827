"""Autograd wrappers (spconv/functional.py upstream: SparseConvFunction / SubMConvFunction)."""
import os

import torch

from second_amd import ops as _ops

# fp32 features AND fp32 weights with a gradient wanted (the reference's default training precision, train.py:232-235): forward and
# backward run in exact fp32 arithmetic whatever the process-wide inference mode is (ops.set_fp32_mode) -- a training step whose
# forward silently computed with 16-bit operand halves while its backward used fp32 would be neither.  SEC_FP32_TRAIN_MODE=split16
# opts in to the faster split-operand forward / data gradient (products good to ~2^-16).
TRAIN_FP32_MODE = os.environ.get("SEC_FP32_TRAIN_MODE", "exact")


class IndiceConvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, weight, rulebook, packed):
        ctx.rulebook = rulebook
        ctx.master_dtype = weight.dtype
        ctx.packed_dgrad = None
        ctx.dw0 = None
        ctx.fp32_mode = None
        if features.dtype == torch.float32 and weight.dtype == torch.float32 and features.is_cuda and any(ctx.needs_input_grad[:2]):
            ctx.fp32_mode = TRAIN_FP32_MODE
            if ctx.fp32_mode == "exact":
                packed = None
        if weight.dtype != features.dtype:
            # mixed precision (fp32 master weight, 16-bit features): the cast copy is made HERE, outside autograd -- the weight
            # gradient comes out of the kernels in fp32 and goes straight to the master weight (no fp32 -> 16-bit -> fp32 round
            # trip: two launches per layer and a rounding of dW less)
            if (weight.is_cuda and weight.dtype == torch.float32 and features.dtype in (torch.bfloat16, torch.float16)
                    and getattr(rulebook, "subm", None) is not None):
                # ... together with BOTH MFMA images of the step, one launch (the backward would pack the transposed one again)
                # ... and the zeroed accumulator of the backward's weight gradient (otherwise a memset node per layer and step)
                want_dw = bool(ctx.needs_input_grad[1])          # (grad mode is off inside forward(): ask the context)
                res = _ops.pack_weight_train(weight.detach().contiguous(), features.dtype, subm=rulebook.nbr_in is None, zero_grad=want_dw)
                weight, packed, ctx.packed_dgrad = res[:3]
                ctx.dw0 = res[3] if want_dw else None
            else:
                weight = weight.detach().to(features.dtype)
                packed = _ops.pack_weight(weight.contiguous()) if weight.is_cuda and weight.dtype != torch.float32 else None
        ctx.save_for_backward(features, weight)
        with _ops.fp32_mode(ctx.fp32_mode):
            return _ops.indice_conv(features.contiguous(), weight.contiguous(), rulebook.nbr_out, rulebook.num_out,
                                    packed=packed, num_out_dev=rulebook.num_out_dev)

    @staticmethod
    def backward(ctx, grad_out):
        features, weight = ctx.saved_tensors
        rb = ctx.rulebook
        if not rb.subm and rb.nbr_in is None:
            raise RuntimeError("this rulebook was built with autograd disabled (no input-major table); rebuild it "
                               "under torch.enable_grad() to back-propagate through a strided sparse conv")
        with _ops.fp32_mode(ctx.fp32_mode):
            dfeat, dw = _ops.indice_conv_backward(features.contiguous(), weight.contiguous(), rb.nbr_out, rb.nbr_in,
                                                  grad_out.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                                  dweight_dtype=ctx.master_dtype, packed_dgrad=ctx.packed_dgrad, dweight_out=ctx.dw0)
        ctx.dw0 = None                  # consumed: a second backward through this node (retain_graph) zeroes its own
        return dfeat, dw, None, None


def indice_conv(features, weight, rulebook, packed=None):
    return IndiceConvFunction.apply(features, weight, rulebook, packed)


indice_subm_conv = indice_conv


class SparseToDenseFunction(torch.autograd.Function):
    """dense() with a gradient (upstream: scatter_nd under autograd, spconv/__init__.py SparseConvTensor.dense)."""

    @staticmethod
    def forward(ctx, features, indices, batch_size, spatial_shape, num_dev=None, channels_last_2d=False):
        ctx.save_for_backward(indices)
        ctx.num_dev = num_dev              # static capacity: rows past num_dev[0] are neither scattered nor gathered back
        ctx.depth = int(spatial_shape[0]) if channels_last_2d else 0
        return _ops.sparse_to_dense(features, indices, batch_size, spatial_shape, channels_last_2d=channels_last_2d, num_dev=num_dev)

    @staticmethod
    def backward(ctx, grad):
        (indices,) = ctx.saved_tensors
        # channels_last_2d: grad is the [B, C * D, H, W] gradient of the RPN input, gathered in whatever strides it arrives in
        return _ops.dense_to_sparse(grad, indices, num_dev=ctx.num_dev, depth=ctx.depth), None, None, None, None, None


class PillarScatterFunction(torch.autograd.Function):
    """PointPillarsScatter with a gradient (pointpillars.py:444-476 relies on index assignment under autograd)."""

    @staticmethod
    def forward(ctx, features, coords, batch_size, ny, nx, channels_last=False):
        ctx.save_for_backward(coords)
        return _ops.pillar_scatter(features, coords, batch_size, ny, nx, channels_last=channels_last)

    @staticmethod
    def backward(ctx, grad):
        (coords,) = ctx.saved_tensors
        return _ops.dense_to_sparse(grad, coords), None, None, None, None, None
