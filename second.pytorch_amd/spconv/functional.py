"""Autograd wrappers (spconv/functional.py upstream: SparseConvFunction / SubMConvFunction)."""
import torch

from second_amd import ops as _ops


class IndiceConvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, weight, rulebook, packed):
        ctx.rulebook = rulebook
        ctx.save_for_backward(features, weight)
        return _ops.indice_conv(features.contiguous(), weight.contiguous(), rulebook.nbr_out, rulebook.num_out,
                                packed=packed, num_out_dev=rulebook.num_out_dev)

    @staticmethod
    def backward(ctx, grad_out):
        features, weight = ctx.saved_tensors
        rb = ctx.rulebook
        dfeat, dw = _ops.indice_conv_backward(features.contiguous(), weight.contiguous(), rb.nbr_out, rb.nbr_in,
                                              grad_out.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dfeat, dw, None, None


def indice_conv(features, weight, rulebook, packed=None):
    return IndiceConvFunction.apply(features, weight, rulebook, packed)


indice_subm_conv = indice_conv
