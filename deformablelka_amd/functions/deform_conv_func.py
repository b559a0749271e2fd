"""``DeformConvFunction`` — same call signature, saved tensors and return arity as the reference's
3D/dcn/functions/deform_conv_func.py:15-56 (copy at 3D/d_lka_former/network_architecture/synapse/deform_conv_func.py)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _triple

from .. import D3D


class DeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, weight, bias, stride, padding, dilation, group, deformable_groups, im2col_step):
        ctx.stride = _triple(stride)
        ctx.padding = _triple(padding)
        ctx.dilation = _triple(dilation)
        ctx.kernel_size = _triple(weight.shape[2:5])
        ctx.group = group
        ctx.deformable_groups = deformable_groups
        ctx.im2col_step = im2col_step
        output = D3D.deform_conv_forward(input, weight, bias, offset, *ctx.kernel_size, *ctx.stride, *ctx.padding,
                                         *ctx.dilation, ctx.group, ctx.deformable_groups, ctx.im2col_step)
        ctx.save_for_backward(input, offset, weight, bias)
        return output

    @staticmethod
    @once_differentiable  # reference: deform_conv_func.py:39 (no double backward)
    def backward(ctx, grad_output):
        input, offset, weight, bias = ctx.saved_tensors
        grad_input, grad_offset, grad_weight, grad_bias = D3D.deform_conv_backward(
            input, weight, bias, offset, grad_output, *ctx.kernel_size, *ctx.stride, *ctx.padding, *ctx.dilation,
            ctx.group, ctx.deformable_groups, ctx.im2col_step)
        return grad_input, grad_offset, grad_weight, grad_bias, None, None, None, None, None, None
