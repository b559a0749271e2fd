"""Twin of ``torchvision.ops.deform_conv2d`` / ``torchvision.ops.DeformConv2d`` (torchvision==0.12.0, the version the
reference pins at 2D/requirements.txt:69) for the call sites in 2D/deformable_LKA/deformable_LKA.py:18-30 —
same constructor, ``forward(input, offset, mask=None)`` signature and parameter names (``weight``, ``bias``), backed
by the HIP kernels.  ``mask`` (DCNv2 modulation) is not used anywhere in the reference and is rejected."""
import math

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn import init
from torch.nn.modules.utils import _pair

from . import ops


class _DeformConv2dFn(Function):
    @staticmethod
    def forward(ctx, input, offset, weight, bias, stride, padding, dilation):
        ctx.cfg = (_pair(stride), _pair(padding), _pair(dilation))
        ctx.has_bias = bias is not None
        ctx.save_for_backward(input, offset, weight)
        return ops.deform_conv2d_forward(input, offset, weight, bias, *ctx.cfg)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, weight = ctx.saved_tensors
        gi, go, gw, gb = ops.deform_conv2d_backward(input, offset, weight, grad_output, *ctx.cfg, with_bias=ctx.has_bias,
                                                    need=tuple(ctx.needs_input_grad[:3]))
        return gi, go, gw, gb, None, None, None


def deform_conv2d(input, offset, weight, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), mask=None):
    if mask is not None:
        raise NotImplementedError("deform_conv2d: modulation mask is not used by the reference D-LKA path and is not implemented")
    return _DeformConv2dFn.apply(input, offset, weight, bias, stride, padding, dilation)


class DeformConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        if in_channels % groups != 0:
            raise ValueError("in_channels must be divisible by groups")
        if out_channels % groups != 0:
            raise ValueError("out_channels must be divisible by groups")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.dilation = _pair(dilation)
        self.groups = groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, self.kernel_size[0], self.kernel_size[1]))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def forward(self, input, offset, mask=None):
        return deform_conv2d(input, offset, self.weight, self.bias, stride=self.stride, padding=self.padding,
                             dilation=self.dilation, mask=mask)
