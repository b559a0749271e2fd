"""autograd.Functions over the HIP conv / GELU kernels, used where the reference calls nn.Conv3d / nn.Conv2d /
nn.GELU inside the D-LKA path (so that those ops also run on the hand-written kernels, not MIOpen)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops


class _Conv3dFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, groups):
        ctx.cfg = (stride, padding, dilation, groups)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight)
        return ops.conv3d_forward(x, weight, bias, stride, padding, dilation, groups)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        need = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        gi, gw, gb = ops.conv3d_backward(x, weight, gy, *ctx.cfg, need=need)
        return gi, gw, gb, None, None, None, None


def conv3d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return _Conv3dFn.apply(x, weight, bias, stride, padding, dilation, groups)


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """2-D conv through the same kernels (D = kd = 1)."""
    s = (1,) + tuple(ops._pair(stride))
    p = (0,) + tuple(ops._pair(padding))
    d = (1,) + tuple(ops._pair(dilation))
    y = _Conv3dFn.apply(x.unsqueeze(2), weight.unsqueeze(2), bias, s, p, d, groups)
    return y.squeeze(2)


class _GeluFn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gelu_forward(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return ops.gelu_backward(x, gy)


def gelu(x):
    return _GeluFn.apply(x)
