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


class _BatchNormPlanarFn(Function):
    """nn.BatchNorm3d in training mode on the planar HIP kernels (csrc/planar_ops.hip); returns y and the batch statistics
    (mean, unbiased variance) the module folds into its running estimates."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, planes=None):
        # planes = (rows,): x (any shape, contiguous) is normalised as the tensor [1, rows, numel / rows] — InstanceNorm / GroupNorm rows — and y comes back in x's OWN
        # shape.  The reshape happens inside the Function on purpose: a caller-side ``y.view_as(x)`` makes y a view of this Function's output, and the in-place
        # LeakyReLU / ``out += residual`` that follow in UnetResBlock then run their backward through CopySlices + AsStridedBackward — four extra full-resolution
        # copies per site (0.4 ms of the full net's iteration, scripts/prof_net_copies.py).
        ctx.shape = None
        out = None
        if planes is not None:
            ctx.shape = tuple(x.shape)
            out = torch.empty_like(x)            # (y is allocated in x's shape and handed to the kernel: a ``.view()`` of the op's result would again be a view)
            x = x.view(1, int(planes[0]), -1)
        y, stats = ops.batchnorm_planar_forward(x, weight, bias, eps, out=out)
        ctx.save_for_backward(x, weight, stats)
        ctx.affine = weight is not None
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _gstats):
        x, weight, stats = ctx.saved_tensors
        gy = gy.contiguous()
        if ctx.shape is not None:
            gy = gy.view(x.shape)
        gx, gw, gb = ops.batchnorm_planar_backward(gy, x, weight, stats, ctx.affine)
        return (gx if ctx.shape is None else gx.view(ctx.shape)), gw, gb, None, None


def batch_norm_train(x, weight, bias, eps=1e-5, planes=None):
    return _BatchNormPlanarFn.apply(x, weight, bias, eps, planes)


class _PointwisePlanarFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight)
        return ops.pointwise_planar_forward(x, weight, bias)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        need = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        gx, gw, gb = ops.pointwise_planar_backward(x, weight, gy.contiguous(), need)
        return gx, gw, gb


def pointwise_planar(x, weight, bias=None):
    """1x1x1 conv on planar fp32 tensors with few channels; callers check ops.pointwise_planar_supported first."""
    return _PointwisePlanarFn.apply(x, weight, bias)
