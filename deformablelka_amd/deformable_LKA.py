"""2-D D-LKA modules — ``DeformConv``, ``deformable_LKA``, ``deformable_LKA_Attention`` with the constructor /
forward signatures and ``state_dict`` keys of 2D/deformable_LKA/deformable_LKA.py:5-30,90-104,124-140.

``deformable_LKA_Attention.forward`` runs the whole block as ONE C-ABI call per direction
(``dlka_lka2d_attention_forward/backward``); the inner modules keep working standalone through the per-op kernels.
"""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import nn_ops, ops
from .tv_ops import DeformConv2d


class DeformConv(nn.Module):
    """deformable_LKA.py:5-30: dense offset net (2*kh*kw channels) + depthwise-capable deformable conv (bias=False)."""

    def __init__(self, in_channels, groups, kernel_size=(3, 3), padding=1, stride=1, dilation=1, bias=True):
        super().__init__()
        self.offset_net = nn.Conv2d(in_channels=in_channels, out_channels=2 * kernel_size[0] * kernel_size[1],
                                    kernel_size=kernel_size, padding=padding, stride=stride, dilation=dilation, bias=True)
        self.deform_conv = DeformConv2d(in_channels=in_channels, out_channels=in_channels, kernel_size=kernel_size,
                                        padding=padding, groups=groups, stride=stride, dilation=dilation, bias=False)

    def forward(self, x):
        o = self.offset_net
        offsets = nn_ops.conv2d(x, o.weight, o.bias, o.stride, o.padding, o.dilation, o.groups)
        return self.deform_conv(x, offsets)


class deformable_LKA(nn.Module):
    """deformable_LKA.py:90-104."""

    def __init__(self, dim):
        super().__init__()
        self.conv0 = DeformConv(dim, kernel_size=(5, 5), padding=2, groups=dim)
        self.conv_spatial = DeformConv(dim, kernel_size=(7, 7), stride=1, padding=9, groups=dim, dilation=3)
        self.conv1 = nn.Conv2d(dim, dim, 1)

    def forward(self, x):
        u = x
        attn = self.conv0(x)
        attn = self.conv_spatial(attn)
        c = self.conv1
        attn = nn_ops.conv2d(attn, c.weight, c.bias)
        return u * attn


class _LKA2dAttentionFn(Function):
    @staticmethod
    def forward(ctx, x, *params):
        y, saved = ops.lka2d_attention_forward(x, params)
        ctx.save_for_backward(x, saved, *params)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, saved, *params = ctx.saved_tensors
        gx, grads = ops.lka2d_attention_backward(x, params, gy, saved)
        return (gx, *grads)


class deformable_LKA_Attention(nn.Module):
    """deformable_LKA.py:124-140."""

    def __init__(self, d_model):
        super().__init__()
        self.proj_1 = nn.Conv2d(d_model, d_model, 1)
        self.activation = nn.GELU()
        self.spatial_gating_unit = deformable_LKA(d_model)
        self.proj_2 = nn.Conv2d(d_model, d_model, 1)

    def block_params(self):
        """The 12 tensors in ``dlka_lka2d_params`` order (include/dlka.h)."""
        s = self.spatial_gating_unit
        return (self.proj_1.weight, self.proj_1.bias, s.conv0.offset_net.weight, s.conv0.offset_net.bias,
                s.conv0.deform_conv.weight, s.conv_spatial.offset_net.weight, s.conv_spatial.offset_net.bias,
                s.conv_spatial.deform_conv.weight, s.conv1.weight, s.conv1.bias, self.proj_2.weight, self.proj_2.bias)

    def forward(self, x):
        # Autocast policy (the reference registers none): inside torch.autocast(dtype=torch.bfloat16) the block takes bf16 activations with fp32
        # parameters / offsets / accumulation where the DLKA_BF16 path covers the width (BASELINE.json config 2: bf16 training).
        act = ops.autocast_activation_dtype(x)
        if act != x.dtype and act == torch.bfloat16 and self.proj_1.weight.dtype == torch.float32 and ops.lka2d_bf16_supported(x.shape[1]):
            x = x.to(act)
        return _LKA2dAttentionFn.apply(x, *self.block_params())
