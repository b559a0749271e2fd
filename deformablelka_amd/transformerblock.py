"""3-D D-LKA modules — ``LKA3d_deform`` and ``LKA_Attention3d_deform`` with the constructor / forward signatures and
``state_dict`` keys of 3D/d_lka_former/network_architecture/synapse/transformerblock.py:634-673.

``LKA_Attention3d_deform.forward(x, B, C, H, W, D)`` runs the whole block (proj_1, GELU, dw 5^3, dw 7^3 dil 3,
offset-predict conv, deformable 3^3 conv, conv1, gate, proj_2, residual) as ONE C-ABI call per direction
(``dlka_lka3d_attention_forward/backward``).  ``LKA3d_deform`` alone runs through the per-op kernels.
"""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import nn_ops, ops
from .modules.deform_conv import DeformConvPack


class LKA3d_deform(nn.Module):
    """transformerblock.py:634-652."""

    def __init__(self, dim):
        super().__init__()
        self.conv0 = nn.Conv3d(dim, dim, 5, padding=2, groups=dim)
        self.conv_spatial = nn.Conv3d(dim, dim, 7, stride=1, padding=9, groups=dim, dilation=3)
        self.deform_conv = DeformConvPack(in_channels=dim, out_channels=dim, kernel_size=(3, 3, 3), stride=1, padding=1)
        self.conv1 = nn.Conv3d(dim, dim, 1)

    def forward(self, x):
        u = x
        c0, cs, c1 = self.conv0, self.conv_spatial, self.conv1
        attn = nn_ops.conv3d(x, c0.weight, c0.bias, c0.stride, c0.padding, c0.dilation, c0.groups)
        attn = nn_ops.conv3d(attn, cs.weight, cs.bias, cs.stride, cs.padding, cs.dilation, cs.groups)
        attn = attn.contiguous()
        attn = self.deform_conv(attn)
        attn = nn_ops.conv3d(attn, c1.weight, c1.bias)
        return u * attn


class _LKA3dAttentionFn(Function):
    @staticmethod
    def forward(ctx, x, *params):
        y, saved = ops.lka3d_attention_forward(x, params)
        ctx.save_for_backward(x, saved, *params)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, saved, *params = ctx.saved_tensors
        gx, grads = ops.lka3d_attention_backward(x, params, gy, saved)
        return (gx, *grads)


class _LKA3dTokensFn(Function):
    """Whole block on the token tensor [B, N, C] (channels-last end to end, no permutes)."""

    @staticmethod
    def forward(ctx, x, dims, *params):
        y, saved = ops.lka3d_attention_tokens_forward(x, params, dims)
        ctx.dims = dims
        ctx.save_for_backward(x, saved, *params)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, saved, *params = ctx.saved_tensors
        gx, grads = ops.lka3d_attention_tokens_backward(x, params, gy, saved, ctx.dims)
        return (gx, None, *grads)


class LKA_Attention3d_deform(nn.Module):
    """transformerblock.py:655-673."""

    def __init__(self, d_model):
        super().__init__()
        self.proj_1 = nn.Conv3d(d_model, d_model, 1)
        self.activation = nn.GELU()
        self.spatial_gating_unit = LKA3d_deform(d_model)
        self.proj_2 = nn.Conv3d(d_model, d_model, 1)

    def block_params(self):
        """The 14 tensors in ``dlka_lka3d_params`` order (include/dlka.h)."""
        s = self.spatial_gating_unit
        d = s.deform_conv
        return (self.proj_1.weight, self.proj_1.bias, s.conv0.weight, s.conv0.bias, s.conv_spatial.weight, s.conv_spatial.bias,
                d.conv_offset.weight, d.conv_offset.bias, d.weight, d.bias, s.conv1.weight, s.conv1.bias,
                self.proj_2.weight, self.proj_2.bias)

    def forward_volume(self, x):
        """x: [B, C, H, W, D] volume -> same shape (the block without the token<->volume permutes)."""
        return _LKA3dAttentionFn.apply(x, *self.block_params())

    def forward(self, x, B, C, H, W, D):
        # Fast path: the token tensor IS the channels-last volume; run the block on it directly.
        if ops.lka3d_tokens_supported(x, B, C, H, W, D):
            return _LKA3dTokensFn.apply(x, (H, W, D), *self.block_params())
        # General path (any C / dtype): the reference's own data movement around the NCDHW block.
        x = x.permute(0, 2, 1).reshape(B, C, H, W, D)  # B N C --> B C N --> B C H W D   (:665)
        x = self.forward_volume(x)
        x = x.reshape(B, C, H * W * D).permute(0, 2, 1)  # (:672)
        return x
