"""2-D decoder of the D-LKA Net: ``deformableLKABlock`` and ``MyDecoderLayer`` of
2D/networks/MaxViT_deform_LKA.py:20-52,142-189,488-620 with the same constructor arguments, attribute names and ``state_dict`` keys.

The deformable large-kernel attention inside (``deformable_LKA_Attention``) and the depthwise 3x3 conv of the ``Mlp`` run on this repo's
HIP kernels; LayerNorm, the 1x1 ``fc1`` / ``fc2`` convs, the linear layers and the patch-expansion reshuffles are stock torch (they are
the MaxViT-side plumbing the survey leaves to PyTorch).  ``timm``'s ``DropPath`` is only instantiated for ``drop_path > 0``, which the
reference never passes (MaxViT_deform_LKA.py:583-585); the stand-in below is stochastic depth as published."""
import torch
import torch.nn as nn

from . import nn_ops
from .deformable_LKA import deformable_LKA_Attention


class DropPath(nn.Module):
    """timm.models.layers.DropPath (stochastic depth per sample)."""

    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask / keep


class DWConvLKA(nn.Module):
    """MaxViT_deform_LKA.py:20-27 — depthwise 3x3, through the HIP conv kernels."""

    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, x):
        c = self.dwconv
        return nn_ops.conv2d(x, c.weight, c.bias, c.stride, c.padding, c.dilation, c.groups)


class Mlp(nn.Module):
    """MaxViT_deform_LKA.py:29-52."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0., linear=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Conv2d(in_features, hidden_features, 1)
        self.dwconv = DWConvLKA(hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Conv2d(hidden_features, out_features, 1)
        self.drop = nn.Dropout(drop)
        self.linear = linear
        if self.linear:
            self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.fc1(x)
        if self.linear:
            x = self.relu(x)
        x = self.dwconv(x)
        x = self.act(x)
        x = self.drop(x)
        x = self.fc2(x)
        x = self.drop(x)
        return x


class deformableLKABlock(nn.Module):
    """MaxViT_deform_LKA.py:142-189: tokens (B, N, C) -> LayerNorm -> deformable LKA attention (layer-scaled residual) -> LayerNorm -> Mlp
    (layer-scaled residual) -> tokens."""

    def __init__(self, dim, mlp_ratio=4., drop=0., drop_path=0., act_layer=nn.GELU, linear=False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = deformable_LKA_Attention(dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop, linear=linear)
        layer_scale_init_value = 1e-2
        self.layer_scale_1 = nn.Parameter(layer_scale_init_value * torch.ones(dim), requires_grad=True)
        self.layer_scale_2 = nn.Parameter(layer_scale_init_value * torch.ones(dim), requires_grad=True)

    def forward(self, x, H, W):
        B, N, C = x.shape
        x = x.permute(0, 2, 1).reshape(B, C, H, W)
        y = self.norm1(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        y = self.attn(y.contiguous())
        x = x + self.drop_path(self.layer_scale_1.unsqueeze(-1).unsqueeze(-1) * y)
        y = self.norm2(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        y = self.mlp(y.contiguous())
        x = x + self.drop_path(self.layer_scale_2.unsqueeze(-1).unsqueeze(-1) * y)
        return x.reshape(B, C, N).permute(0, 2, 1)


def _pixel_shuffle_tokens(x, H, W, p, c):
    """einops ``"b h w (p1 p2 c) -> b (h p1) (w p2) c"`` (MaxViT_deform_LKA.py:509,537)."""
    B = x.shape[0]
    x = x.reshape(B, H, W, p, p, c).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, H * p * W * p, c)


class PatchExpand(nn.Module):
    """MaxViT_deform_LKA.py:488-513."""

    def __init__(self, input_resolution, dim, dim_scale=2, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution = input_resolution
        self.dim = dim
        self.expand = nn.Linear(dim, 2 * dim, bias=False) if dim_scale == 2 else nn.Identity()
        self.norm = norm_layer(dim // dim_scale)

    def forward(self, x):
        H, W = self.input_resolution
        x = self.expand(x)
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        return self.norm(_pixel_shuffle_tokens(x, H, W, 2, C // 4))


class FinalPatchExpand_X4(nn.Module):
    """MaxViT_deform_LKA.py:516-542."""

    def __init__(self, input_resolution, dim, dim_scale=4, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution = input_resolution
        self.dim = dim
        self.dim_scale = dim_scale
        self.expand = nn.Linear(dim, 16 * dim, bias=False)
        self.output_dim = dim
        self.norm = norm_layer(self.output_dim)

    def forward(self, x):
        H, W = self.input_resolution
        x = self.expand(x)
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        return self.norm(_pixel_shuffle_tokens(x, H, W, self.dim_scale, C // (self.dim_scale ** 2)))


class MyDecoderLayer(nn.Module):
    """MaxViT_deform_LKA.py:545-620: skip fusion by addition, two ``deformableLKABlock``s, patch expansion (and the 1x1 class head on
    the last layer).  Both D-LKA blocks are always constructed; the layer called without a skip (``decoder_3`` of the net, :691) never
    runs them, so their parameters get no gradient — a data-parallel wrapper must tolerate that (``training.wrap_data_parallel``)."""

    def __init__(self, input_size, in_out_chan, head_count, token_mlp_mode, n_class=9, norm_layer=nn.LayerNorm, is_last=False):
        super().__init__()
        dims, out_dim, key_dim, value_dim, x1_dim = in_out_chan
        self.x1_linear = nn.Linear(x1_dim, out_dim)
        if not is_last:
            self.layer_up = PatchExpand(input_resolution=input_size, dim=out_dim, dim_scale=2, norm_layer=norm_layer)
            self.last_layer = None
        else:
            self.layer_up = FinalPatchExpand_X4(input_resolution=input_size, dim=out_dim, dim_scale=4, norm_layer=norm_layer)
            self.last_layer = nn.Conv2d(out_dim, n_class, 1)
        self.layer_lka_1 = deformableLKABlock(dim=out_dim)
        self.layer_lka_2 = deformableLKABlock(dim=out_dim)
        for m in self.modules():   # init_weights, :587-599
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x1, x2=None):
        if x2 is None:
            return self.layer_up(x1)
        b, h, w, c = x2.shape
        x2 = x2.reshape(b, -1, c)
        x = self.x1_linear(x1) + x2
        x = self.layer_lka_1(x, h, w)
        x = self.layer_lka_2(x, h, w)
        if self.last_layer:
            return self.last_layer(self.layer_up(x).reshape(b, 4 * h, 4 * w, -1).permute(0, 3, 1, 2))
        return self.layer_up(x)
