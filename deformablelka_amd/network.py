"""``D_LKA_Former`` — the 3-D network whose 21 transformer blocks are the D-LKA hot path.

Mirrors 3D/d_lka_former/network_architecture/synapse/d_lka_former_synapse.py:8-167 and
synapse/model_components.py:13-150: same constructor arguments, attribute names and ``state_dict`` keys (a reference
checkpoint loads with ``strict=True``), same forward dataflow and deep-supervision outputs.

What runs where:
  * the 21 ``TransformerBlock_3D_single_deform_LKA`` (encoder stages 0-3, decoders 5/4/3) — this repo's HIP kernels, chained in the
    channels-last layout (``keep_channels_last``) so that no block boundary costs a layout copy;
  * the plumbing around them — stem / down-sampling convs + GroupNorm, transposed convs, ``encoder1`` / ``decoder2`` conv blocks with
    InstanceNorm, the 1x1x1 output heads — ``torch.nn`` parameter containers (local stand-ins for the MONAI 0.8.1 factories the
    reference builds them from: ``get_conv_layer`` -> ``Convolution(conv_only=True)`` = a Sequential whose child ``conv`` is the torch
    conv; ``get_norm_layer`` -> GroupNorm / InstanceNorm3d / BatchNorm3d; dynunet_block.py:226-270).  Norms run as stock torch ops; the
    convs are computed as GEMMs / by this repo's HIP conv kernels on the GPU in fp32, because MIOpen's fp32 3-D paths are pathologically
    slow on this stack (see ``Convolution``).  None of this is hot-path code; it exists so that the full net trains.
The reference hard-codes the token counts of the Synapse patch (model_components.py:14, d_lka_former_synapse.py:97-132); here they
follow from ``img_size`` and ``patch_size`` (defaults = the reference's), which also covers the pancreas variant (96^3, stem (2,2,2)).
"""
from __future__ import annotations

from typing import Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from .dynunet_block import UnetResBlock as _HipUnetResBlock  # noqa: F401  (re-exported for checkpoint tools)
from .transformerblock import TransformerBlock_3D_single_deform_LKA


# ---- stand-ins for the MONAI factories (dynunet_block.py:226-292) ------------------------------------------------------------------
def get_padding(kernel_size, stride):
    k, s = np.atleast_1d(kernel_size), np.atleast_1d(stride)
    p = (k - s + 1) / 2
    if np.min(p) < 0:
        raise AssertionError("padding value should not be negative, please change the kernel size and/or stride.")
    p = tuple(int(v) for v in p)
    return p if len(p) > 1 else p[0]


def get_output_padding(kernel_size, stride, padding):
    k, s, p = np.atleast_1d(kernel_size), np.atleast_1d(stride), np.atleast_1d(padding)
    o = 2 * p + s - k
    if np.min(o) < 0:
        raise AssertionError("out_padding value should not be negative, please change the kernel size and/or stride.")
    o = tuple(int(v) for v in o)
    return o if len(o) > 1 else o[0]


class _RowsMatmul(torch.autograd.Function):
    """``y = a @ w`` for a tall ``a`` [M, K] (M = voxels of a batch, K <= a few hundred) and a small ``w`` [K, N] — the GEMM behind the kernel == stride convolutions below.
    autograd's own weight gradient ``a^T @ gy`` is ONE GEMM with a 65 536-long contraction and a 32 x 512 (or 32 x 32) result: rocBLAS runs it at 225 - 240 us at the net's
    full-resolution layers (the stem and the last up-sampling), a fifth of the iteration's rocBLAS time.  Here the rows are cut into chunks of 1024, the chunk products are one
    batched GEMM and their sum one reduction: 47 us (``scripts/time_upconv_gemms.py``).  Same products; the row sum is formed in another order (fp32 rounding only)."""

    CHUNK = 1024

    @staticmethod
    def forward(ctx, a, w):
        ctx.save_for_backward(a, w)
        return torch.mm(a, w)

    @staticmethod
    def backward(ctx, gy):
        a, w = ctx.saved_tensors
        ga = gw = None
        if ctx.needs_input_grad[0]:
            ga = torch.mm(gy, w.t())
        if ctx.needs_input_grad[1]:
            M, c = a.shape[0], _RowsMatmul.CHUNK
            gy = gy.contiguous()
            if M >= 16 * c and M % c == 0 and a.is_contiguous():
                gw = torch.bmm(a.view(M // c, c, -1).transpose(1, 2), gy.view(M // c, c, -1)).sum(0)
            else:
                gw = torch.mm(a.t(), gy)
        return ga, gw


class Convolution(nn.Sequential):
    """``monai.networks.blocks.Convolution(..., conv_only=True)``: a Sequential with ONE child named ``conv`` (the parameters live there, so
    the ``state_dict`` keys are the reference's).

    On the GPU in fp32 the forward does not call MIOpen: measured on the MI355X (scripts/probe_miopen_conv3d.py, profiles/archive/r03f) its fp32 3-D
    paths cost 320 ms for ONE 3x3x3 16 -> 16 conv fwd+bwd at 2 x 64x128x128 (weight gradient) and 10 ms for the (2,4,4) transposed conv — a
    1.03 s training step of which the 21 D-LKA blocks are 24 ms.  The four shapes the net uses are instead computed as
      * kernel == stride, no padding (stem, down-sampling):       patchify + one GEMM,
      * transposed, kernel == stride (up-sampling):               one GEMM + depth-to-space,
      * 1x1x1 (output heads):                                     one GEMM over the channel axis,
      * 3x3x3 stride 1 (encoder1 / decoder2 conv blocks):         this repo's general HIP conv kernels (``nn_ops.conv3d``),
    all exact re-expressions of the same convolution (fp32; the GEMMs are rocBLAS through ``torch.matmul``, autograd included)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias=False, is_transposed=False):
        super().__init__()
        pad = get_padding(kernel_size, stride)
        if is_transposed:
            conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride, pad, get_output_padding(kernel_size, stride, pad), bias=bias)
        else:
            conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride, pad, bias=bias)
        self.add_module("conv", conv)
        self.gemm_path = True   # set False to force the stock torch / MIOpen layer

    def forward(self, x):
        c = self.conv
        if not (self.gemm_path and x.is_cuda and c.weight.dtype == torch.float32 and x.dtype in (torch.float32, torch.bfloat16)):
            return c(x)
        if x.dtype != torch.float32:   # (a bf16 tensor from an autocast region: the re-expressions below are fp32 — widen explicitly, do not fall back to
            x = x.float()              #  the MIOpen 3-D paths this class exists to avoid)
        if torch.is_autocast_enabled():
            with torch.autocast(x.device.type, enabled=False):
                return self._forward_fp32(x)
        return self._forward_fp32(x)

    def _forward_fp32(self, x):
        c = self.conv
        k, s, p = tuple(c.kernel_size), tuple(c.stride), tuple(c.padding)
        B, Cin = x.shape[:2]
        if isinstance(c, nn.ConvTranspose3d):
            if k == s and p == (0, 0, 0) and tuple(c.output_padding) == (0, 0, 0) and c.groups == 1:
                D, H, W = x.shape[2:]
                Cout = c.weight.shape[1]
                y = _RowsMatmul.apply(x.permute(0, 2, 3, 4, 1).reshape(-1, Cin), c.weight.reshape(Cin, -1))       # (B D H W, Cout kd kh kw)
                y = y.reshape(B, D, H, W, Cout, *k).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(B, Cout, D * k[0], H * k[1], W * k[2])
                return y if c.bias is None else y + c.bias.view(1, -1, 1, 1, 1)
            return c(x)
        if c.groups != 1 or tuple(c.dilation) != (1, 1, 1):
            return c(x)
        Cout = c.weight.shape[0]
        if k == (1, 1, 1) and s == (1, 1, 1) and p == (0, 0, 0):
            from . import nn_ops, ops
            if ops.pointwise_planar_supported(x, c.weight, need_weight_grad=c.weight.requires_grad):
                # the 16 -> 14 head at 2 x 64x128x128 as a GEMM lands on a 16 x 16 hipBLASLt macro-tile: 3.8 ms + 1.7 ms of gradients; streamed: 0.3 ms
                return nn_ops.pointwise_planar(x.contiguous(), c.weight, c.bias)
            y = torch.matmul(c.weight.reshape(Cout, Cin), x.reshape(B, Cin, -1)).reshape(B, Cout, *x.shape[2:])
            return y if c.bias is None else y + c.bias.view(1, -1, 1, 1, 1)
        if k == s and p == (0, 0, 0) and all(n % kk == 0 for n, kk in zip(x.shape[2:], k)):
            D, H, W = (n // kk for n, kk in zip(x.shape[2:], k))
            cols = x.reshape(B, Cin, D, k[0], H, k[1], W, k[2]).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B * D * H * W, -1)   # patches
            y = _RowsMatmul.apply(cols, c.weight.reshape(Cout, -1).t()).reshape(B, D, H, W, Cout).permute(0, 4, 1, 2, 3).contiguous()
            return y if c.bias is None else y + c.bias.view(1, -1, 1, 1, 1)
        if s == (1, 1, 1):
            from . import nn_ops
            return nn_ops.conv3d(x.contiguous(), c.weight, c.bias, s, p, (1, 1, 1), 1)
        return c(x)


def get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=3, stride=1, dropout=None, bias=False, conv_only=True,
                   is_transposed=False):
    if spatial_dims != 3 or not conv_only or dropout not in (None, 0.0):
        raise NotImplementedError("stand-in covers the calls the reference makes: 3-D, conv_only=True, no dropout")
    return Convolution(in_channels, out_channels, kernel_size, stride, bias=bias, is_transposed=is_transposed)


def get_norm_layer(name, spatial_dims=3, channels=1):
    kind, kw = (name, {}) if isinstance(name, str) else (name[0], dict(name[1]) if len(name) > 1 else {})
    kind = kind.lower()
    if kind == "group":
        return GroupNorm(num_channels=channels, **kw)
    if kind == "instance":
        return InstanceNorm3d(channels, **kw)
    if kind == "batch":
        return BatchNorm3d(channels, **kw)
    raise NotImplementedError(f"norm {name!r}")


class GroupNorm(nn.GroupNorm):
    """nn.GroupNorm (same parameters and state_dict keys: the norm behind the stem and the down-sampling convs, model_components.py:27-34).  The stem's
    has ONE group (``num_groups = in_channels``): torch computes its moments with one workgroup per (sample, group) row — two workgroups for the
    2 x 32 x 32 x 32 x 32 stem output, 0.68 ms per iteration.  Rows that long go through the planar statistics / normalisation kernels (each row spread over
    the chip, as ``InstanceNorm3d`` below), followed by the per-channel affine as torch ops; short rows (the later stages) stay on the stock layer."""

    LONG_ROW = 1 << 18   # elements per (sample, group) row from which the one-workgroup-per-row moments under-fill the chip

    def forward(self, x):
        B, C, G = x.shape[0], x.shape[1], self.num_groups
        row = (C // G) * (x[0, 0].numel() if x.dim() > 2 else 1)
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3 and not torch.is_autocast_enabled() and row >= self.LONG_ROW
                and B * G <= 65535 and (self.weight is None or self.weight.dtype == torch.float32)):
            return super().forward(x)
        return self.planar_forward(x)

    def planar_forward(self, x):
        """The long-row path (any device the library runs on: the CPU tests call it on the emulator build)."""
        from . import nn_ops
        B, C, G = x.shape[0], x.shape[1], self.num_groups
        y, _ = nn_ops.batch_norm_train(x.contiguous(), None, None, self.eps, planes=(B * G,))   # (y in x's shape, not a view of the op's output: see nn_ops)
        if self.weight is not None:
            shape = (1, C) + (1,) * (x.dim() - 2)
            y = torch.addcmul(self.bias.view(shape), y, self.weight.view(shape))
        return y


class InstanceNorm3d(nn.InstanceNorm3d):
    """nn.InstanceNorm3d (the net's default ``norm_name``; no affine parameters, no running statistics: dynunet_block.py get_norm_layer
    ("instance")).  torch evaluates it as a batch norm over B*C channels with ONE workgroup per channel — 32 workgroups for the 2 x 16 full-resolution
    planes of encoder1 / decoder2, 1.3 ms per layer and iteration; here every plane is spread over the chip (csrc/planar_ops.hip, the same
    kernels as ``BatchNorm3d`` with B*C single-plane channels).  Affine / tracked variants go through the stock layer."""

    def forward(self, x):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and not self.affine and not self.track_running_stats
                and not torch.is_autocast_enabled() and x.shape[0] * x.shape[1] <= 65535):   # (one grid row per (b, c) plane)
            return super().forward(x)
        from . import nn_ops
        B, C = x.shape[:2]
        y, _ = nn_ops.batch_norm_train(x.contiguous(), None, None, self.eps, planes=(B * C,))   # (y in x's shape, not a view of the op's output: see nn_ops)
        return y


class BatchNorm3d(nn.BatchNorm3d):
    """nn.BatchNorm3d (same parameters, buffers and state_dict keys).  In training mode on fp32 GPU tensors the batch statistics, the
    normalisation and their gradients run on this repo's planar kernels: torch's batch-norm kernels use ONE workgroup per channel — 16
    workgroups on 256 CUs for the 16-channel full-resolution tensors of encoder1 / decoder2: 1.3 ms per layer and iteration against 0.3 ms
    (csrc/planar_ops.hip).  Evaluation mode and every other input go through the stock layer."""

    def forward(self, x):
        if not (self.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and not torch.is_autocast_enabled()
                and (self.weight is None or self.weight.dtype == torch.float32) and x.shape[0] * x.shape[1] <= 65535):
            return super().forward(x)
        from . import nn_ops
        y, stats = nn_ops.batch_norm_train(x.contiguous(), self.weight, self.bias, self.eps)
        if self.track_running_stats and self.running_mean is not None:
            with torch.no_grad():
                self.num_batches_tracked += 1
                if self.momentum is not None:
                    torch._foreach_lerp_([self.running_mean, self.running_var], [stats[0], stats[2]], float(self.momentum))
                else:
                    m = 1.0 / float(self.num_batches_tracked)
                    self.running_mean.mul_(1.0 - m).add_(stats[0], alpha=m)
                    self.running_var.mul_(1.0 - m).add_(stats[2], alpha=m)
        return y


class UnetResBlock(nn.Module):
    """dynunet_block.py:12-80 in full generality (any norm, in != out channels -> the 1x1x1 ``conv3`` / ``norm3`` branch) on torch
    layers: ``encoder1`` (1 -> 16 channels at full resolution) and ``decoder2`` of the net.  The C -> C batch-norm instance inside every
    transformer block is ``deformablelka_amd.dynunet_block.UnetResBlock`` (HIP)."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name,
                 act_name=("leakyrelu", {"inplace": True, "negative_slope": 0.01}), dropout=None):
        super().__init__()
        self.conv1 = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=kernel_size, stride=stride, dropout=dropout)
        self.conv2 = get_conv_layer(spatial_dims, out_channels, out_channels, kernel_size=kernel_size, stride=1, dropout=dropout)
        self.lrelu = nn.LeakyReLU(**act_name[1])
        self.norm1 = get_norm_layer(norm_name, spatial_dims, out_channels)
        self.norm2 = get_norm_layer(norm_name, spatial_dims, out_channels)
        self.downsample = in_channels != out_channels or not np.all(np.atleast_1d(stride) == 1)
        if self.downsample:
            self.conv3 = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=1, stride=stride, dropout=dropout)
            self.norm3 = get_norm_layer(norm_name, spatial_dims, out_channels)

    def forward(self, inp):   # dynunet_block.py:66-80
        residual = inp
        out = self.lrelu(self.norm1(self.conv1(inp)))
        out = self.norm2(self.conv2(out))
        if self.downsample:
            residual = self.norm3(self.conv3(residual))
        out = out + residual
        return self.lrelu(out)


class UnetOutBlock(nn.Module):
    """dynunet_block.py:211-223."""

    def __init__(self, spatial_dims, in_channels, out_channels, dropout=None):
        super().__init__()
        self.conv = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=1, stride=1, dropout=dropout, bias=True)

    def forward(self, inp):
        return self.conv(inp)


def _chain(blocks: nn.Sequential, x):
    """Runs the blocks of one stage back to back in the channels-last layout and hands a contiguous NCDHW tensor to the torch layers
    that follow (the blocks in between exchange the channels_last_3d view: no layout copy at their boundaries)."""
    n = len(blocks)
    for i, blk in enumerate(blocks):
        if isinstance(blk, TransformerBlock_3D_single_deform_LKA):
            x = blk(x, keep_channels_last=i + 1 < n)   # per call: no module state is touched (a stand-alone call of the block keeps its contiguous output)
        else:
            x = blk(x)
    return x


class D_LKA_Former_Encoder(nn.Module):
    """model_components.py:13-66."""

    def __init__(self, input_size=(32 * 32 * 32, 16 * 16 * 16, 8 * 8 * 8, 4 * 4 * 4), dims=(32, 64, 128, 256), proj_size=(64, 64, 64, 32),
                 depths=(3, 3, 3, 3), num_heads=4, spatial_dims=3, in_channels=1, dropout=0.0, transformer_dropout_rate=0.15,
                 trans_block=TransformerBlock_3D_single_deform_LKA, patch_size=(2, 4, 4), **kwargs):
        super().__init__()
        self.downsample_layers = nn.ModuleList()
        self.downsample_layers.append(nn.Sequential(
            get_conv_layer(spatial_dims, in_channels, dims[0], kernel_size=patch_size, stride=patch_size, dropout=dropout),
            get_norm_layer(("group", {"num_groups": in_channels}), channels=dims[0])))
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(
                get_conv_layer(spatial_dims, dims[i], dims[i + 1], kernel_size=(2, 2, 2), stride=(2, 2, 2), dropout=dropout),
                get_norm_layer(("group", {"num_groups": dims[i]}), channels=dims[i + 1])))
        self.stages = nn.ModuleList()
        for i in range(4):
            self.stages.append(nn.Sequential(*[
                trans_block(input_size=input_size[i], hidden_size=dims[i], proj_size=proj_size[i], num_heads=num_heads,
                            dropout_rate=transformer_dropout_rate, pos_embed=True) for _ in range(depths[i])]))
        self.hidden_states = []
        self.apply(self._init_weights)

    def _init_weights(self, m):   # model_components.py:43-50 (Conv2d / Linear only: none exist here; LayerNorm to (1, 0))
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward_features(self, x):
        hidden_states = []
        x = self.downsample_layers[0](x)
        x = _chain(self.stages[0], x)
        hidden_states.append(x)
        for i in range(1, 4):
            x = self.downsample_layers[i](x)
            x = _chain(self.stages[i], x)
            if i == 3:   # einops.rearrange(x, "b c h w d -> b (h w d) c"), model_components.py:62
                b, c = x.shape[:2]
                x = x.reshape(b, c, -1).permute(0, 2, 1)
            hidden_states.append(x)
        return x, hidden_states

    def forward(self, x):
        return self.forward_features(x)


class D_LKA_FormerUpBlock(nn.Module):
    """model_components.py:69-150."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, upsample_kernel_size, norm_name, proj_size=64, num_heads=4,
                 out_size=0, depth=3, conv_decoder=False, trans_block=TransformerBlock_3D_single_deform_LKA, use_skip=True):
        super().__init__()
        self.transp_conv = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=upsample_kernel_size,
                                          stride=upsample_kernel_size, is_transposed=True)
        self.use_skip = use_skip
        self.decoder_block = nn.ModuleList()
        if conv_decoder:
            self.decoder_block.append(UnetResBlock(spatial_dims, out_channels, out_channels, kernel_size=kernel_size, stride=1, norm_name=norm_name))
        else:
            self.decoder_block.append(nn.Sequential(*[
                trans_block(input_size=out_size, hidden_size=out_channels, proj_size=proj_size, num_heads=num_heads, dropout_rate=0.15,
                            pos_embed=True) for _ in range(depth)]))

    def forward(self, inp, skip):
        out = self.transp_conv(inp)
        if self.use_skip:
            out = out + skip
        blk = self.decoder_block[0]
        return _chain(blk, out) if isinstance(blk, nn.Sequential) else blk(out)


class D_LKA_Former(nn.Module):
    """d_lka_former_synapse.py:8-167 (``SegmentationNetwork`` there is nnU-Net's inference base class; the sliding-window predictor
    lives in ``deformablelka_amd.inference``)."""

    def __init__(self, in_channels: int, out_channels: int, img_size: Sequence[int] = (64, 128, 128), feature_size: int = 16,
                 hidden_size: int = 256, num_heads: int = 4, pos_embed: str = "perceptron", norm_name: Union[Tuple, str] = "instance",
                 dropout_rate: float = 0.0, depths=None, dims=None, conv_op=nn.Conv3d, do_ds=True,
                 trans_block=TransformerBlock_3D_single_deform_LKA, skip_connections=(True, True, True, True),
                 patch_size: Sequence[int] = (2, 4, 4)) -> None:
        super().__init__()
        depths = [3, 3, 3, 3] if depths is None else list(depths)
        dims = [32, 64, 128, 256] if dims is None else list(dims)
        self.do_ds, self.conv_op, self.num_classes = do_ds, conv_op, out_channels
        if not (0 <= dropout_rate <= 1):
            raise AssertionError("dropout_rate should be between 0 and 1.")
        if pos_embed not in ("conv", "perceptron"):
            raise KeyError(f"Position embedding layer of type {pos_embed} is not supported.")
        self.patch_size = tuple(patch_size)
        self.feat_size = tuple(img_size[i] // self.patch_size[i] // 8 for i in range(3))
        self.hidden_size = hidden_size
        tok = [int(np.prod([img_size[a] // self.patch_size[a] // (2 ** s) for a in range(3)])) for s in range(4)]   # 32^3, 16^3, 8^3, 4^3 for Synapse
        self.d_lka_former_encoder = D_LKA_Former_Encoder(input_size=tok, dims=dims, depths=depths, num_heads=num_heads, in_channels=in_channels,
                                                         trans_block=trans_block, patch_size=self.patch_size)
        self.encoder1 = UnetResBlock(3, in_channels, feature_size, kernel_size=3, stride=1, norm_name=norm_name)
        up = dict(spatial_dims=3, kernel_size=3, norm_name=norm_name, trans_block=trans_block)
        self.decoder5 = D_LKA_FormerUpBlock(in_channels=feature_size * 16, out_channels=feature_size * 8, upsample_kernel_size=2, out_size=tok[2],
                                            use_skip=skip_connections[0], **up)
        self.decoder4 = D_LKA_FormerUpBlock(in_channels=feature_size * 8, out_channels=feature_size * 4, upsample_kernel_size=2, out_size=tok[1],
                                            use_skip=skip_connections[1], **up)
        self.decoder3 = D_LKA_FormerUpBlock(in_channels=feature_size * 4, out_channels=feature_size * 2, upsample_kernel_size=2, out_size=tok[0],
                                            use_skip=skip_connections[2], **up)
        self.decoder2 = D_LKA_FormerUpBlock(in_channels=feature_size * 2, out_channels=feature_size, upsample_kernel_size=self.patch_size,
                                            out_size=int(np.prod(img_size)), conv_decoder=True, use_skip=skip_connections[3], **up)
        self.out1 = UnetOutBlock(3, feature_size, out_channels)
        if self.do_ds:
            self.out2 = UnetOutBlock(3, feature_size * 2, out_channels)
            self.out3 = UnetOutBlock(3, feature_size * 4, out_channels)

    def proj_feat(self, x, hidden_size, feat_size):
        x = x.reshape(x.size(0), feat_size[0], feat_size[1], feat_size[2], hidden_size)
        return x.permute(0, 4, 1, 2, 3).contiguous()

    def forward(self, x_in):
        x_output, hidden_states = self.d_lka_former_encoder(x_in)
        conv_block = self.encoder1(x_in)
        enc1, enc2, enc3, enc4 = hidden_states
        dec4 = self.proj_feat(enc4, self.hidden_size, self.feat_size)
        dec3 = self.decoder5(dec4, enc3)
        dec2 = self.decoder4(dec3, enc2)
        dec1 = self.decoder3(dec2, enc1)
        out = self.decoder2(dec1, conv_block)
        if self.do_ds:
            return [self.out1(out), self.out2(dec1), self.out3(dec2)]
        return self.out1(out)

    def dlka_blocks(self):
        """The 21 D-LKA transformer blocks, in forward order."""
        return [m for m in self.modules() if isinstance(m, TransformerBlock_3D_single_deform_LKA)]
