"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

CPU restatement of the reference's D-LKA *blocks*: the plain convs / GELU run through ATen's CPU kernels (that IS
the reference's CPU path for nn.Conv3d / nn.Conv2d / nn.GELU), the deformable conv through the C oracle.
Parameters are addressed by the reference's ``state_dict`` keys so that the same dict drives the reference
modules, this oracle and the HIP modules.
"""
import torch
import torch.nn.functional as F

import oracle


def lka3d_attention_volume(x, P):
    """LKA_Attention3d_deform on an NCDHW volume — 3D/d_lka_former/network_architecture/synapse/transformerblock.py:664-673
    (minus the token permutes), LKA3d_deform.forward :644-652, DeformConvPack.forward synapse/deform_conv.py:93-105."""
    C = x.shape[1]
    shortcut = x.clone()                                                         # :666
    a = F.gelu(F.conv3d(x, P["proj_1.weight"], P["proj_1.bias"]))                # :667-668
    u = a.clone()                                                                # :645
    s = "spatial_gating_unit."
    attn = F.conv3d(a, P[s + "conv0.weight"], P[s + "conv0.bias"], padding=2, groups=C)                             # :646
    attn = F.conv3d(attn, P[s + "conv_spatial.weight"], P[s + "conv_spatial.bias"], padding=9, dilation=3, groups=C)  # :647
    attn = attn.contiguous()                                                     # :648
    off = F.conv3d(attn, P[s + "deform_conv.conv_offset.weight"], P[s + "deform_conv.conv_offset.bias"], stride=1, padding=1)
    attn = oracle.DeformConv3dFunction.apply(attn, off, P[s + "deform_conv.weight"], P[s + "deform_conv.bias"],
                                             1, 1, 1, 1, 1, 64)                  # deform_conv.py:95-105
    attn = F.conv3d(attn, P[s + "conv1.weight"], P[s + "conv1.bias"])            # :650
    y = F.conv3d(u * attn, P["proj_2.weight"], P["proj_2.bias"])                 # :652, :670
    return y + shortcut                                                          # :671


def lka3d_attention_tokens(x, P, B, C, H, W, D):
    """The full forward(x, B, C, H, W, D) on (B, N, C) tokens, :664-673."""
    v = x.permute(0, 2, 1).reshape(B, C, H, W, D)
    v = lka3d_attention_volume(v, P)
    return v.reshape(B, C, H * W * D).permute(0, 2, 1)


def deform_conv_2d_pack(x, P, prefix, k, pad, dil, groups):
    """2-D ``DeformConv.forward`` — 2D/deformable_LKA/deformable_LKA.py:27-30."""
    off = F.conv2d(x, P[prefix + "offset_net.weight"], P[prefix + "offset_net.bias"], padding=pad, dilation=dil)
    return oracle.DeformConv2dFunction.apply(x, off, P[prefix + "deform_conv.weight"], None, 1, pad, dil)


def lka2d_attention(x, P):
    """deformable_LKA_Attention.forward — deformable_LKA.py:133-140 with deformable_LKA.forward :98-104."""
    C = x.shape[1]
    shortcut = x.clone()
    a = F.gelu(F.conv2d(x, P["proj_1.weight"], P["proj_1.bias"]))
    u = a.clone()
    s = "spatial_gating_unit."
    attn = deform_conv_2d_pack(a, P, s + "conv0.", 5, 2, 1, C)                   # :93
    attn = deform_conv_2d_pack(attn, P, s + "conv_spatial.", 7, 9, 3, C)         # :94
    attn = F.conv2d(attn, P[s + "conv1.weight"], P[s + "conv1.bias"])
    y = F.conv2d(u * attn, P["proj_2.weight"], P["proj_2.bias"])
    return y + shortcut


def randomize_offsets_(module, std=0.05, seed=123):
    """Fresh modules have zero offset predictors (deform_conv.py:86-88) which would make every sample
    integer-aligned; give the offset convs small random weights so the deformable path is really exercised."""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if "conv_offset" in name or "offset_net" in name:
                p.copy_(torch.randn(p.shape, generator=gen) * std)
