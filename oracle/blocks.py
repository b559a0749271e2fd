"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

CPU restatement of the reference's D-LKA *blocks*: the plain convs / GELU run through ATen's CPU kernels (that IS
the reference's CPU path for nn.Conv3d / nn.Conv2d / nn.GELU), the deformable conv through the C oracle.
Parameters are addressed by the reference's ``state_dict`` keys so that the same dict drives the reference
modules, this oracle and the HIP modules.
"""
import torch
import torch.nn.functional as F

import oracle


def bf16_storage(t):
    """Straight-through bf16 rounding: the value a tensor has after being STORED as bf16 (what the DLKA_BF16 path does to every activation
    it writes), with the identity as its derivative."""
    return t + (t.bfloat16().float() - t).detach()


def lka3d_attention_volume(x, P, store=None, chain_store=None, offsets_override=None, offsets_out=None, x_chain=None):
    """LKA_Attention3d_deform on an NCDHW volume — 3D/d_lka_former/network_architecture/synapse/transformerblock.py:664-673
    (minus the token permutes), LKA3d_deform.forward :644-652, DeformConvPack.forward synapse/deform_conv.py:93-105.
    store: None = the reference's fp32 block; ``bf16_storage`` = the model of the DLKA_BF16 path: the same arithmetic with every activation
    that path writes to HBM AS bf16 rounded where it is written.  The chain that decides the sampling cells — a = GELU(proj_1 x) -> conv0 ->
    conv_spatial -> conv_offset — stays fp32 there (deformablelka_amd/csrc/dlka_capi_cl.hip, TokGeoms), so it is not rounded here either; the
    gate and the deformable conv's SAMPLES read the bf16 copies of a and t.
    chain_store: rounding applied to the chain tensors a / t1 / t as the offset-determining convs READ them (None = fp32, what the product does;
    ``bf16_storage`` = the round-2 design with every activation in bf16 — kept so that tests/test_oracle_bf16_model.py can show why it was dropped).
    x_chain: the UNROUNDED twin of a bf16-rounded x (the wrapper block's mixed mode hands the attention LayerNorm's fp32 output beside its bf16 copy,
    dlka_tblock3d_forward_v): the offset-determining chain — and with it the tensor t whose bf16 copy the deformable conv samples — then starts from it;
    the gate, the shortcut and proj_1's saved activations read x."""
    st = store if store is not None else (lambda t: t)
    cs = chain_store if chain_store is not None else (lambda t: t)
    C = x.shape[1]
    shortcut = x.clone()                                                         # :666
    a = F.gelu(F.conv3d(x, P["proj_1.weight"], P["proj_1.bias"]))                # :667-668
    u = st(a)                                                                    # :645 (the gate's copy)
    if x_chain is not None:
        a = F.gelu(F.conv3d(x_chain, P["proj_1.weight"], P["proj_1.bias"]))
    s = "spatial_gating_unit."
    # the depthwise pair by its weight shapes: Synapse 5^3 p2 + 7^3 dil 3 p9 (:637-638); ACDC (acdc/transformerblock.py:213-237) (5,7,7) dil 3
    # p (6,9,9) / (3,5,5) dil (1,3,3) p (1,6,6) / 3^3 p1 after 5^3 p2 / 3^3 p1 — "same" padding with dilation 3 on every axis whose kernel > 3
    k0, k1 = tuple(P[s + "conv0.weight"].shape[2:]), tuple(P[s + "conv_spatial.weight"].shape[2:])
    d1 = {(7, 7, 7): (3, 3, 3), (5, 7, 7): (3, 3, 3), (3, 5, 5): (1, 3, 3), (3, 3, 3): (1, 1, 1)}[k1]
    p0 = tuple(k // 2 for k in k0)
    p1 = tuple(d * (k - 1) // 2 for k, d in zip(k1, d1))
    attn = cs(F.conv3d(cs(a), P[s + "conv0.weight"], P[s + "conv0.bias"], padding=p0, groups=C))                        # :646
    attn = cs(F.conv3d(attn, P[s + "conv_spatial.weight"], P[s + "conv_spatial.bias"], padding=p1, dilation=d1, groups=C))  # :647
    attn = attn.contiguous()                                                     # :648
    off = F.conv3d(attn, P[s + "deform_conv.conv_offset.weight"], P[s + "deform_conv.conv_offset.bias"], stride=1, padding=1)
    if offsets_override is not None:   # VALUES from another implementation's forward pass, gradient path unchanged (straight through): both sides
        off = off + (offsets_override - off).detach()   # then sample the same cells, which removes grad_offset's discontinuity from a comparison
    if offsets_out is not None:        # the offset VALUES this very run samples with (a second evaluation of the chain need not be bit-identical:
        offsets_out.append(off.detach().clone())   # ATen's CPU convs split their sums by thread count)
    attn = st(oracle.DeformConv3dFunction.apply(st(attn), off, P[s + "deform_conv.weight"], P[s + "deform_conv.bias"],
                                                1, 1, 1, 1, 1, 64))              # deform_conv.py:95-105
    attn = F.conv3d(attn, P[s + "conv1.weight"], P[s + "conv1.bias"])            # :650  (the gate consumes conv1's fp32 value in the fused epilogue)
    y = F.conv3d(st(u * attn), P["proj_2.weight"], P["proj_2.bias"])             # :652, :670
    return st(y + shortcut)                                                      # :671


def lka3d_attention_tokens(x, P, B, C, H, W, D, store=None, chain_store=None, offsets_override=None, offsets_out=None, x_chain=None):
    """The full forward(x, B, C, H, W, D) on (B, N, C) tokens, :664-673."""
    v = x.permute(0, 2, 1).reshape(B, C, H, W, D)
    if x_chain is not None:
        x_chain = x_chain.permute(0, 2, 1).reshape(B, C, H, W, D)
    v = lka3d_attention_volume(v, P, store, chain_store, offsets_override, offsets_out, x_chain)
    return v.reshape(B, C, H * W * D).permute(0, 2, 1)


def leaky_relu_signed(z, sign=None, known=None, pre_out=None, slope=0.01):
    """LeakyReLU whose activation PATTERN can be taken from another implementation's forward pass: sign (bool, z's shape) = that implementation's
    `z > 0`, used where `known` (bool; None = everywhere), the own `z > 0` elsewhere.  Two correct fp32 implementations disagree about the pattern only
    for pre-activations within rounding of 0, and there the slope jumps by a factor 100 — the same role `offsets_override` plays for floor() in the
    deformable conv.  pre_out: list receiving (z.detach(), pattern used) for the tests' count of such elements."""
    if sign is None:
        if pre_out is not None:
            pre_out.append((z.detach(), z.detach() > 0))
        return F.leaky_relu(z, slope)
    pat = sign if known is None else torch.where(known, sign, z.detach() > 0)
    pat = pat.contiguous()   # (torch 2.10 CPU: the backward of torch.where with a NON-contiguous condition — e.g. a permuted token view at B = 1 — is wrong)
    if pre_out is not None:
        pre_out.append((z.detach(), pat))
    return torch.where(pat, z, slope * z)


def unet_res_block(x, P, prefix, training, stats_out=None, act_signs=None, pre_out=None):
    """UnetResBlock(3, C, C, kernel_size=3, stride=1, norm_name="batch").forward — dynunet_block.py:66-80 with MONAI 0.8's
    factories resolved (Convolution conv_only -> Conv3d bias=False padding 1; "batch" -> BatchNorm3d; LeakyReLU 0.01).
    Running statistics are NOT updated in place (the caller's dict stays intact); in training mode batch statistics are used."""
    def bn(v, n):
        rm, rv = P[prefix + n + ".running_mean"], P[prefix + n + ".running_var"]
        if training:
            return F.batch_norm(v, None, None, P[prefix + n + ".weight"], P[prefix + n + ".bias"], True, 0.1, 1e-5)
        return F.batch_norm(v, rm, rv, P[prefix + n + ".weight"], P[prefix + n + ".bias"], False, 0.1, 1e-5)
    s1, s2, k2 = act_signs if act_signs is not None else (None, None, None)      # (another implementation's activation patterns: leaky_relu_signed)
    out = F.conv3d(x, P[prefix + "conv1.conv.weight"], None, padding=1)          # :68
    out = leaky_relu_signed(bn(out, "norm1"), s1, None, pre_out)                 # :69-70
    out = F.conv3d(out, P[prefix + "conv2.conv.weight"], None, padding=1)        # :71
    out = bn(out, "norm2")                                                       # :72
    return leaky_relu_signed(out + x, s2, k2, pre_out)                           # :77-79


def transformer_block_3d(x, P, training=False, drop_mask=None, offsets_override=None, offsets_out=None, lka_store=None, lka_chain_fp32=True,
                         act_signs=None, pre_out=None):
    """TransformerBlock_3D_single_deform_LKA.forward — transformerblock.py:617-630.  drop_mask: the (B, C) multipliers of
    conv8[0] = Dropout3d(0.1) (None = eval / no dropout).  lka_store: ``bf16_storage`` = the model of the wrapper block's MIXED mode (the D-LKA
    attention on bf16 activations: its input, every tensor it stores and its output rounded where they are written; the wrapper itself fp32)."""
    B, C, H, W, D = x.shape
    t = x.reshape(B, C, H * W * D).permute(0, 2, 1)                              # :620
    if "pos_embed" in P and P["pos_embed"] is not None:
        t = t + P["pos_embed"]                                                   # :622-623
    n = F.layer_norm(t, (C,), P["norm.weight"], P["norm.bias"], 1e-5)
    n32 = None
    if lka_store is not None:   # the attention's bf16 input, and (round 5) its unrounded twin for the offset-determining chain
        n32, n = (n if lka_chain_fp32 else None), lka_store(n)
    lka = {k[len("epa_block."):]: v for k, v in P.items() if k.startswith("epa_block.")}
    attn = t + P["gamma"] * lka3d_attention_tokens(n, lka, B, C, H, W, D, store=lka_store, offsets_override=offsets_override, offsets_out=offsets_out,
                                                   x_chain=n32)        # :624
    skip = attn.reshape(B, H, W, D, C).permute(0, 4, 1, 2, 3)                    # :626
    a = unet_res_block(skip, P, "conv51.", training, act_signs=act_signs, pre_out=pre_out)   # :627  (act_signs: NCDHW bool (s1, s2, s2_known), leaky_relu_signed)
    if drop_mask is not None:
        a = a * drop_mask.view(B, C, 1, 1, 1)
    return skip + F.conv3d(a, P["conv8.1.weight"], P["conv8.1.bias"])            # :628


def deform_conv_2d_pack(x, P, prefix, k, pad, dil, groups, sample_store=None, offset_override=None, offsets_out=None):
    """2-D ``DeformConv.forward`` — 2D/deformable_LKA/deformable_LKA.py:27-30.  sample_store: rounding applied to the tensor the deformable conv
    SAMPLES (the offset net always reads x as it is).  offset_override / offsets_out: as in ``lka3d_attention_volume``."""
    off = F.conv2d(x, P[prefix + "offset_net.weight"], P[prefix + "offset_net.bias"], padding=pad, dilation=dil)
    if offset_override is not None:   # another implementation's offset VALUES, gradient path unchanged (straight through)
        off = off + (offset_override - off).detach()
    if offsets_out is not None:
        offsets_out.append(off.detach().clone())
    xs = x if sample_store is None else sample_store(x)
    return oracle.DeformConv2dFunction.apply(xs, off, P[prefix + "deform_conv.weight"], None, 1, pad, dil)


def lka2d_attention(x, P, store=None, offsets_override=None, offsets_out=None):
    """deformable_LKA_Attention.forward — deformable_LKA.py:133-140 with deformable_LKA.forward :98-104.
    store: None = the reference's fp32 block; ``bf16_storage`` = the model of the DLKA_BF16 2-D path: every activation that path writes to HBM as
    bf16 is rounded where it is written; the chain that decides the sampling cells (a -> offset net 5 -> t1 = DDW5(a) -> offset net 7) stays fp32
    there (dlka_capi_cl.hip, Lka2dCl) and is not rounded here either.
    offsets_override: (conv0's offsets, conv_spatial's offsets) VALUES from another implementation's forward pass (straight through, see
    ``lka3d_attention_volume``); offsets_out: list that receives the two offset tensors this run samples with."""
    st = store if store is not None else (lambda t: t)
    ov = offsets_override if offsets_override is not None else (None, None)
    C = x.shape[1]
    shortcut = x.clone()
    a = F.gelu(F.conv2d(x, P["proj_1.weight"], P["proj_1.bias"]))
    u = st(a)
    s = "spatial_gating_unit."
    attn = deform_conv_2d_pack(a, P, s + "conv0.", 5, 2, 1, C, offset_override=ov[0], offsets_out=offsets_out)   # :93   fp32 in, fp32 out (t1_32)
    attn = st(deform_conv_2d_pack(attn, P, s + "conv_spatial.", 7, 9, 3, C, offset_override=ov[1], offsets_out=offsets_out))   # :94   samples the fp32 t1, t2 stored as bf16
    attn = F.conv2d(attn, P[s + "conv1.weight"], P[s + "conv1.bias"])
    y = F.conv2d(st(u * attn), P["proj_2.weight"], P["proj_2.bias"])
    return st(y + shortcut)


def randomize_offsets_(module, std=0.05, seed=123):
    """Fresh modules have zero offset predictors (deform_conv.py:86-88) which would make every sample
    integer-aligned; give the offset convs small random weights so the deformable path is really exercised."""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if "conv_offset" in name or "offset_net" in name:
                p.copy_(torch.randn(p.shape, generator=gen) * std)
