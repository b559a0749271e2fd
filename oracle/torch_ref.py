"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

An *independent* second restatement of the reference's sampling rule, written with torch index
arithmetic + ``gather`` so that autograd supplies the backward.  It exists only to pin the C oracle
(``dlka_oracle.c``): two restatements written in different styles that agree to fp64 round-off,
plus ``F.grid_sample`` as a third opinion on the forward.

Rule restated (all citations relative to /root/reference):
  * coordinate = int base + float offset, guard ``q > -1 && q < size`` per axis
    (3D/dcn/src/cuda/deform_im2col_cuda.cuh:244-247)
  * floor, 8 corners, each corner zero unless low>=0 / high<=size-1 (cuh:30-66)
  * offset channel order: tap-major then (d,h,w)  (cuh:237-239)
  * grouped contraction + bias always added (3D/dcn/src/cuda/deform_conv_cuda.cu:111-119)
  * 2-D: torchvision 0.12 ``deform_conv2d`` — offset order (dy,dx) per tap, same corner rule.
"""
from __future__ import annotations

import itertools

import torch
import torch.nn.functional as F


def _t3(v):
    return (v, v, v) if isinstance(v, int) else tuple(int(x) for x in v)


def _t2(v):
    return (v, v) if isinstance(v, int) else tuple(int(x) for x in v)


def deform_sample3d(input, offset, kernel_size, stride=1, padding=0, dilation=1, deformable_groups=1):
    """Returns the deformable "columns" tensor S[b, c, tap, od, oh, ow] (never materialised by the product)."""
    k, s, p, dl = _t3(kernel_size), _t3(stride), _t3(padding), _t3(dilation)
    B, C, D, H, W = input.shape
    dg = deformable_groups
    K = k[0] * k[1] * k[2]
    Do, Ho, Wo = offset.shape[2:]
    No, Ni = Do * Ho * Wo, D * H * W
    dt = input.dtype
    off = offset.reshape(B, dg, K, 3, Do, Ho, Wo)
    bd = (torch.arange(Do) * s[0] - p[0]).view(Do, 1, 1)
    bh = (torch.arange(Ho) * s[1] - p[1]).view(1, Ho, 1)
    bw = (torch.arange(Wo) * s[2] - p[2]).view(1, 1, Wo)
    vol = input.reshape(B, dg, C // dg, Ni)
    cols = []
    for tap, (i, j, kk) in enumerate(itertools.product(range(k[0]), range(k[1]), range(k[2]))):
        qd = (bd + i * dl[0]).to(dt) + off[:, :, tap, 0]
        qh = (bh + j * dl[1]).to(dt) + off[:, :, tap, 1]
        qw = (bw + kk * dl[2]).to(dt) + off[:, :, tap, 2]
        inside = (qd > -1) & (qh > -1) & (qw > -1) & (qd < D) & (qh < H) & (qw < W)
        d0, h0, w0 = torch.floor(qd), torch.floor(qh), torch.floor(qw)
        ld, lh, lw = qd - d0, qh - h0, qw - w0
        d0, h0, w0 = d0.long(), h0.long(), w0.long()
        val = 0
        for cd, ch, cw in itertools.product((0, 1), repeat=3):
            zd, zh, zw = d0 + cd, h0 + ch, w0 + cw
            ok = inside & (zd >= 0) & (zd <= D - 1) & (zh >= 0) & (zh <= H - 1) & (zw >= 0) & (zw <= W - 1)
            wt = (ld if cd else 1 - ld) * (lh if ch else 1 - lh) * (lw if cw else 1 - lw)
            lin = (zd.clamp(0, D - 1) * H + zh.clamp(0, H - 1)) * W + zw.clamp(0, W - 1)
            g = vol.gather(3, lin.reshape(B, dg, 1, No).expand(B, dg, C // dg, No))
            val = val + (wt * ok.to(dt)).reshape(B, dg, 1, No) * g
        cols.append(val.reshape(B, C, No))
    return torch.stack(cols, 2).reshape(B, C, K, Do, Ho, Wo)


def deform_conv3d(input, offset, weight, bias, stride=1, padding=0, dilation=1, group=1, deformable_groups=1):
    Cout, Cg, kd, kh, kw = weight.shape
    B, C = input.shape[:2]
    S = deform_sample3d(input, offset, (kd, kh, kw), stride, padding, dilation, deformable_groups)
    Do, Ho, Wo = S.shape[3:]
    K = kd * kh * kw
    col = S.reshape(B, group, Cg * K, Do * Ho * Wo)
    w = weight.reshape(group, Cout // group, Cg * K)
    out = torch.einsum("bgkn,gok->bgon", col, w).reshape(B, Cout, Do, Ho, Wo)
    return out + bias.view(1, -1, 1, 1, 1)


def deform_conv3d_grid_sample(input, offset, weight, bias, stride=1, padding=0, dilation=1, group=1,
                              deformable_groups=1):
    """Third opinion on the forward: F.grid_sample(bilinear, zeros, align_corners=True) per tap.
    Only valid for deformable_groups == 1 (one grid per batch item)."""
    assert deformable_groups == 1
    Cout, Cg, kd, kh, kw = weight.shape
    k, s, p, dl = (kd, kh, kw), _t3(stride), _t3(padding), _t3(dilation)
    B, C, D, H, W = input.shape
    Do, Ho, Wo = offset.shape[2:]
    K = kd * kh * kw
    dt = input.dtype
    off = offset.reshape(B, K, 3, Do, Ho, Wo)
    bd = (torch.arange(Do) * s[0] - p[0]).view(Do, 1, 1)
    bh = (torch.arange(Ho) * s[1] - p[1]).view(1, Ho, 1)
    bw = (torch.arange(Wo) * s[2] - p[2]).view(1, 1, Wo)
    cols = []
    for tap, (i, j, kk) in enumerate(itertools.product(range(kd), range(kh), range(kw))):
        qd = (bd + i * dl[0]).to(dt) + off[:, tap, 0]
        qh = (bh + j * dl[1]).to(dt) + off[:, tap, 1]
        qw = (bw + kk * dl[2]).to(dt) + off[:, tap, 2]
        # align_corners=True: x_norm = 2*x/(size-1) - 1 ; grid last dim is (x=w, y=h, z=d)
        def nrm(q, n):
            return 2 * q / max(n - 1, 1) - 1 if n > 1 else torch.zeros_like(q)
        grid = torch.stack((nrm(qw, W), nrm(qh, H), nrm(qd, D)), -1)
        cols.append(F.grid_sample(input, grid, mode="bilinear", padding_mode="zeros", align_corners=True))
    S = torch.stack(cols, 2)  # B, C, K, Do, Ho, Wo
    col = S.reshape(B, group, Cg * K, Do * Ho * Wo)
    w = weight.reshape(group, Cout // group, Cg * K)
    out = torch.einsum("bgkn,gok->bgon", col, w).reshape(B, Cout, Do, Ho, Wo)
    return out + bias.view(1, -1, 1, 1, 1)


def deform_conv2d(input, offset, weight, bias=None, stride=1, padding=0, dilation=1):
    """torchvision 0.12 deform_conv2d(mask=None) restated with gather + autograd."""
    s, p, dl = _t2(stride), _t2(padding), _t2(dilation)
    B, C, H, W = input.shape
    Cout, Cg, kh, kw = weight.shape
    group = C // Cg
    K = kh * kw
    og = offset.shape[1] // (2 * K)
    Ho, Wo = offset.shape[2:]
    No, Ni = Ho * Wo, H * W
    dt = input.dtype
    off = offset.reshape(B, og, K, 2, Ho, Wo)
    by = (torch.arange(Ho) * s[0] - p[0]).view(Ho, 1)
    bx = (torch.arange(Wo) * s[1] - p[1]).view(1, Wo)
    img = input.reshape(B, og, C // og, Ni)
    cols = []
    for tap, (i, j) in enumerate(itertools.product(range(kh), range(kw))):
        qy = (by + i * dl[0]).to(dt) + off[:, :, tap, 0]
        qx = (bx + j * dl[1]).to(dt) + off[:, :, tap, 1]
        inside = (qy > -1) & (qx > -1) & (qy < H) & (qx < W)
        y0, x0 = torch.floor(qy), torch.floor(qx)
        ly, lx = qy - y0, qx - x0
        y0, x0 = y0.long(), x0.long()
        val = 0
        for cy, cx in itertools.product((0, 1), repeat=2):
            zy, zx = y0 + cy, x0 + cx
            ok = inside & (zy >= 0) & (zy <= H - 1) & (zx >= 0) & (zx <= W - 1)
            wt = (ly if cy else 1 - ly) * (lx if cx else 1 - lx)
            lin = zy.clamp(0, H - 1) * W + zx.clamp(0, W - 1)
            g = img.gather(3, lin.reshape(B, og, 1, No).expand(B, og, C // og, No))
            val = val + (wt * ok.to(dt)).reshape(B, og, 1, No) * g
        cols.append(val.reshape(B, C, No))
    S = torch.stack(cols, 2)  # B, C, K, No
    col = S.reshape(B, group, Cg * K, No)
    w = weight.reshape(group, Cout // group, Cg * K)
    out = torch.einsum("bgkn,gok->bgon", col, w).reshape(B, Cout, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out
