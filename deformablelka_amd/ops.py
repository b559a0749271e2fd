"""Tensor-level wrappers over the C-ABI.  PyTorch is used for device memory and streams only."""
from __future__ import annotations

import ctypes
from ctypes import byref
from typing import Optional, Sequence, Tuple

import torch

from . import _lib as L


def _triple(v) -> Tuple[int, int, int]:
    if isinstance(v, int):
        return (v, v, v)
    v = tuple(int(x) for x in v)
    if len(v) != 3:
        raise ValueError(f"expected an int or a 3-tuple, got {v}")
    return v


def _pair(v) -> Tuple[int, int]:
    if isinstance(v, int):
        return (v, v)
    v = tuple(int(x) for x in v)
    if len(v) != 2:
        raise ValueError(f"expected an int or a 2-tuple, got {v}")
    return v


def _geom(x_shape, cout, k, s, p, d, group, dg=1, im2col_step=64) -> L.ConvGeom:
    B, C, D, H, W = (int(v) for v in x_shape)
    return L.ConvGeom(B, C, D, H, W, int(cout), *k, *s, *p, *d, int(group), int(dg), int(im2col_step))


def _out_dims(g: L.ConvGeom):
    f = L.get_lib().dlka_conv_out_size
    return (f(g.D, g.pd, g.dd, g.kd, g.sd), f(g.H, g.ph, g.dh, g.kh, g.sh), f(g.W, g.pw, g.dw, g.kw, g.sw))


# ------------------------------------------------------------------------------------------------------------
# 3-D deformable conv
# ------------------------------------------------------------------------------------------------------------
def deform_conv3d_forward(input, weight, bias, offset, kernel_size, stride, padding, dilation, group, deformable_groups,
                          im2col_step=64):
    """``D3D.deform_conv_forward`` (3D/dcn/src/deform_conv.h:10-47)."""
    # deform_conv_cuda.cu:41-47
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")
    L.require_device(input, weight, bias, offset)
    k, s, p, d = _triple(kernel_size), _triple(stride), _triple(padding), _triple(dilation)
    if tuple(weight.shape[2:5]) != k:  # deform_conv_cuda.cu:72-73
        raise RuntimeError(f"Input shape and kernel shape wont match: ({k} vs {tuple(weight.shape[2:5])}).")
    if input.shape[1] != weight.shape[1] * group:  # :75-76
        raise RuntimeError(f"Input shape and kernel channels wont match: ({input.shape[1]} vs {weight.shape[1] * group}).")
    offset = offset.contiguous()  # the reference does not check offset; a strided view would be misread
    bias = bias.contiguous()
    lib = L.get_lib()
    g = _geom(input.shape, weight.shape[0], k, s, p, d, group, deformable_groups, im2col_step)
    dt = L.dtype_code(input, allow_f64=True)
    Do, Ho, Wo = _out_dims(g)
    if min(Do, Ho, Wo) <= 0:
        L.check(-4, "deform_conv_forward")
    K = k[0] * k[1] * k[2]
    if tuple(offset.shape) != (g.B, deformable_groups * 3 * K, Do, Ho, Wo):
        raise RuntimeError(f"offset shape {tuple(offset.shape)} does not match {(g.B, deformable_groups * 3 * K, Do, Ho, Wo)}")
    out = torch.empty((g.B, g.Cout, Do, Ho, Wo), dtype=input.dtype, device=input.device)
    wsb = lib.dlka_deform_conv3d_forward_workspace(byref(g), dt)
    ws = L.scratch(wsb, input)
    rc = lib.dlka_deform_conv3d_forward(L.ptr(input), L.ptr(offset), L.ptr(weight), L.ptr(bias), L.ptr(out), L.ptr(ws), wsb,
                                        byref(g), dt, L.stream_ptr(input))
    L.check(rc, "deform_conv_forward")
    return out


def deform_conv3d_backward(input, weight, bias, offset, grad_output, kernel_size, stride, padding, dilation, group,
                           deformable_groups, im2col_step=64, need=(True, True, True, True)):
    """``D3D.deform_conv_backward`` (3D/dcn/src/deform_conv.h:49-91) -> (grad_input, grad_offset, grad_weight, grad_bias)."""
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")
    L.require_device(input, weight, bias, offset, grad_output)
    k, s, p, d = _triple(kernel_size), _triple(stride), _triple(padding), _triple(dilation)
    offset = offset.contiguous()
    grad_output = grad_output.contiguous()  # reference: permute(...).contiguous() copies, deform_conv_cuda.cu:228
    lib = L.get_lib()
    g = _geom(input.shape, weight.shape[0], k, s, p, d, group, deformable_groups, im2col_step)
    dt = L.dtype_code(input, allow_f64=True)
    Do, Ho, Wo = _out_dims(g)
    if tuple(grad_output.shape) != (g.B, g.Cout, Do, Ho, Wo):  # deform_conv_cuda.cu:193-200
        raise RuntimeError(f"Input shape and grad_out shape wont match: ({(g.B, g.Cout, Do, Ho, Wo)} vs {tuple(grad_output.shape)}).")
    gi = torch.empty_like(input) if need[0] else None
    go = torch.empty_like(offset) if need[1] else None
    gw = torch.empty_like(weight) if need[2] else None
    gb = torch.empty_like(bias, memory_format=torch.contiguous_format) if need[3] else None
    wsb = lib.dlka_deform_conv3d_backward_workspace(byref(g), dt)
    ws = L.scratch(wsb, input)
    rc = lib.dlka_deform_conv3d_backward(L.ptr(input), L.ptr(offset), L.ptr(weight), L.ptr(grad_output), L.ptr(gi), L.ptr(go),
                                         L.ptr(gw), L.ptr(gb), L.ptr(ws), wsb, byref(g), dt, L.stream_ptr(input))
    L.check(rc, "deform_conv_backward")
    return gi, go, gw, gb


def deform_conv3d_sample_index(offset, in_size: Sequence[int], kernel_size, stride, padding, dilation, deformable_groups=1, path=0):
    """floor() indices [B,dg,K,Do,Ho,Wo,3] (int32) and guard mask [B,dg,K,Do,Ho,Wo] (uint8).
    ``path``: 0 = ``sample_cell3`` on its own (the one rule every kernel calls; the fixed-point grad_input kernel calls it directly),
    1 = through ``setup_tap`` (general kernels), 2 = through ``gather_describe3`` (channels-last gathers), 3 = through ``lane_tap``
    (grad_input window kernels); idx = 0 where the mask is 0."""
    L.require_device(offset)
    k, s, p, d = _triple(kernel_size), _triple(stride), _triple(padding), _triple(dilation)
    offset = offset.contiguous()
    B = offset.shape[0]
    g = _geom((B, deformable_groups, *in_size), deformable_groups, k, s, p, d, 1, deformable_groups, 64)
    Do, Ho, Wo = _out_dims(g)
    K = k[0] * k[1] * k[2]
    idx = torch.empty((B, deformable_groups, K, Do, Ho, Wo, 3), dtype=torch.int32, device=offset.device)
    mask = torch.empty((B, deformable_groups, K, Do, Ho, Wo), dtype=torch.uint8, device=offset.device)
    rc = L.get_lib().dlka_deform_conv3d_sample_index_path(L.ptr(offset), L.ptr(idx), L.ptr(mask), byref(g), L.dtype_code(offset),
                                                          int(path), L.stream_ptr(offset))
    L.check(rc, "deform_conv3d_sample_index")
    return idx, mask


# ------------------------------------------------------------------------------------------------------------
# 2-D deformable conv (torchvision semantics)
# ------------------------------------------------------------------------------------------------------------
def _geom2d(x_shape, weight_shape, s, p, d, offset_channels):
    B, C, H, W = (int(v) for v in x_shape)
    Cout, Cg, kh, kw = (int(v) for v in weight_shape)
    if C % Cg != 0:
        raise RuntimeError("input channels must be divisible by weight.shape[1]")
    og = offset_channels // (2 * kh * kw)
    if og == 0 or offset_channels != og * 2 * kh * kw:
        raise RuntimeError(f"offset.shape[1] = {offset_channels} is not a multiple of 2*kh*kw = {2 * kh * kw}")
    return L.ConvGeom(B, C, 1, H, W, Cout, 1, kh, kw, 1, s[0], s[1], 0, p[0], p[1], 1, d[0], d[1], C // Cg, og, 64)


def deform_conv2d_forward(input, offset, weight, bias=None, stride=1, padding=0, dilation=1):
    L.require_device(input, offset, weight, bias)
    s, p, d = _pair(stride), _pair(padding), _pair(dilation)
    input, offset, weight = input.contiguous(), offset.contiguous(), weight.contiguous()
    bias = None if bias is None else bias.contiguous()
    g = _geom2d(input.shape, weight.shape, s, p, d, offset.shape[1])
    lib = L.get_lib()
    dt = L.dtype_code(input, allow_f64=True)
    _, Ho, Wo = _out_dims(g)
    if tuple(offset.shape[2:]) != (Ho, Wo):
        raise RuntimeError(f"offset spatial size {tuple(offset.shape[2:])} does not match output {(Ho, Wo)}")
    out = torch.empty((g.B, g.Cout, Ho, Wo), dtype=input.dtype, device=input.device)
    wsb = lib.dlka_deform_conv2d_forward_workspace(byref(g), dt)
    ws = L.scratch(wsb, input)
    rc = lib.dlka_deform_conv2d_forward(L.ptr(input), L.ptr(offset), L.ptr(weight), L.ptr(bias), L.ptr(out), L.ptr(ws), wsb,
                                        byref(g), dt, L.stream_ptr(input))
    L.check(rc, "deform_conv2d")
    return out


def deform_conv2d_backward(input, offset, weight, grad_output, stride=1, padding=0, dilation=1, with_bias=False,
                           need=(True, True, True)):
    L.require_device(input, offset, weight, grad_output)
    s, p, d = _pair(stride), _pair(padding), _pair(dilation)
    input, offset, weight, grad_output = input.contiguous(), offset.contiguous(), weight.contiguous(), grad_output.contiguous()
    g = _geom2d(input.shape, weight.shape, s, p, d, offset.shape[1])
    lib = L.get_lib()
    dt = L.dtype_code(input, allow_f64=True)
    gi = torch.empty_like(input) if need[0] else None
    go = torch.empty_like(offset) if need[1] else None
    gw = torch.empty_like(weight) if need[2] else None
    gb = torch.empty((g.Cout,), dtype=input.dtype, device=input.device) if with_bias else None
    wsb = lib.dlka_deform_conv2d_backward_workspace(byref(g), dt)
    ws = L.scratch(wsb, input)
    rc = lib.dlka_deform_conv2d_backward(L.ptr(input), L.ptr(offset), L.ptr(weight), L.ptr(grad_output), L.ptr(gi), L.ptr(go),
                                         L.ptr(gw), L.ptr(gb), L.ptr(ws), wsb, byref(g), dt, L.stream_ptr(input))
    L.check(rc, "deform_conv2d backward")
    return gi, go, gw, gb


def deform_conv2d_sample_index(offset, in_size: Sequence[int], kernel_size, stride=1, padding=0, dilation=1, offset_groups=1, path=0):
    """2-D index-parity entry: floor cell [B,og,K,Ho,Wo,2] (int32; 0 outside `reach`) and mask [B,og,K,Ho,Wo] (uint8: bit 0 = sample inside
    the guard, bit 1 = reach).  ``path``: 0 = ``sample_cell2`` on its own, 1 = ``setup_tap<2>`` (general kernels), 2 = ``describe2``
    (channels-last depthwise kernels)."""
    L.require_device(offset)
    k, s, p, d = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
    offset = offset.contiguous()
    B, H, W = int(offset.shape[0]), int(in_size[0]), int(in_size[1])
    g = L.ConvGeom(B, offset_groups, 1, H, W, offset_groups, 1, k[0], k[1], 1, s[0], s[1], 0, p[0], p[1], 1, d[0], d[1], 1, offset_groups, 64)
    _, Ho, Wo = _out_dims(g)
    K = k[0] * k[1]
    if tuple(offset.shape) != (B, offset_groups * 2 * K, Ho, Wo):
        raise RuntimeError(f"offset shape {tuple(offset.shape)} does not match {(B, offset_groups * 2 * K, Ho, Wo)}")
    idx = torch.empty((B, offset_groups, K, Ho, Wo, 2), dtype=torch.int32, device=offset.device)
    mask = torch.empty((B, offset_groups, K, Ho, Wo), dtype=torch.uint8, device=offset.device)
    rc = L.get_lib().dlka_deform_conv2d_sample_index_path(L.ptr(offset), L.ptr(idx), L.ptr(mask), byref(g), L.dtype_code(offset), int(path),
                                                          L.stream_ptr(offset))
    L.check(rc, "deform_conv2d_sample_index")
    return idx, mask


def _geom_dw2d_cl(x, weight, padding, dilation):
    B, H, W, C = (int(v) for v in x.shape)
    p, d = _pair(padding), _pair(dilation)
    if tuple(weight.shape[:2]) != (C, 1):
        raise RuntimeError(f"depthwise weight [C][1][kh][kw] expected, got {tuple(weight.shape)}")
    kh, kw = int(weight.shape[2]), int(weight.shape[3])
    return L.ConvGeom(B, C, 1, H, W, C, 1, kh, kw, 1, 1, 1, 0, p[0], p[1], 1, d[0], d[1], C, 1, 64)


def deform_dwconv2d_forward_cl(x, offset, weight, padding, dilation=1):
    """The 2-D D-LKA block's depthwise deformable conv on its own, channels-last (cl_ddw2d.hip): x [B,H,W,C], offset [B,2K,H,W] planar
    ((dy, dx) per tap, torchvision layout), weight [C,1,kh,kw] -> out [B,H,W,C]."""
    L.require_device(x, offset, weight)
    x, offset, weight = x.contiguous(), offset.contiguous(), weight.contiguous()
    g = _geom_dw2d_cl(x, weight, padding, dilation)
    lib, dt = L.get_lib(), L.dtype_code(x)
    out = torch.empty_like(x)
    wsb = lib.dlka_deform_dwconv2d_cl_workspace(byref(g), dt, 0)
    ws = L.scratch(wsb, x)
    rc = lib.dlka_deform_dwconv2d_forward_cl(L.ptr(x), L.ptr(offset), L.ptr(weight), L.ptr(out), L.ptr(ws), wsb, byref(g), dt, L.stream_ptr(x))
    L.check(rc, "deform_dwconv2d_forward_cl")
    return out


def deform_dwconv2d_backward_cl(x, offset, weight, grad_out, padding, dilation=1):
    """-> (grad_x [B,H,W,C], grad_offset [B,2K,H,W], grad_weight [C,1,kh,kw])."""
    L.require_device(x, offset, weight, grad_out)
    x, offset, weight, grad_out = x.contiguous(), offset.contiguous(), weight.contiguous(), grad_out.contiguous()
    g = _geom_dw2d_cl(x, weight, padding, dilation)
    lib, dt = L.get_lib(), L.dtype_code(x)
    gx, go, gw = torch.empty_like(x), torch.empty_like(offset), torch.empty_like(weight)
    wsb = lib.dlka_deform_dwconv2d_cl_workspace(byref(g), dt, 1)
    ws = L.scratch(wsb, x)
    rc = lib.dlka_deform_dwconv2d_backward_cl(L.ptr(x), L.ptr(offset), L.ptr(weight), L.ptr(grad_out), L.ptr(gx), L.ptr(go), L.ptr(gw), L.ptr(ws),
                                              wsb, byref(g), dt, L.stream_ptr(x))
    L.check(rc, "deform_dwconv2d_backward_cl")
    return gx, go, gw


# ------------------------------------------------------------------------------------------------------------
# plain conv
# ------------------------------------------------------------------------------------------------------------
def conv3d_forward(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    L.require_device(input, weight, bias)
    s, p, d = _triple(stride), _triple(padding), _triple(dilation)
    input, weight = input.contiguous(), weight.contiguous()
    bias = None if bias is None else bias.contiguous()
    g = _geom(input.shape, weight.shape[0], tuple(weight.shape[2:5]), s, p, d, groups)
    lib = L.get_lib()
    dt = L.dtype_code(input, allow_f64=True)
    Do, Ho, Wo = _out_dims(g)
    out = torch.empty((g.B, g.Cout, Do, Ho, Wo), dtype=input.dtype, device=input.device)
    wsb = lib.dlka_conv3d_forward_workspace(byref(g), dt)
    ws = L.scratch(wsb, input)
    rc = lib.dlka_conv3d_forward(L.ptr(input), L.ptr(weight), L.ptr(bias), L.ptr(out), L.ptr(ws), wsb, byref(g), dt,
                                 L.stream_ptr(input))
    L.check(rc, "conv3d")
    return out


def conv3d_backward(input, weight, grad_output, stride=1, padding=0, dilation=1, groups=1, need=(True, True, True)):
    L.require_device(input, weight, grad_output)
    s, p, d = _triple(stride), _triple(padding), _triple(dilation)
    input, weight, grad_output = input.contiguous(), weight.contiguous(), grad_output.contiguous()
    g = _geom(input.shape, weight.shape[0], tuple(weight.shape[2:5]), s, p, d, groups)
    lib = L.get_lib()
    dt = L.dtype_code(input, allow_f64=True)
    gi = torch.empty_like(input) if need[0] else None
    gw = torch.empty_like(weight) if need[1] else None
    gb = torch.empty((g.Cout,), dtype=input.dtype, device=input.device) if need[2] else None
    wsb = lib.dlka_conv3d_backward_workspace(byref(g), dt)
    ws = L.scratch(wsb, input)
    rc = lib.dlka_conv3d_backward(L.ptr(input), L.ptr(weight), L.ptr(grad_output), L.ptr(gi), L.ptr(gw), L.ptr(gb), L.ptr(ws), wsb,
                                  byref(g), dt, L.stream_ptr(input))
    L.check(rc, "conv3d backward")
    return gi, gw, gb


def gelu_forward(x):
    L.require_device(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    L.check(L.get_lib().dlka_gelu_forward(L.ptr(x), L.ptr(y), x.numel(), L.dtype_code(x), L.stream_ptr(x)), "gelu")
    return y


def gelu_backward(x, gy):
    L.require_device(x, gy)
    x, gy = x.contiguous(), gy.contiguous()
    gx = torch.empty_like(x)
    L.check(L.get_lib().dlka_gelu_backward(L.ptr(x), L.ptr(gy), L.ptr(gx), x.numel(), L.dtype_code(x), L.stream_ptr(x)), "gelu bwd")
    return gx


# ------------------------------------------------------------------------------------------------------------
# planar (NCDHW) plumbing of the full net: BatchNorm3d in training mode, 1x1x1 convs on few channels (csrc/planar_ops.hip)
# ------------------------------------------------------------------------------------------------------------
PLANAR_PW_CIN = (1, 2, 4, 8, 14, 16, 32)


def batchnorm_planar_forward(x, weight, bias, eps=1e-5, out=None):
    """x (B, C, *spatial) fp32 contiguous -> y, stats (4, C) = mean, rstd, unbiased variance, mean - pivot (batch statistics).
    out: a contiguous fp32 tensor of x's element count (any shape) that receives y instead of a fresh tensor of x's shape."""
    L.require_device(x)
    B, C = x.shape[:2]
    N = x[0, 0].numel()
    if out is None:
        y = torch.empty_like(x)
    else:
        assert out.is_contiguous() and out.numel() == x.numel() and out.dtype == x.dtype and out.device == x.device
        y = out
    stats = torch.empty(4, C, dtype=torch.float32, device=x.device)
    scratch = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    L.check(L.get_lib().dlka_batchnorm_planar_forward(L.ptr(x), L.ptr(weight), L.ptr(bias), L.ptr(stats), L.ptr(y), L.ptr(scratch), B, C, N, float(eps),
                                                      L.stream_ptr(x)), "batchnorm_planar_forward")
    return y, stats


def batchnorm_planar_backward(g, x, weight, stats, affine=True):
    L.require_device(x, g)
    B, C = x.shape[:2]
    N = x[0, 0].numel()
    gx = torch.empty_like(x)
    gw = torch.empty(C, dtype=torch.float32, device=x.device) if affine else None
    gb = torch.empty(C, dtype=torch.float32, device=x.device) if affine else None
    scratch = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    L.check(L.get_lib().dlka_batchnorm_planar_backward(L.ptr(g), L.ptr(x), L.ptr(weight), L.ptr(stats), L.ptr(gx), L.ptr(gw), L.ptr(gb), L.ptr(scratch), B, C, N,
                                                       L.stream_ptr(x)), "batchnorm_planar_backward")
    return gx, gw, gb


def pointwise_planar_supported(x, weight, need_weight_grad=True, need_input_grad=None) -> bool:
    """Whether the planar 1x1x1 kernels cover this call INCLUDING the backward pass it may need.  The forward kernel is instantiated on Cin (the
    channel count it holds in registers), the data-gradient kernel on Cout (``pl_pw_bwd_data_kernel<CO>``, planar_ops.hip: the same menu), the weight
    gradient takes Cout <= 16.  need_input_grad: None = ``x.requires_grad`` under grad mode."""
    Cout, Cin = weight.shape[:2]
    if need_input_grad is None:
        need_input_grad = bool(x.requires_grad and torch.is_grad_enabled())
    if not (x.dtype == torch.float32 and weight.dtype == torch.float32 and Cin in PLANAR_PW_CIN and x[0, 0].numel() % 4 == 0 and x.shape[0] <= 65535):
        return False
    if need_weight_grad:
        return Cout <= 16 and Cout in PLANAR_PW_CIN
    if need_input_grad:
        return Cout in PLANAR_PW_CIN
    return Cout <= 64


def pointwise_planar_forward(x, weight, bias=None):
    """1x1x1 conv: x (B, Cin, *spatial) fp32 contiguous, weight (Cout, Cin, 1, 1, 1)."""
    L.require_device(x, weight)
    B, Cin = x.shape[:2]
    Cout = weight.shape[0]
    y = torch.empty((B, Cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    L.check(L.get_lib().dlka_pointwise_planar_forward(L.ptr(x), L.ptr(weight), L.ptr(bias), L.ptr(y), B, Cin, Cout, x[0, 0].numel(), L.stream_ptr(x)),
            "pointwise_planar_forward")
    return y


def pointwise_planar_backward(x, weight, g, need=(True, True, True)):
    L.require_device(x, weight, g)
    B, Cin = x.shape[:2]
    Cout = weight.shape[0]
    gx = torch.empty_like(x) if need[0] else None
    gw = torch.empty_like(weight) if need[1] else None
    gb = torch.empty(Cout, dtype=torch.float32, device=x.device) if (need[1] and need[2]) else None
    if need[0] or need[1]:
        L.check(L.get_lib().dlka_pointwise_planar_backward(L.ptr(x), L.ptr(weight), L.ptr(g), L.ptr(gx), L.ptr(gw), L.ptr(gb), B, Cin, Cout, x[0, 0].numel(),
                                                           L.stream_ptr(x)), "pointwise_planar_backward")
    if need[2] and gb is None:   # frozen weight, trainable bias (fine-tuning a head): the bias gradient rides in the weight-gradient kernel, which did
        gb = g.sum(dim=[0] + list(range(2, g.dim())))   # not run — a plain reduction of grad_out (not hot: a Cout-length result)
    return gx, gw, gb


# ------------------------------------------------------------------------------------------------------------
# whole blocks
# ------------------------------------------------------------------------------------------------------------
def _ptr_struct(cls, fields, tensors):
    st = cls()
    for n, t in zip(fields, tensors):
        setattr(st, n, t.data_ptr())
    return st


def lka3d_attention_forward(x, params: Sequence[torch.Tensor]):
    """x: [B,C,D,H,W]; params: the 14 tensors in ``_lib.LKA3D_FIELDS`` order. Returns (y, saved)."""
    L.require_device(x, *params)
    x = x.contiguous()
    params = [t.contiguous() for t in params]
    B, C, D, H, W = (int(v) for v in x.shape)
    lib = L.get_lib()
    dt = L.dtype_code(x)
    sb, wb = lib.dlka_lka3d_saved_bytes(B, C, D, H, W, dt), lib.dlka_lka3d_workspace_bytes(B, C, D, H, W, dt)
    if sb == 0:
        L.check(-4, "lka3d_attention_forward")
    saved, ws = L.scratch(sb, x), L.scratch(wb, x)
    y = torch.empty_like(x)
    ps = _ptr_struct(L.Lka3dPtrs, L.LKA3D_FIELDS, params)
    rc = lib.dlka_lka3d_attention_forward(L.ptr(x), byref(ps), L.ptr(y), L.ptr(saved), sb, L.ptr(ws), wb, B, C, D, H, W, dt,
                                          L.stream_ptr(x))
    L.check(rc, "lka3d_attention_forward")
    return y, saved


def lka3d_attention_backward(x, params, grad_y, saved):
    L.require_device(x, grad_y, saved, *params)
    x, grad_y = x.contiguous(), grad_y.contiguous()
    params = [t.contiguous() for t in params]
    B, C, D, H, W = (int(v) for v in x.shape)
    lib = L.get_lib()
    dt = L.dtype_code(x)
    wb = lib.dlka_lka3d_workspace_bytes(B, C, D, H, W, dt)
    ws = L.scratch(wb, x)
    gx = torch.empty_like(x)
    grads = [torch.empty_like(t) for t in params]
    ps = _ptr_struct(L.Lka3dPtrs, L.LKA3D_FIELDS, params)
    gs = _ptr_struct(L.Lka3dPtrs, L.LKA3D_FIELDS, grads)
    rc = lib.dlka_lka3d_attention_backward(L.ptr(x), byref(ps), L.ptr(grad_y), L.ptr(saved), saved.numel(), L.ptr(gx), byref(gs),
                                           L.ptr(ws), wb, B, C, D, H, W, dt, L.stream_ptr(x))
    L.check(rc, "lka3d_attention_backward")
    return gx, grads


def lka2d_bf16_supported(C: int) -> bool:
    """Widths the DLKA_BF16 2-D block covers (channels-last fast path: C / 32 in {1, 2, 3, 4, 6, 8, 12})."""
    return C % 32 == 0 and C // 32 in (1, 2, 3, 4, 6, 8, 12)


def _lka2d_params(x, params):
    if x.dtype == torch.bfloat16:   # DLKA_BF16: bf16 ACTIVATIONS, fp32 master parameters (include/dlka.h)
        if not lka2d_bf16_supported(int(x.shape[1])):
            raise RuntimeError(f"deformable_LKA_Attention with bfloat16 activations needs C / 32 in {{1,2,3,4,6,8,12}}, got C={int(x.shape[1])}")
        return [_fp32_param(t) for t in params]
    return [t.contiguous() for t in params]


def lka2d_attention_forward(x, params: Sequence[torch.Tensor]):
    L.require_device(x, *params)
    x = x.contiguous()
    params = _lka2d_params(x, params)
    B, C, H, W = (int(v) for v in x.shape)
    lib = L.get_lib()
    dt = L.dtype_code(x)
    sb, wb = lib.dlka_lka2d_saved_bytes(B, C, H, W, dt), lib.dlka_lka2d_workspace_bytes(B, C, H, W, dt)
    if sb == 0:
        L.check(-4, "lka2d_attention_forward")
    saved, ws = L.scratch(sb, x), L.scratch(wb, x)
    y = torch.empty_like(x)
    ps = _ptr_struct(L.Lka2dPtrs, L.LKA2D_FIELDS, params)
    rc = lib.dlka_lka2d_attention_forward(L.ptr(x), byref(ps), L.ptr(y), L.ptr(saved), sb, L.ptr(ws), wb, B, C, H, W, dt,
                                          L.stream_ptr(x))
    L.check(rc, "lka2d_attention_forward")
    return y, saved


def lka2d_saved_offsets(saved, x):
    """The two predicted offset tensors ([B, 50, H, W] of conv0, [B, 98, H, W] of conv_spatial; torchvision's planar layout) inside the opaque
    ``saved`` buffer of ``lka2d_attention_forward(x, ...)`` — for the path such a call takes now (``dlka_lka2d_saved_offsets``).  Diagnostics:
    the parity tests' cell-flip analysis reads them."""
    B, C, H, W = (int(v) for v in x.shape)
    lib = L.get_lib()
    offs = (ctypes.c_size_t * 2)()
    eb = ctypes.c_int(0)
    L.check(lib.dlka_lka2d_saved_offsets(B, C, H, W, L.dtype_code(x), offs, byref(eb)), "lka2d_saved_offsets")
    dt = torch.float32 if eb.value == 4 else torch.bfloat16
    out = []
    for o, ch in zip(offs, (50, 98)):
        n = B * ch * H * W * eb.value
        out.append(saved[int(o):int(o) + n].view(dt).view(B, ch, H, W))
    return out


def lka2d_attention_backward(x, params, grad_y, saved):
    L.require_device(x, grad_y, saved, *params)
    x, grad_y = x.contiguous(), grad_y.to(x.dtype).contiguous()
    params = _lka2d_params(x, params)
    B, C, H, W = (int(v) for v in x.shape)
    lib = L.get_lib()
    dt = L.dtype_code(x)
    wb = lib.dlka_lka2d_workspace_bytes(B, C, H, W, dt)
    ws = L.scratch(wb, x)
    gx = torch.empty_like(x)
    grads = [torch.empty_like(t) for t in params]
    ps = _ptr_struct(L.Lka2dPtrs, L.LKA2D_FIELDS, params)
    gs = _ptr_struct(L.Lka2dPtrs, L.LKA2D_FIELDS, grads)
    rc = lib.dlka_lka2d_attention_backward(L.ptr(x), byref(ps), L.ptr(grad_y), L.ptr(saved), saved.numel(), L.ptr(gx), byref(gs),
                                           L.ptr(ws), wb, B, C, H, W, dt, L.stream_ptr(x))
    L.check(rc, "lka2d_attention_backward")
    return gx, grads


# ------------------------------------------------------------------------------------------------------------
# channels-last fast path (fp32): x [B, D, H, W, C]
# ------------------------------------------------------------------------------------------------------------
def _geom_cl(x_shape, cout, k, p, d, group, dg=1):
    B, D, H, W, C = (int(v) for v in x_shape)
    return L.ConvGeom(B, C, D, H, W, int(cout), *k, 1, 1, 1, *p, *d, int(group), int(dg), 64)


def conv3d_forward_cl(x, weight, bias=None, padding=0, dilation=1, groups=1, out_planar=False):
    L.require_device(x, weight, bias)
    p, d = _triple(padding), _triple(dilation)
    x, weight = x.contiguous(), weight.contiguous()
    bias = None if bias is None else bias.contiguous()
    g = _geom_cl(x.shape, weight.shape[0], tuple(weight.shape[2:5]), p, d, groups)
    lib = L.get_lib()
    dt = L.dtype_code(x)
    B, D, H, W, _ = x.shape
    shape = (B, g.Cout, D, H, W) if out_planar else (B, D, H, W, g.Cout)
    out = torch.empty(shape, dtype=x.dtype, device=x.device)
    wsb = lib.dlka_conv3d_cl_workspace(byref(g), dt, 0)
    ws = L.scratch(wsb, x)
    rc = lib.dlka_conv3d_forward_cl(L.ptr(x), L.ptr(weight), L.ptr(bias), L.ptr(out), int(out_planar), L.ptr(ws), wsb, byref(g), dt,
                                    L.stream_ptr(x))
    L.check(rc, "conv3d_forward_cl")
    return out


def conv3d_backward_cl(x, weight, grad_out, padding=0, dilation=1, groups=1, grad_out_planar=False):
    L.require_device(x, weight, grad_out)
    p, d = _triple(padding), _triple(dilation)
    x, weight, grad_out = x.contiguous(), weight.contiguous(), grad_out.contiguous()
    g = _geom_cl(x.shape, weight.shape[0], tuple(weight.shape[2:5]), p, d, groups)
    lib = L.get_lib()
    dt = L.dtype_code(x)
    gi, gw = torch.empty_like(x), torch.empty_like(weight)
    gb = torch.empty((g.Cout,), dtype=x.dtype, device=x.device)
    wsb = lib.dlka_conv3d_cl_workspace(byref(g), dt, 1)
    ws = L.scratch(wsb, x)
    rc = lib.dlka_conv3d_backward_cl(L.ptr(x), L.ptr(weight), L.ptr(grad_out), int(grad_out_planar), L.ptr(gi), L.ptr(gw), L.ptr(gb),
                                     L.ptr(ws), wsb, byref(g), dt, L.stream_ptr(x))
    L.check(rc, "conv3d_backward_cl")
    return gi, gw, gb


def deform_conv3d_forward_cl(x, offset, weight, bias, padding=1, dilation=1):
    """x [B,D,H,W,C] channels-last, offset [B,3K,D,H,W] planar -> out [B,D,H,W,Cout]."""
    L.require_device(x, offset, weight, bias)
    p, d = _triple(padding), _triple(dilation)
    x, offset, weight, bias = x.contiguous(), offset.contiguous(), weight.contiguous(), bias.contiguous()
    g = _geom_cl(x.shape, weight.shape[0], tuple(weight.shape[2:5]), p, d, 1)
    lib = L.get_lib()
    dt = L.dtype_code(x)
    B, D, H, W, _ = x.shape
    out = torch.empty((B, D, H, W, g.Cout), dtype=x.dtype, device=x.device)
    wsb = lib.dlka_deform_conv3d_cl_workspace(byref(g), dt, 0)
    ws = L.scratch(wsb, x)
    rc = lib.dlka_deform_conv3d_forward_cl(L.ptr(x), L.ptr(offset), L.ptr(weight), L.ptr(bias), L.ptr(out), L.ptr(ws), wsb, byref(g), dt,
                                           L.stream_ptr(x))
    L.check(rc, "deform_conv3d_forward_cl")
    return out


def deform_conv3d_backward_cl(x, offset, weight, grad_out, padding=1, dilation=1):
    L.require_device(x, offset, weight, grad_out)
    p, d = _triple(padding), _triple(dilation)
    x, offset, weight, grad_out = x.contiguous(), offset.contiguous(), weight.contiguous(), grad_out.contiguous()
    g = _geom_cl(x.shape, weight.shape[0], tuple(weight.shape[2:5]), p, d, 1)
    lib = L.get_lib()
    dt = L.dtype_code(x)
    gi, go, gw = torch.empty_like(x), torch.empty_like(offset), torch.empty_like(weight)
    gb = torch.empty((g.Cout,), dtype=x.dtype, device=x.device)
    wsb = lib.dlka_deform_conv3d_cl_workspace(byref(g), dt, 1)
    ws = L.scratch(wsb, x)
    rc = lib.dlka_deform_conv3d_backward_cl(L.ptr(x), L.ptr(offset), L.ptr(weight), L.ptr(grad_out), L.ptr(gi), L.ptr(go), L.ptr(gw),
                                            L.ptr(gb), L.ptr(ws), wsb, byref(g), dt, L.stream_ptr(x))
    L.check(rc, "deform_conv3d_backward_cl")
    return gi, go, gw, gb


def lka3d_tokens_supported(x, B, C, D, H, W, variant=0) -> bool:
    """x: a tensor or a torch dtype; variant: 0 = Synapse depthwise pair, 1 = ACDC (include/dlka.h: dlka_lka3d_variant).  float32, or bfloat16 activations (DLKA_BF16: bf16 storage of x / y / saved activations, fp32 parameters,
    offsets and accumulation)."""
    dt = x if isinstance(x, torch.dtype) else x.dtype
    if dt not in (torch.float32, torch.bfloat16):
        return False
    return bool(L.get_lib().dlka_lka3d_tokens_supported_v(B, C, D, H, W, L.DLKA_F32 if dt == torch.float32 else L.DLKA_BF16, int(variant)))


def autocast_activation_dtype(x):
    """The autocast policy of the token-layout D-LKA block (the reference registers none): inside ``torch.autocast(dtype=
    torch.bfloat16)`` the block runs on bf16 activations with fp32 parameters / accumulation; anything else keeps x's dtype."""
    try:
        on = torch.is_autocast_enabled(x.device.type)
        dt = torch.get_autocast_dtype(x.device.type) if on else None
    except (TypeError, RuntimeError):
        return x.dtype
    return torch.bfloat16 if (on and dt == torch.bfloat16 and x.dtype in (torch.float32, torch.bfloat16)) else x.dtype



def _fp32_param(t):
    """Parameters of the token-layout block are fp32 masters whatever the activation dtype is."""
    if t.dtype != torch.float32:
        raise RuntimeError(f"the token-layout D-LKA block keeps its parameters in float32 (bf16 is an ACTIVATION dtype); got {t.dtype}")
    return t.contiguous()


def lka3d_attention_tokens_forward(x, params, dims, variant=0):
    """x: [B, N, C] tokens, dims = (D, H, W) spatial extents (the reference's H, W, D). Returns (y, saved)."""
    L.require_device(x, *params)
    x = x.contiguous()
    params = [_fp32_param(t) for t in params]
    B, N, C = (int(v) for v in x.shape)
    D, H, W = (int(v) for v in dims)
    assert N == D * H * W
    lib = L.get_lib()
    dt = L.dtype_code(x)
    v = int(variant)
    sb, wb = lib.dlka_lka3d_tokens_saved_bytes_v(B, C, D, H, W, dt, v), lib.dlka_lka3d_tokens_workspace_bytes_v(B, C, D, H, W, dt, v)
    if sb == 0:
        L.check(-8, "lka3d_attention_tokens_forward")
    saved, ws = L.scratch(sb, x), L.scratch(wb, x)
    y = torch.empty_like(x)
    ps = _ptr_struct(L.Lka3dPtrs, L.LKA3D_FIELDS, params)
    rc = lib.dlka_lka3d_attention_tokens_forward_v(L.ptr(x), byref(ps), L.ptr(y), L.ptr(saved), sb, L.ptr(ws), wb, B, C, D, H, W, dt, v,
                                                   L.stream_ptr(x))
    L.check(rc, "lka3d_attention_tokens_forward")
    return y, saved


def lka3d_tokens_saved_offsets(saved, B, C, dims, act_dtype=torch.float32, variant=0):
    """The predicted sampling offsets [B, 81, D, H, W] (fp32, the reference's planar layout) inside the opaque ``saved`` buffer of a token-layout
    forward call (``dlka_lka3d_tokens_saved_offsets_v``; the NCDHW entry point's fp32 ``saved`` starts with the same four tensors).  Diagnostics:
    bench.py's health check and the cell-flip analysis of the parity tests read them."""
    D, H, W = (int(v) for v in dims)
    off = ctypes.c_size_t(0)
    dt = L.DLKA_F32 if act_dtype == torch.float32 else L.DLKA_BF16
    lib = L.get_lib()
    if lib.dlka_lka3d_tokens_saved_offsets_v(B, C, D, H, W, dt, int(variant), byref(off)) != 0:   # widths outside the token path (general NCDHW entry)
        sb = 4 if act_dtype == torch.float32 else 2
        off = ctypes.c_size_t(4 * ((B * C * D * H * W * sb + 255) & ~255))
    o = int(off.value)
    return saved[o:o + B * 81 * D * H * W * 4].view(torch.float32).view(B, 81, D, H, W)


def tblock3d_saved_offsets(saved, B, C, dims, variant=0, lka_bf16=False):
    """The same tensor inside the ``saved`` buffer of a wrapper-block forward call (``tblock3d_forward``)."""
    D, H, W = (int(v) for v in dims)
    off = ctypes.c_size_t(0)
    L.check(L.get_lib().dlka_tblock3d_saved_offsets_v(B, C, D, H, W, L.DLKA_BF16 if lka_bf16 else L.DLKA_F32, int(variant), byref(off)), "tblock3d_saved_offsets")
    o = int(off.value)
    return saved[o:o + B * 81 * D * H * W * 4].view(torch.float32).view(B, 81, D, H, W)


def tblock3d_saved_activation_signs(saved, B, C, dims, variant=0, lka_bf16=False):
    """(a1 > 0, rd > 0, rd != 0) as bool tensors [B, N, C]: the activation pattern of UnetResBlock's two LeakyReLUs in the forward call that wrote ``saved``
    (``dlka_tblock3d_saved_activations_v``; rd carries the Dropout3d multipliers: a dropped channel is all zeros, hence the third tensor).  Diagnostics for
    the parity tests' kink analysis."""
    D, H, W = (int(v) for v in dims)
    offs = (ctypes.c_size_t * 2)()
    L.check(L.get_lib().dlka_tblock3d_saved_activations_v(B, C, D, H, W, L.DLKA_BF16 if lka_bf16 else L.DLKA_F32, int(variant), offs), "tblock3d_saved_activations")
    n = B * D * H * W * C
    a1 = saved[int(offs[0]):int(offs[0]) + n * 4].view(torch.float32).view(B, D * H * W, C)
    rd = saved[int(offs[1]):int(offs[1]) + n * 4].view(torch.float32).view(B, D * H * W, C)
    return a1 > 0, rd > 0, rd != 0


def lka3d_attention_tokens_backward(x, params, grad_y, saved, dims, variant=0, side_stream=None):
    """side_stream (a torch.cuda.Stream, optional): the pass in two parts (``dlka_lka3d_attention_tokens_backward_phase_v``) — the data-gradient chain on the current
    stream, the five weight-gradient launches and the fold of their partial sums on ``side_stream`` behind an event — and NOT joined: the returned parameter gradients are
    complete only once the current stream has waited for ``side_stream`` (``transformerblock.WgradOverlap`` joins once per backward pass).  Returns (gx, grads, keep):
    ``keep`` = what the side stream still reads, to be held until the join.  "inline": both parts on the current stream (tests)."""
    L.require_device(x, grad_y, saved, *params)
    x, grad_y = x.contiguous(), grad_y.to(x.dtype).contiguous()
    params = [_fp32_param(t) for t in params]
    B, N, C = (int(v) for v in x.shape)
    D, H, W = (int(v) for v in dims)
    lib = L.get_lib()
    dt = L.dtype_code(x)
    wb = lib.dlka_lka3d_tokens_workspace_bytes_v(B, C, D, H, W, dt, int(variant))
    ws = L.scratch(wb, x)
    gx = torch.empty_like(x)
    grads = [torch.empty_like(t) for t in params]
    ps = _ptr_struct(L.Lka3dPtrs, L.LKA3D_FIELDS, params)
    gs = _ptr_struct(L.Lka3dPtrs, L.LKA3D_FIELDS, grads)
    if side_stream is not None:
        pb = lib.dlka_lka3d_tokens_partials_bytes_v(B, C, D, H, W, dt, int(variant))
        part = L.scratch(pb, x)
        nplan = lib.dlka_wgrad_finalize_plan_bytes(1)
        plan = torch.zeros(nplan, dtype=torch.uint8)   # host job table of ONE block (its folds are launched per slot: no device copy is needed)
        planp = ctypes.c_void_p(plan.data_ptr())
        L.check(lib.dlka_wgrad_finalize_plan_init(planp, nplan, 1), "wgrad_finalize_plan_init")
        args = (L.ptr(x), byref(ps), L.ptr(grad_y), L.ptr(saved), saved.numel(), L.ptr(gx), byref(gs), L.ptr(ws), wb, L.ptr(part), pb)
        tail = (B, C, D, H, W, dt, int(variant))
        cur_ptr = L.stream_ptr(x)
        L.check(lib.dlka_lka3d_attention_tokens_backward_phase_v(*args, None, 0, 1, *tail, cur_ptr), "lka3d_attention_tokens_backward (data chain)")
        if isinstance(side_stream, str):
            sp = cur_ptr
        else:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(x.device))
            side_stream.wait_event(ev)
            sp = ctypes.c_void_p(side_stream.cuda_stream)
        L.check(lib.dlka_lka3d_attention_tokens_backward_phase_v(*args, planp, 0, 2, *tail, sp), "lka3d_attention_tokens_backward (weight gradients)")
        L.check(lib.dlka_wgrad_finalize_run_slot(planp, 0, sp), "wgrad_finalize_run_slot")
        return gx, grads, [ws, part, grad_y, saved, x]
    rc = lib.dlka_lka3d_attention_tokens_backward_v(L.ptr(x), byref(ps), L.ptr(grad_y), L.ptr(saved), saved.numel(), L.ptr(gx), byref(gs),
                                                  L.ptr(ws), wb, B, C, D, H, W, dt, int(variant), L.stream_ptr(x))
    L.check(rc, "lka3d_attention_tokens_backward")
    return gx, grads


# ------------------------------------------------------------------------------------------------------------
# TransformerBlock_3D_single_deform_LKA (transformerblock.py:570-630): the wrapper around the D-LKA block
# ------------------------------------------------------------------------------------------------------------
def tblock3d_supported(x, B, C, D, H, W, variant=0) -> bool:
    if x.dtype != torch.float32:
        return False
    return bool(L.get_lib().dlka_tblock3d_supported_v(B, C, D, H, W, L.DLKA_F32, int(variant)))


def tblock3d_lka_bf16_supported(B, C, D, H, W, variant=0) -> bool:
    """The wrapper block's MIXED mode (include/dlka.h: dtype = DLKA_BF16 on dlka_tblock3d_*): fp32 wrapper tensors, the D-LKA attention inside on bf16 activations."""
    return bool(L.get_lib().dlka_tblock3d_supported_v(B, C, D, H, W, L.DLKA_BF16, int(variant)))


def _opt_ptr_struct(cls, fields, tensors):
    st = cls()
    for f, t in zip(fields, tensors):
        setattr(st, f, None if t is None else L.ptr(t))
    return st


def tblock3d_forward(x, x_planar, tparams, lka_params, drop_mask, training, bn_stats, dims, ln_eps=1e-5, bn_eps=1e-5, variant=0, lka_bf16=False):
    """x: [B, C, N...] contiguous NCDHW (x_planar) or [B, N, C] tokens (fp32); dims = the reference's (H, W, D).
    Returns (y tokens [B, N, C], saved).  bn_stats [6*C] is written (training) or read (eval).
    lka_bf16: the MIXED mode — the D-LKA attention inside runs on bf16 activations (DLKA_BF16), the wrapper's own tensors stay fp32."""
    L.require_device(x, bn_stats, drop_mask, *[t for t in tparams if t is not None], *lka_params)
    assert x.is_contiguous() and bn_stats.is_contiguous()
    tparams = [None if t is None else t.contiguous() for t in tparams]
    lka_params = [t.contiguous() for t in lka_params]
    D, H, W = (int(v) for v in dims)
    N = D * H * W
    B = int(x.shape[0])
    C = int(x.shape[1] if x_planar else x.shape[-1])
    assert x.numel() == B * N * C
    lib = L.get_lib()
    if x.dtype != torch.float32:
        raise RuntimeError("tblock3d_forward: the wrapper block takes float32 tensors (lka_bf16 selects bf16 activations for the D-LKA attention inside)")
    dt = L.DLKA_BF16 if lka_bf16 else L.DLKA_F32
    v = int(variant)
    sb, wb = lib.dlka_tblock3d_saved_bytes_v(B, C, D, H, W, dt, v), lib.dlka_tblock3d_workspace_bytes_v(B, C, D, H, W, dt, v)
    if sb == 0:
        L.check(-8, "tblock3d_forward")
    saved, ws = L.scratch(sb, x), L.scratch(wb, x)
    y = torch.empty((B, N, C), dtype=x.dtype, device=x.device)
    ps = _opt_ptr_struct(L.TBlock3dPtrs, L.TBLOCK3D_FIELDS, tparams)
    lk = _ptr_struct(L.Lka3dPtrs, L.LKA3D_FIELDS, lka_params)
    rc = lib.dlka_tblock3d_forward_v(L.ptr(x), int(bool(x_planar)), byref(ps), byref(lk), L.ptr(drop_mask), int(bool(training)), L.ptr(bn_stats),
                                     L.ptr(y), L.ptr(saved), sb, L.ptr(ws), wb, B, C, D, H, W, float(ln_eps), float(bn_eps), dt, v, L.stream_ptr(x))
    L.check(rc, "tblock3d_forward")
    return y, saved


def tblock3d_backward(tparams, lka_params, drop_mask, training, bn_stats, grad_y, saved, dims, variant=0, lka_bf16=False, side_stream=None):
    """Returns (grad_x tokens [B, N, C], grads of tparams (None where the parameter is None), grads of lka_params).

    side_stream (a torch.cuda.Stream, optional): the pass is issued in two parts (``dlka_tblock3d_backward_phase_v``) — the data-gradient chain on the current stream,
    the weight gradients on ``side_stream`` behind an event — and NOT joined: every returned tensor that phase 2 writes is then complete only once the current stream has
    waited for ``side_stream`` (the caller's job: ``transformerblock.WgradOverlap`` joins once per backward pass).  The returned fourth element is the list of tensors
    that must stay alive until that join."""
    L.require_device(grad_y, saved, bn_stats)
    grad_y = grad_y.contiguous()
    tparams = [None if t is None else t.contiguous() for t in tparams]
    lka_params = [t.contiguous() for t in lka_params]
    B, N, C = (int(v) for v in grad_y.shape)
    D, H, W = (int(v) for v in dims)
    lib = L.get_lib()
    dt = L.DLKA_BF16 if lka_bf16 else L.DLKA_F32
    grad_y = grad_y.float()
    wb = lib.dlka_tblock3d_workspace_bytes_v(B, C, D, H, W, dt, int(variant))
    ws = L.scratch(wb, grad_y)
    gx = torch.empty_like(grad_y)
    tg = [None if t is None else torch.empty_like(t) for t in tparams]
    lg = [torch.empty_like(t) for t in lka_params]
    ps = _opt_ptr_struct(L.TBlock3dPtrs, L.TBLOCK3D_FIELDS, tparams)
    lk = _ptr_struct(L.Lka3dPtrs, L.LKA3D_FIELDS, lka_params)
    gs = _opt_ptr_struct(L.TBlock3dPtrs, L.TBLOCK3D_FIELDS, tg)
    gl = _ptr_struct(L.Lka3dPtrs, L.LKA3D_FIELDS, lg)
    args = (byref(ps), byref(lk), L.ptr(drop_mask), int(bool(training)), L.ptr(bn_stats), L.ptr(grad_y), L.ptr(saved),
            saved.numel(), L.ptr(gx), byref(gs), byref(gl), L.ptr(ws), wb, B, C, D, H, W, dt, int(variant))
    if side_stream is None:
        L.check(lib.dlka_tblock3d_backward_v(*args, L.stream_ptr(grad_y)), "tblock3d_backward")
        return gx, tg, lg
    if isinstance(side_stream, str):   # "inline": both parts back to back on the current stream (tests: the split pass equals the whole one)
        L.check(lib.dlka_tblock3d_backward_phase_v(*args, 1, L.stream_ptr(grad_y)), "tblock3d_backward (data chain)")
        L.check(lib.dlka_tblock3d_backward_phase_v(*args, 2, L.stream_ptr(grad_y)), "tblock3d_backward (weight gradients)")
        return gx, tg, lg
    cur = torch.cuda.current_stream(grad_y.device)
    L.check(lib.dlka_tblock3d_backward_phase_v(*args, 1, L.stream_ptr(grad_y)), "tblock3d_backward (data chain)")
    ev = torch.cuda.Event()
    ev.record(cur)
    side_stream.wait_event(ev)
    L.check(lib.dlka_tblock3d_backward_phase_v(*args, 2, ctypes.c_void_p(side_stream.cuda_stream)), "tblock3d_backward (weight gradients)")
    # What phase 2 READS stays referenced until the caller's join: memory released after the join is reused by work that is ordered behind it, so no
    # record_stream() is needed.  What it WRITES — the parameter gradients — must NOT be referenced: autograd's AccumulateGrad takes ownership of a returned
    # gradient only while nobody else holds it, and would otherwise COPY it on the current stream, i.e. before the side stream has written it (seen as garbage
    # gradients, profiles/r08_notes.md); as `.grad` they outlive the join by themselves.
    keep = [ws, grad_y, saved, bn_stats, drop_mask]
    return gx, tg, lg, keep


# ---- the wrapper's pieces, one by one (used by UnetResBlock standalone and by the parity tests) ---------------------------
def layernorm_tokens_forward(x, x_planar, pos, weight, bias, eps=1e-5):
    """x: NCDHW-contiguous [B, C, ...] (x_planar) or tokens [B, N, C].  Returns (xt, xn, stats[M, 2])."""
    L.require_device(x, pos, weight, bias)
    x = x.contiguous()
    B = int(x.shape[0])
    C = int(x.shape[1] if x_planar else x.shape[-1])
    N = x.numel() // (B * C)
    xt = torch.empty((B, N, C), dtype=x.dtype, device=x.device)
    xn = torch.empty_like(xt)
    stats = torch.empty((B * N, 2), dtype=torch.float32, device=x.device)
    rc = L.get_lib().dlka_layernorm_tokens_forward(L.ptr(x), int(bool(x_planar)), L.ptr(None if pos is None else pos.contiguous()), L.ptr(weight.contiguous()),
                                                   L.ptr(bias.contiguous()), L.ptr(xt), L.ptr(xn), L.ptr(stats), B, N, C, float(eps), L.dtype_code(x),
                                                   L.stream_ptr(x))
    L.check(rc, "layernorm_tokens_forward")
    return xt, xn, stats


def layernorm_tokens_backward(g_xn, g_res, xt, stats, weight, with_pos=False):
    L.require_device(g_xn, g_res, xt, stats, weight)
    B, N, C = (int(v) for v in xt.shape)
    g_xn = g_xn.contiguous()
    g_res = None if g_res is None else g_res.contiguous()
    gxt = torch.empty_like(xt)
    gw, gb = torch.empty_like(weight), torch.empty_like(weight)
    gpos = torch.empty((1, N, C), dtype=xt.dtype, device=xt.device) if with_pos else None
    rc = L.get_lib().dlka_layernorm_tokens_backward(L.ptr(g_xn), L.ptr(g_res), L.ptr(xt), L.ptr(stats), L.ptr(weight.contiguous()), L.ptr(gxt), L.ptr(gw),
                                                    L.ptr(gb), L.ptr(gpos), B, N, C, L.dtype_code(xt), L.stream_ptr(xt))
    L.check(rc, "layernorm_tokens_backward")
    return gxt, gw, gb, gpos


def scale_residual_forward(xt, e, gamma):
    L.require_device(xt, e, gamma)
    xt, e = xt.contiguous(), e.contiguous()
    out = torch.empty_like(xt)
    C = int(xt.shape[-1])
    rc = L.get_lib().dlka_scale_residual_forward(L.ptr(xt), L.ptr(e), L.ptr(gamma.contiguous()), L.ptr(out), xt.numel() // C, C, L.dtype_code(xt),
                                                 L.stream_ptr(xt))
    L.check(rc, "scale_residual_forward")
    return out


def scale_residual_backward(g, e, gamma):
    L.require_device(g, e, gamma)
    g, e = g.contiguous(), e.contiguous()
    ge, gg = torch.empty_like(e), torch.empty_like(gamma)
    C = int(e.shape[-1])
    rc = L.get_lib().dlka_scale_residual_backward(L.ptr(g), L.ptr(e), L.ptr(gamma.contiguous()), L.ptr(ge), L.ptr(gg), e.numel() // C, C, L.dtype_code(e),
                                                  L.stream_ptr(e))
    L.check(rc, "scale_residual_backward")
    return ge, gg


def batchnorm_cl_forward(x, res, weight, bias, stats, training, eps=1e-5, slope=0.01):
    """x: channels-last [..., C].  y = LeakyReLU(BN(x) (+ res)).  stats [3*C] written (training) or read (eval: {mean, rstd})."""
    L.require_device(x, res, weight, bias, stats)
    x = x.contiguous()
    res = None if res is None else res.contiguous()
    C = int(x.shape[-1])
    y = torch.empty_like(x)
    scr = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    rc = L.get_lib().dlka_batchnorm_cl_forward(L.ptr(x), L.ptr(res), L.ptr(weight.contiguous()), L.ptr(bias.contiguous()), L.ptr(stats), int(bool(training)),
                                               L.ptr(y), L.ptr(scr), x.numel() // C, C, float(eps), float(slope), L.dtype_code(x), L.stream_ptr(x))
    L.check(rc, "batchnorm_cl_forward")
    return y


def batchnorm_cl_backward(g, x, y, weight, stats, training, with_res=False, slope=0.01):
    L.require_device(g, x, y, weight, stats)
    g, x, y = g.contiguous(), x.contiguous(), y.contiguous()
    C = int(x.shape[-1])
    gx = torch.empty_like(x)
    gres = torch.empty_like(x) if with_res else None
    gw, gb = torch.empty_like(weight), torch.empty_like(weight)
    scr = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    rc = L.get_lib().dlka_batchnorm_cl_backward(L.ptr(g), L.ptr(x), L.ptr(y), L.ptr(weight.contiguous()), L.ptr(stats), int(bool(training)), L.ptr(gx),
                                                L.ptr(gres), L.ptr(gw), L.ptr(gb), L.ptr(scr), x.numel() // C, C, float(slope), L.dtype_code(x),
                                                L.stream_ptr(x))
    L.check(rc, "batchnorm_cl_backward")
    return gx, gres, gw, gb


def channel_scale(x, mask):
    """x [B, ..., C] channels-last, mask [B, C]."""
    L.require_device(x, mask)
    x, mask = x.contiguous(), mask.contiguous()
    B, C = int(x.shape[0]), int(x.shape[-1])
    y = torch.empty_like(x)
    rc = L.get_lib().dlka_channel_scale(L.ptr(x), L.ptr(mask), L.ptr(y), B, x.numel() // (B * C), C, L.dtype_code(x), L.stream_ptr(x))
    L.check(rc, "channel_scale")
    return y


def ncdhw_to_ndhwc(x):
    """[B, C, *S] contiguous -> [B, *S, C] contiguous (HIP transpose)."""
    L.require_device(x)
    x = x.contiguous()
    B, C = int(x.shape[0]), int(x.shape[1])
    out = torch.empty((B, *x.shape[2:], C), dtype=x.dtype, device=x.device)
    L.check(L.get_lib().dlka_ncdhw_to_ndhwc(L.ptr(x), L.ptr(out), B, C, x.numel() // (B * C), L.dtype_code(x), L.stream_ptr(x)), "ncdhw_to_ndhwc")
    return out


def ndhwc_to_ncdhw(x):
    L.require_device(x)
    x = x.contiguous()
    B, C = int(x.shape[0]), int(x.shape[-1])
    out = torch.empty((B, C, *x.shape[1:-1]), dtype=x.dtype, device=x.device)
    L.check(L.get_lib().dlka_ndhwc_to_ncdhw(L.ptr(x), L.ptr(out), B, C, x.numel() // (B * C), L.dtype_code(x), L.stream_ptr(x)), "ndhwc_to_ncdhw")
    return out
