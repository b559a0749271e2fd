"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

CPU oracle for the D-LKA hot path: ctypes bindings over ``oracle/dlka_oracle.c`` (a plain-C
restatement of the reference kernels, see ``dlka_oracle_impl.h`` for file:line citations) plus a few
torch-level compositions that restate the reference's Python modules.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  ``deformablelka_amd`` never does (tests/test_no_oracle_in_product.py enforces it).

Parity status (3-D, D3D): **pinned to the reference's own arithmetic**.  The reference has no golden vectors for this
path, but its native op builds for gfx950 (``oracle/ref.mk`` -> ``oracle/_ref/D3D.so``, sources unmodified): the C oracle is
checked against its recorded outputs on CPU (tests/golden/d3d_reference_vectors.pt, tests/test_oracle_vs_reference_vectors.py)
and against the op itself on the MI355X (tests/test_ref_d3d_gpu.py), besides derived known-answer properties and the golden
vectors generated through the reference's Python modules (tests/golden/make_golden.py).
Parity status (2-D, torchvision 0.12 deform_conv2d): **pinned by the reference's own 3-D op**.  torchvision is neither vendored by the
reference nor installed here, but the D3D kernels compute exactly the 2-D operator on a depth-1 embedding (depth axis of size 1, kd = 1, pad_d = 0,
zero depth offsets: qd = 0 -> floor 0, ld = 0, upper-depth corner dropped; deform_im2col_cuda.cuh:26-72,245-259).  ``oracle/_ref/D3D.so`` run on
that embedding pins the 2-D restatement on the MI355X (tests/test_ref_d3d_2d_gpu.py) and, through recorded vectors
(tests/golden/d3d_reference_vectors_2d.pt), in the CPU suite (tests/test_oracle_2d_pinned.py).  One line stays restated from torchvision's
published kernel: its coordinate weight is unguarded, which differs from D3D's at q == -1 exactly (isolated in its own test).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Sequence, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdlka_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (idempotent)."""
    src = [os.path.join(_HERE, f) for f in ("dlka_oracle.c", "dlka_oracle_impl.h", "Makefile")]
    def stale():
        return not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)

    if force or stale():
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lk:   # several test processes (pytest-xdist workers, spawned ranks) may get here at once
            fcntl.flock(lk, fcntl.LOCK_EX)
            if force or stale():
                subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _triple(v) -> Tuple[int, int, int]:
    if isinstance(v, int):
        return (v, v, v)
    v = tuple(int(x) for x in v)
    assert len(v) == 3
    return v


def _pair(v) -> Tuple[int, int]:
    if isinstance(v, int):
        return (v, v)
    v = tuple(int(x) for x in v)
    assert len(v) == 2
    return v


def _sfx(t: torch.Tensor) -> str:
    if t.dtype == torch.float32:
        return "_f32"
    if t.dtype == torch.float64:
        return "_f64"
    raise TypeError(f"oracle supports float32/float64 only (reference: AT_DISPATCH_FLOATING_TYPES), got {t.dtype}")


class _Args:
    """Collects contiguous CPU views of the arguments and keeps them alive across the C call
    (a bare ``t.contiguous().data_ptr()`` would dangle as soon as the temporary is collected)."""

    def __init__(self):
        self.keep = []

    def __call__(self, t):
        if t is None:
            return ctypes.c_void_p(0)
        assert t.device.type == "cpu", "oracle wants CPU tensors"
        t = t.detach().contiguous()
        self.keep.append(t)
        return ctypes.c_void_p(t.data_ptr())


def _out_size(i, p, d, k, s):
    return (i + 2 * p - (d * (k - 1) + 1)) // s + 1


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with code {rc}")


# --------------------------------------------------------------------------------------------
# 3-D deformable conv (D3D semantics)
# --------------------------------------------------------------------------------------------
def deform_conv3d_forward(input, weight, bias, offset, stride=1, padding=0, dilation=1, group=1,
                          deformable_groups=1, im2col_step=64):
    """Restates ``deform_conv_cuda_forward`` (3D/dcn/src/cuda/deform_conv_cuda.cu:18-126)."""
    s, p, d = _triple(stride), _triple(padding), _triple(dilation)
    B, C, D, H, W = input.shape
    Cout, _, kd, kh, kw = weight.shape
    Do, Ho, Wo = _out_size(D, p[0], d[0], kd, s[0]), _out_size(H, p[1], d[1], kh, s[1]), _out_size(W, p[2], d[2], kw, s[2])
    out = torch.empty((B, Cout, Do, Ho, Wo), dtype=input.dtype)
    _p = _Args()
    fn = getattr(lib(), "dlka_oracle_deform_conv3d_forward" + _sfx(input))
    rc = fn(_p(input), _p(weight), _p(bias), _p(offset), _p(out),
            B, C, D, H, W, Cout, kd, kh, kw, *s, *p, *d, group, deformable_groups, im2col_step)
    _check(rc, "deform_conv3d_forward")
    return out


def deform_conv3d_backward(input, weight, bias, offset, grad_output, stride=1, padding=0, dilation=1, group=1,
                           deformable_groups=1, im2col_step=64, q1_literal=True):
    """Restates ``deform_conv_cuda_backward`` (deform_conv_cuda.cu:128-285)."""
    s, p, d = _triple(stride), _triple(padding), _triple(dilation)
    B, C, D, H, W = input.shape
    Cout, _, kd, kh, kw = weight.shape
    gi, go = torch.empty_like(input, memory_format=torch.contiguous_format), torch.empty_like(offset, memory_format=torch.contiguous_format)
    gw, gb = torch.empty_like(weight, memory_format=torch.contiguous_format), torch.empty_like(bias, memory_format=torch.contiguous_format)
    _p = _Args()
    fn = getattr(lib(), "dlka_oracle_deform_conv3d_backward" + _sfx(input))
    rc = fn(_p(input), _p(weight), _p(bias), _p(offset),
            _p(grad_output), _p(gi), _p(go), _p(gw), _p(gb),
            B, C, D, H, W, Cout, kd, kh, kw, *s, *p, *d, group, deformable_groups, im2col_step, int(bool(q1_literal)))
    _check(rc, "deform_conv3d_backward")
    return gi, go, gw, gb


def deform_conv3d_sample_index(offset, in_size: Sequence[int], kernel_size, stride=1, padding=0, dilation=1,
                               deformable_groups=1):
    """floor() indices and in-range mask per (b, dg, tap, out voxel) — the bit-exact index oracle."""
    s, p, d, k = _triple(stride), _triple(padding), _triple(dilation), _triple(kernel_size)
    D, H, W = in_size
    B = offset.shape[0]
    Do, Ho, Wo = offset.shape[2:]
    K = k[0] * k[1] * k[2]
    idx = torch.empty((B, deformable_groups, K, Do, Ho, Wo, 3), dtype=torch.int32)
    mask = torch.empty((B, deformable_groups, K, Do, Ho, Wo), dtype=torch.uint8)
    _p = _Args()
    fn = getattr(lib(), "dlka_oracle_deform_conv3d_sample_index" + _sfx(offset))
    rc = fn(_p(offset), _p(idx), _p(mask), B, D, H, W, *k, *s, *p, *d, deformable_groups)
    _check(rc, "deform_conv3d_sample_index")
    return idx, mask


class DeformConv3dFunction(torch.autograd.Function):
    """Oracle twin of the reference's ``DeformConvFunction`` (3D/dcn/functions/deform_conv_func.py:15-56)."""

    @staticmethod
    def forward(ctx, input, offset, weight, bias, stride, padding, dilation, group, deformable_groups, im2col_step):
        ctx.cfg = (_triple(stride), _triple(padding), _triple(dilation), group, deformable_groups, im2col_step)
        ctx.save_for_backward(input, offset, weight, bias)
        return deform_conv3d_forward(input.detach(), weight.detach(), bias.detach(), offset.detach(), *ctx.cfg)

    @staticmethod
    def backward(ctx, grad_output):
        input, offset, weight, bias = ctx.saved_tensors
        gi, go, gw, gb = deform_conv3d_backward(input.detach(), weight.detach(), bias.detach(), offset.detach(),
                                                grad_output.contiguous(), *ctx.cfg)
        return gi, go, gw, gb, None, None, None, None, None, None


# --------------------------------------------------------------------------------------------
# 2-D deformable conv (torchvision 0.12 semantics, mask=None)
# --------------------------------------------------------------------------------------------
def deform_conv2d_forward(input, offset, weight, bias=None, stride=1, padding=0, dilation=1):
    s, p, d = _pair(stride), _pair(padding), _pair(dilation)
    B, C, H, W = input.shape
    Cout, Cg, kh, kw = weight.shape
    group = C // Cg
    og = offset.shape[1] // (2 * kh * kw)
    Ho, Wo = _out_size(H, p[0], d[0], kh, s[0]), _out_size(W, p[1], d[1], kw, s[1])
    out = torch.empty((B, Cout, Ho, Wo), dtype=input.dtype)
    _p = _Args()
    fn = getattr(lib(), "dlka_oracle_deform_conv2d_forward" + _sfx(input))
    rc = fn(_p(input), _p(weight), _p(bias),
            _p(offset), _p(out), B, C, H, W, Cout, kh, kw, *s, *p, *d, group, og)
    _check(rc, "deform_conv2d_forward")
    return out


def deform_conv2d_backward(input, offset, weight, grad_output, stride=1, padding=0, dilation=1, with_bias=False):
    s, p, d = _pair(stride), _pair(padding), _pair(dilation)
    B, C, H, W = input.shape
    Cout, Cg, kh, kw = weight.shape
    group = C // Cg
    og = offset.shape[1] // (2 * kh * kw)
    gi, go, gw = torch.empty_like(input, memory_format=torch.contiguous_format), torch.empty_like(offset, memory_format=torch.contiguous_format), torch.empty_like(weight, memory_format=torch.contiguous_format)
    gb = torch.empty((Cout,), dtype=input.dtype) if with_bias else None
    _p = _Args()
    fn = getattr(lib(), "dlka_oracle_deform_conv2d_backward" + _sfx(input))
    rc = fn(_p(input), _p(weight), _p(offset), _p(grad_output),
            _p(gi), _p(go), _p(gw), _p(gb), B, C, H, W, Cout, kh, kw, *s, *p, *d, group, og)
    _check(rc, "deform_conv2d_backward")
    return gi, go, gw, gb


class DeformConv2dFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, offset, weight, bias, stride, padding, dilation):
        ctx.cfg = (_pair(stride), _pair(padding), _pair(dilation))
        ctx.has_bias = bias is not None
        ctx.save_for_backward(input, offset, weight)
        return deform_conv2d_forward(input.detach(), offset.detach(), weight.detach(),
                                     None if bias is None else bias.detach(), *ctx.cfg)

    @staticmethod
    def backward(ctx, grad_output):
        input, offset, weight = ctx.saved_tensors
        gi, go, gw, gb = deform_conv2d_backward(input.detach(), offset.detach(), weight.detach(),
                                                grad_output.contiguous(), *ctx.cfg, with_bias=ctx.has_bias)
        return gi, go, gw, gb, None, None, None


# --------------------------------------------------------------------------------------------
# plain conv (naive C loops) — cross-check for F.conv3d / the HIP dw + dense conv kernels
# --------------------------------------------------------------------------------------------
def conv3d_forward(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    s, p, d = _triple(stride), _triple(padding), _triple(dilation)
    B, C, D, H, W = input.shape
    Cout, _, kd, kh, kw = weight.shape
    Do, Ho, Wo = _out_size(D, p[0], d[0], kd, s[0]), _out_size(H, p[1], d[1], kh, s[1]), _out_size(W, p[2], d[2], kw, s[2])
    out = torch.empty((B, Cout, Do, Ho, Wo), dtype=input.dtype)
    _p = _Args()
    fn = getattr(lib(), "dlka_oracle_conv3d_forward" + _sfx(input))
    rc = fn(_p(input), _p(weight), _p(bias), _p(out),
            B, C, D, H, W, Cout, kd, kh, kw, *s, *p, *d, groups)
    _check(rc, "conv3d_forward")
    return out


def conv3d_backward(input, weight, grad_output, stride=1, padding=0, dilation=1, groups=1, with_bias=True):
    s, p, d = _triple(stride), _triple(padding), _triple(dilation)
    B, C, D, H, W = input.shape
    Cout, _, kd, kh, kw = weight.shape
    gi, gw = torch.empty_like(input, memory_format=torch.contiguous_format), torch.empty_like(weight, memory_format=torch.contiguous_format)
    gb = torch.empty((Cout,), dtype=input.dtype) if with_bias else None
    _p = _Args()
    fn = getattr(lib(), "dlka_oracle_conv3d_backward" + _sfx(input))
    rc = fn(_p(input), _p(weight), _p(grad_output), _p(gi), _p(gw), _p(gb),
            B, C, D, H, W, Cout, kd, kh, kw, *s, *p, *d, groups)
    _check(rc, "conv3d_backward")
    return gi, gw, gb
