"""Pins the C oracle (oracle/dlka_oracle.c) — the reference has no golden vectors for this path
(SURVEY §4, §8c), so the pins are the known-answer properties its semantics imply plus two independent
restatements (torch gather+autograd, F.grid_sample)."""
import itertools

import pytest
import torch
import torch.nn.functional as F

from oracle import torch_ref

torch.manual_seed(0)


def _mk(B, C, Cout, dims, k, s, p, d, g, dg, dtype=torch.float64, off_scale=1.5, seed=0):
    gen = torch.Generator().manual_seed(seed)
    D, H, W = dims
    k3 = (k,) * 3 if isinstance(k, int) else k
    x = torch.randn(B, C, D, H, W, generator=gen, dtype=dtype)
    w = torch.randn(Cout, C // g, *k3, generator=gen, dtype=dtype) * 0.2
    b = torch.randn(Cout, generator=gen, dtype=dtype)
    o = lambda i, kk, ss, pp, dd: (i + 2 * pp - (dd * (kk - 1) + 1)) // ss + 1
    s3 = (s,) * 3 if isinstance(s, int) else s
    p3 = (p,) * 3 if isinstance(p, int) else p
    d3 = (d,) * 3 if isinstance(d, int) else d
    Do, Ho, Wo = (o(D, k3[0], s3[0], p3[0], d3[0]), o(H, k3[1], s3[1], p3[1], d3[1]), o(W, k3[2], s3[2], p3[2], d3[2]))
    K = k3[0] * k3[1] * k3[2]
    off = torch.randn(B, dg * 3 * K, Do, Ho, Wo, generator=gen, dtype=dtype) * off_scale
    return x, off, w, b


CASES = [
    # B, C, Cout, dims, k, s, p, d, g, dg
    (2, 4, 6, (5, 6, 7), 3, 1, 1, 1, 1, 1),      # the hot-path configuration, small
    (1, 4, 4, (6, 5, 4), 3, 1, 1, 1, 4, 1),      # depthwise (3D/dcn/test.py:28 uses groups=dim)
    (2, 4, 2, (7, 6, 5), (3, 2, 3), (2, 1, 1), (1, 0, 2), (1, 2, 1), 2, 2),  # ragged everything
    (1, 2, 2, (9, 9, 9), 5, 1, 2, 1, 1, 1),      # k=5 p=2 (3D/dcn/test.py:16-22)
    (1, 2, 2, (8, 8, 8), 3, 1, 3, 3, 1, 1),      # dilation 3
]


@pytest.mark.parametrize("case", CASES)
def test_forward_matches_torch_restatement(oracle, case):
    B, C, Cout, dims, k, s, p, d, g, dg = case
    x, off, w, b = _mk(B, C, Cout, dims, k, s, p, d, g, dg)
    y = oracle.deform_conv3d_forward(x, w, b, off, s, p, d, g, dg)
    y2 = torch_ref.deform_conv3d(x, off, w, b, s, p, d, g, dg)
    assert y.shape == y2.shape
    assert torch.allclose(y, y2, atol=1e-11, rtol=1e-11), (y - y2).abs().max()


@pytest.mark.parametrize("case", [c for c in CASES if c[-1] == 1])
def test_forward_matches_grid_sample(oracle, case):
    B, C, Cout, dims, k, s, p, d, g, dg = case
    x, off, w, b = _mk(B, C, Cout, dims, k, s, p, d, g, dg)
    y = oracle.deform_conv3d_forward(x, w, b, off, s, p, d, g, dg)
    y3 = torch_ref.deform_conv3d_grid_sample(x, off, w, b, s, p, d, g, dg)
    assert torch.allclose(y, y3, atol=1e-9, rtol=1e-9), (y - y3).abs().max()


@pytest.mark.parametrize("case", CASES)
def test_backward_matches_autograd_of_restatement(oracle, case):
    B, C, Cout, dims, k, s, p, d, g, dg = case
    x, off, w, b = _mk(B, C, Cout, dims, k, s, p, d, g, dg, seed=3)
    # keep coordinates away from exact integers so that the derivative is two-sided
    xr, offr, wr, br = (t.clone().requires_grad_(True) for t in (x, off, w, b))
    y2 = torch_ref.deform_conv3d(xr, offr, wr, br, s, p, d, g, dg)
    go = torch.randn(y2.shape, generator=torch.Generator().manual_seed(9), dtype=y2.dtype)
    y2.backward(go)
    # the reference's Q1 slip only matters when pad_h != pad_w; compare the consistent variant everywhere
    gi, goff, gw, gb = oracle.deform_conv3d_backward(x, w, b, off, go, s, p, d, g, dg, q1_literal=False)
    for name, a, r in (("gi", gi, xr.grad), ("goff", goff, offr.grad), ("gw", gw, wr.grad), ("gb", gb, br.grad)):
        assert torch.allclose(a, r, atol=1e-9, rtol=1e-9), (name, (a - r).abs().max())


def test_q1_literal_only_differs_when_pad_h_ne_pad_w(oracle):
    """col2im is launched with (pad_d, pad_h, pad_h) — deform_im2col_cuda.cuh:448."""
    x, off, w, b = _mk(1, 2, 2, (5, 6, 7), 3, 1, (1, 1, 1), 1, 1, 1)
    go = torch.randn(oracle.deform_conv3d_forward(x, w, b, off, 1, 1, 1, 1, 1).shape, dtype=x.dtype)
    a = oracle.deform_conv3d_backward(x, w, b, off, go, 1, 1, 1, 1, 1, q1_literal=True)
    c = oracle.deform_conv3d_backward(x, w, b, off, go, 1, 1, 1, 1, 1, q1_literal=False)
    assert all(torch.equal(u, v) for u, v in zip(a, c))
    x, off, w, b = _mk(1, 2, 2, (5, 6, 7), 3, 1, (1, 1, 2), 1, 1, 1)
    go = torch.randn(oracle.deform_conv3d_forward(x, w, b, off, 1, (1, 1, 2), 1, 1, 1).shape, dtype=x.dtype)
    a = oracle.deform_conv3d_backward(x, w, b, off, go, 1, (1, 1, 2), 1, 1, 1, q1_literal=True)
    c = oracle.deform_conv3d_backward(x, w, b, off, go, 1, (1, 1, 2), 1, 1, 1, q1_literal=False)
    assert not torch.equal(a[0], c[0])          # grad_input differs
    assert all(torch.equal(u, v) for u, v in zip(a[1:], c[1:]))


@pytest.mark.parametrize("g", [1, 4])
def test_zero_offset_is_plain_conv(oracle, g):
    """Known answer #1: a freshly built DeformConvPack has zero conv_offset
    (3D/dcn/modules/deform_conv.py:86-88) => output == F.conv3d."""
    x, off, w, b = _mk(2, 4, 8, (6, 7, 5), 3, 1, 1, 1, g, 1, dtype=torch.float32)
    off.zero_()
    y = oracle.deform_conv3d_forward(x, w, b, off, 1, 1, 1, g, 1)
    ref = F.conv3d(x, w, b, 1, 1, 1, g)
    assert torch.allclose(y, ref, atol=2e-5, rtol=1e-5), (y - ref).abs().max()


def test_integer_offset_is_shifted_conv(oracle):
    """Known answer #2: constant integer offsets == conv over a shifted, zero-padded input."""
    x, off, w, b = _mk(1, 3, 2, (6, 6, 6), 3, 1, 1, 1, 1, 1)
    sh = (1, -2, 1)
    off.zero_()
    offv = off.view(1, 27, 3, 6, 6, 6)
    for a in range(3):
        offv[:, :, a] = sh[a]
    y = oracle.deform_conv3d_forward(x, w, b, off, 1, 1, 1, 1, 1)
    pad = 3
    xp = F.pad(x, (pad,) * 6)
    xs = xp[:, :, pad + sh[0]:pad + sh[0] + 6, pad + sh[1]:pad + sh[1] + 6, pad + sh[2]:pad + sh[2] + 6]
    # shifted view then ordinary padding would re-introduce real data at the border: emulate zero padding
    # of the *original* volume by convolving the big padded tensor and cropping.
    big = F.conv3d(xp, w, b, 1, 1)
    ref = big[:, :, pad + sh[0]:pad + sh[0] + 6, pad + sh[1]:pad + sh[1] + 6, pad + sh[2]:pad + sh[2] + 6]
    assert torch.allclose(y, ref, atol=1e-10), (y - ref).abs().max()
    del xs


def test_boundary_guard_and_exact_integers(oracle):
    """coordinates exactly -1, 0, size-1, size and just inside: guard is strict (> -1, < size)."""
    D = 4
    x = torch.arange(1, D * D * D + 1, dtype=torch.float64).view(1, 1, D, D, D)
    w = torch.ones(1, 1, 1, 1, 1, dtype=torch.float64)
    b = torch.zeros(1, dtype=torch.float64)
    for delta, expect_zero in ((-1.0, True), (-0.999, False), (float(D) - 0.001, None), (float(D), True)):
        off = torch.zeros(1, 3, D, D, D, dtype=torch.float64)
        # sample voxel (0,0,0) at coordinate delta along w
        off[0, 2, 0, 0, 0] = delta
        y = oracle.deform_conv3d_forward(x, w, b, off, 1, 0, 1, 1, 1)
        y2 = torch_ref.deform_conv3d(x, off, w, b, 1, 0, 1, 1, 1)
        assert torch.allclose(y, y2, atol=1e-12)
        if expect_zero is True:
            assert y[0, 0, 0, 0, 0] == 0
        if expect_zero is False:
            assert y[0, 0, 0, 0, 0] != 0


def test_sample_index_is_floor_of_int_base_plus_offset(oracle):
    x, off, w, b = _mk(2, 2, 2, (5, 6, 7), 3, 1, 1, 1, 1, 1, dtype=torch.float32, off_scale=4.0)
    idx, mask = oracle.deform_conv3d_sample_index(off, (5, 6, 7), 3, 1, 1, 1, 1)
    K = 27
    offv = off.view(2, 1, K, 3, 5, 6, 7)
    base = torch.stack(torch.meshgrid(torch.arange(5), torch.arange(6), torch.arange(7), indexing="ij"), 0)  # 3,D,H,W
    taps = torch.tensor(list(itertools.product(range(3), repeat=3)))  # K,3
    q = (base.view(1, 1, 1, 3, 5, 6, 7) - 1 + taps.view(1, 1, K, 3, 1, 1, 1)).to(torch.float32) + offv
    assert torch.equal(idx, torch.floor(q).to(torch.int32).permute(0, 1, 2, 4, 5, 6, 3))
    lim = torch.tensor([5, 6, 7]).view(1, 1, 1, 3, 1, 1, 1)
    m = ((q > -1) & (q < lim)).all(3)
    assert torch.equal(mask.bool(), m)


def test_im2col_step_split_equals_single_step(oracle):
    x, off, w, b = _mk(4, 2, 2, (4, 4, 4), 3, 1, 1, 1, 1, 1)
    y1 = oracle.deform_conv3d_forward(x, w, b, off, 1, 1, 1, 1, 1, im2col_step=64)
    y2 = oracle.deform_conv3d_forward(x, w, b, off, 1, 1, 1, 1, 1, im2col_step=2)
    assert torch.equal(y1, y2)
    go = torch.randn_like(y1)
    g1 = oracle.deform_conv3d_backward(x, w, b, off, go, 1, 1, 1, 1, 1, im2col_step=64)
    g2 = oracle.deform_conv3d_backward(x, w, b, off, go, 1, 1, 1, 1, 1, im2col_step=2)
    for a, c in zip(g1, g2):
        assert torch.allclose(a, c, atol=1e-12)
    with pytest.raises(RuntimeError):
        oracle.deform_conv3d_forward(x, w, b, off, 1, 1, 1, 1, 1, im2col_step=3)  # 4 % 3 != 0 (deform_conv_cuda.cu:61)


def test_autograd_function_wrapper(oracle):
    x, off, w, b = _mk(1, 2, 2, (4, 4, 4), 3, 1, 1, 1, 1, 1)
    xs = [t.clone().requires_grad_(True) for t in (x, off, w, b)]
    y = oracle.DeformConv3dFunction.apply(xs[0], xs[1], xs[2], xs[3], 1, 1, 1, 1, 1, 64)
    y.sum().backward()
    assert all(t.grad is not None for t in xs)
    assert torch.allclose(xs[3].grad, torch.full((2,), 64.0, dtype=torch.float64))
