"""Pins the 2-D deformable oracle (torchvision-0.12 semantics restated) and the naive conv oracle."""
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_ref


def _mk2(B, C, Cout, H, W, k, s, p, d, g, og, dtype=torch.float64, seed=0, scale=1.5):
    gen = torch.Generator().manual_seed(seed)
    kh, kw = k
    x = torch.randn(B, C, H, W, generator=gen, dtype=dtype)
    w = torch.randn(Cout, C // g, kh, kw, generator=gen, dtype=dtype) * 0.3
    o = lambda i, kk, ss, pp, dd: (i + 2 * pp - (dd * (kk - 1) + 1)) // ss + 1
    Ho, Wo = o(H, kh, s, p, d), o(W, kw, s, p, d)
    off = torch.randn(B, og * 2 * kh * kw, Ho, Wo, generator=gen, dtype=dtype) * scale
    return x, off, w


CASES2 = [
    (2, 6, 6, 9, 8, (5, 5), 1, 2, 1, 6, 1),    # D-LKA conv0: depthwise 5x5 pad 2 (2D/deformable_LKA/deformable_LKA.py:93)
    (1, 4, 4, 12, 11, (7, 7), 1, 9, 3, 4, 1),  # D-LKA conv_spatial: depthwise 7x7 dil 3 pad 9 (:94)
    (2, 4, 6, 7, 9, (3, 3), 2, 1, 1, 2, 2),    # grouped, strided, 2 offset groups
    (1, 3, 5, 6, 6, (3, 3), 1, 1, 1, 1, 1),    # dense
]


@pytest.mark.parametrize("case", CASES2)
def test_deform2d_forward_backward(oracle, case):
    B, C, Cout, H, W, k, s, p, d, g, og = case
    x, off, w = _mk2(B, C, Cout, H, W, k, s, p, d, g, og)
    bias = torch.randn(Cout, dtype=x.dtype)
    y = oracle.deform_conv2d_forward(x, off, w, bias, s, p, d)
    xr, offr, wr, br = (t.clone().requires_grad_(True) for t in (x, off, w, bias))
    y2 = torch_ref.deform_conv2d(xr, offr, wr, br, s, p, d)
    assert torch.allclose(y, y2, atol=1e-11), (y - y2).abs().max()
    go = torch.randn_like(y2)
    y2.backward(go)
    gi, goff, gw, gb = oracle.deform_conv2d_backward(x, off, w, go, s, p, d, with_bias=True)
    for name, a, r in (("gi", gi, xr.grad), ("goff", goff, offr.grad), ("gw", gw, wr.grad), ("gb", gb, br.grad)):
        assert torch.allclose(a, r, atol=1e-9, rtol=1e-9), (name, (a - r).abs().max())


def test_deform2d_zero_offset_is_conv2d(oracle):
    x, off, w = _mk2(2, 8, 8, 10, 10, (7, 7), 1, 9, 3, 8, 1, dtype=torch.float32)
    off.zero_()
    y = oracle.deform_conv2d_forward(x, off, w, None, 1, 9, 3)
    ref = F.conv2d(x, w, None, 1, 9, 3, 8)
    assert torch.allclose(y, ref, atol=2e-5), (y - ref).abs().max()


CONV = [
    (2, 4, 4, (6, 7, 8), 5, 1, 2, 1, 4),     # dw 5^3 pad 2         (synapse/transformerblock.py:637)
    (1, 3, 3, (10, 9, 11), 7, 1, 9, 3, 3),   # dw 7^3 dil 3 pad 9   (:638)
    (2, 4, 9, (5, 6, 4), 3, 1, 1, 1, 1),     # dense 3^3 (conv_offset, deform_conv.py:80-85)
    (1, 4, 6, (5, 5, 5), 1, 1, 0, 1, 1),     # pointwise
    (1, 4, 4, (7, 8, 6), (3, 5, 5), 1, (1, 6, 6), (1, 3, 3), 4),  # ACDC anisotropic dw (acdc/transformerblock.py:213-231)
    (1, 4, 6, (7, 6, 5), 3, 2, 1, 1, 2),     # strided grouped
]


@pytest.mark.parametrize("case", CONV)
def test_conv3d_oracle_matches_aten(oracle, case):
    B, C, Cout, dims, k, s, p, d, g = case
    k3 = (k,) * 3 if isinstance(k, int) else k
    x = torch.randn(B, C, *dims, dtype=torch.float64)
    w = torch.randn(Cout, C // g, *k3, dtype=torch.float64)
    b = torch.randn(Cout, dtype=torch.float64)
    y = oracle.conv3d_forward(x, w, b, s, p, d, g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = F.conv3d(xr, wr, br, s, p, d, g)
    assert torch.allclose(y, ref, atol=1e-10), (y - ref).abs().max()
    go = torch.randn_like(ref)
    ref.backward(go)
    gi, gw, gb = oracle.conv3d_backward(x, w, go, s, p, d, g)
    assert torch.allclose(gi, xr.grad, atol=1e-10)
    assert torch.allclose(gw, wr.grad, atol=1e-9)
    assert torch.allclose(gb, br.grad, atol=1e-9)
