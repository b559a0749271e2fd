"""TEST INFRASTRUCTURE.  DLKA_F64 checks shared by tests/test_f64_gpu.py (MI355X) and tests/test_parity_emu.py (the same kernel sources on the host emulator)."""
import pytest
import torch


def _offsets(shape, gen):
    """Offsets whose sampling coordinates stay away from integers (gradcheck differentiates numerically: the interpolant has kinks at integer coordinates)."""
    off = (torch.rand(shape, generator=gen, dtype=torch.float64) * 0.6 + 0.2)   # fractional part in [0.2, 0.8]
    return off + torch.randint(-1, 2, shape, generator=gen).double()


def gradcheck_deform_conv3d(dev):
    """3D/dcn/test.py:16-22 (DeformConv(32, 32, k, padding) on a (2, 32, 32, 32, 32) volume; gradcheck is imported at :9) shrunk to what numerical differentiation can
    afford: every input of DeformConvFunction.apply, incl. groups = 2 and deformable_groups = 2."""
    from deformablelka_amd.functions.deform_conv_func import DeformConvFunction
    gen = torch.Generator().manual_seed(0)
    for (B, C, Co, dims, k, s, p, d, g, dg) in [(1, 2, 2, (2, 2, 3), 3, 1, 1, 1, 1, 1), (1, 4, 4, (3, 3, 3), (3, 2, 3), (1, 1, 2), (1, 0, 1), (1, 2, 1), 2, 2)]:
        k3 = (k,) * 3 if isinstance(k, int) else k
        s3 = (s,) * 3 if isinstance(s, int) else s
        p3 = (p,) * 3 if isinstance(p, int) else p
        d3 = (d,) * 3 if isinstance(d, int) else d
        od = [(n + 2 * pp - dd * (kk - 1) - 1) // ss + 1 for n, kk, ss, pp, dd in zip(dims, k3, s3, p3, d3)]
        K = k3[0] * k3[1] * k3[2]
        x = torch.randn(B, C, *dims, generator=gen, dtype=torch.float64).to(dev).requires_grad_(True)
        off = _offsets((B, dg * 3 * K, *od), gen).to(dev).requires_grad_(True)
        w = (torch.randn(Co, C // g, *k3, generator=gen, dtype=torch.float64) * 0.3).to(dev).requires_grad_(True)
        b = torch.randn(Co, generator=gen, dtype=torch.float64).to(dev).requires_grad_(True)
        fn = lambda x_, o_, w_, b_: DeformConvFunction.apply(x_, o_, w_, b_, s3, p3, d3, g, dg, 64)
        assert torch.autograd.gradcheck(fn, (x, off, w, b), eps=1e-6, atol=1e-7, rtol=1e-6, nondet_tol=1e-12)


def gradcheck_deform_conv2d(dev):
    from deformablelka_amd import tv_ops
    gen = torch.Generator().manual_seed(1)
    B, C, H, W, k = 1, 2, 3, 4, 3
    x = torch.randn(B, C, H, W, generator=gen, dtype=torch.float64).to(dev).requires_grad_(True)
    off = _offsets((B, 2 * k * k, H, W), gen).to(dev).requires_grad_(True)
    w = (torch.randn(C, 1, k, k, generator=gen, dtype=torch.float64) * 0.3).to(dev).requires_grad_(True)   # depthwise, as the 2-D D-LKA block uses it
    fn = lambda x_, o_, w_: tv_ops.deform_conv2d(x_, o_, w_, None, stride=1, padding=1, dilation=1)
    assert torch.autograd.gradcheck(fn, (x, off, w), eps=1e-6, atol=1e-7, rtol=1e-6, nondet_tol=1e-12)


def gradcheck_conv3d(dev):
    from deformablelka_amd import nn_ops
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(1, 4, 2, 4, 5, generator=gen, dtype=torch.float64).to(dev).requires_grad_(True)
    w = (torch.randn(6, 2, 3, 3, 3, generator=gen, dtype=torch.float64) * 0.3).to(dev).requires_grad_(True)
    b = torch.randn(6, generator=gen, dtype=torch.float64).to(dev).requires_grad_(True)
    fn = lambda x_, w_, b_: nn_ops.conv3d(x_, w_, b_, (1, 2, 1), (1, 1, 0), (1, 1, 2), 2)
    assert torch.autograd.gradcheck(fn, (x, w, b), eps=1e-6, atol=1e-7, rtol=1e-6)
    ref = torch.nn.functional.conv3d(x.detach().cpu(), w.detach().cpu(), b.detach().cpu(), (1, 2, 1), (1, 1, 0), (1, 1, 2), 2)
    assert float((fn(x, w, b).detach().cpu() - ref).abs().max()) <= 1e-12 * float(ref.abs().max())


def fast_paths_refuse_double(dev):
    """float64 is a dtype of the GENERAL operators only (include/dlka.h, dlka_dtype): the fused token block raises instead of misreading the buffers."""
    import deformablelka_amd as dk
    m = dk.LKA_Attention3d_deform(32).to(dev).double()
    x = torch.randn(1, 8, 32, dtype=torch.float64).to(dev)
    with pytest.raises((RuntimeError, NotImplementedError)):
        m(x, 1, 32, 2, 2, 2)
