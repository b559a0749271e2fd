"""-m gpu: pins parity to the REFERENCE'S OWN ARITHMETIC.  oracle/_ref/D3D.so is the reference's pybind11 module `D3D`
(3D/dcn/src/vision.cpp, cuda/deform_conv_cuda.cu, cuda/deform_im2col_cuda.cuh) compiled unmodified by oracle/ref.mk; here it runs
on the MI355X next to (1) the C oracle and (2) the HIP kernels — general NCDHW path and channels-last fast path — on the
reference's smoke-script shapes and on BASELINE.json's stage shapes incl. the headline (C=32, 32^3, B=2) with ~1-voxel offsets.

Tolerances: forward <= 1e-4 abs (north_star); gradients <= 1e-3 rel of max|ref| (the reference's own col2im uses fp32
atomicAdd, cuh:326-328, so its grad_input is order-dependent at the 1e-6 level)."""
import pytest
import torch

from tests import parity, ref_cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def backends(oracle):
    from deformablelka_amd import _lib
    from oracle import ref
    _lib._set_backend_for_tests(None)
    assert torch.cuda.is_available()
    _lib.get_lib()
    if not ref.available():
        pytest.fail("oracle/_ref/D3D.so is missing — run __graft_entry__.build() where /root/reference is mounted; "
                    "the prebuilt file travels to the GPU box")
    ref.D3D()
    yield


def _oracle_all(t, q1_literal):
    import oracle
    out = oracle.deform_conv3d_forward(t["x"], t["w"], t["b"], t["off"], t["s"], t["p"], t["d"], t["g"], t["dg"], t["step"])
    g = oracle.deform_conv3d_backward(t["x"], t["w"], t["b"], t["off"], t["go"], t["s"], t["p"], t["d"], t["g"], t["dg"], t["step"],
                                      q1_literal=q1_literal)
    return [out, *g]


def _hip_all(t):
    from deformablelka_amd import ops
    x, w, b, off, go = (t[k].to(DEV) for k in ("x", "w", "b", "off", "go"))
    k = tuple(w.shape[2:5])
    out = ops.deform_conv3d_forward(x, w, b, off, k, t["s"], t["p"], t["d"], t["g"], t["dg"], t["step"])
    g = ops.deform_conv3d_backward(x, w, b, off, go, k, t["s"], t["p"], t["d"], t["g"], t["dg"], t["step"])
    return [out, *g]


def _hip_cl_all(t):
    from deformablelka_amd import ops
    x, w, b, off, go = (t[k].to(DEV) for k in ("x", "w", "b", "off", "go"))
    out = ops.deform_conv3d_forward_cl(parity.to_cl(x), off, w, b, 1, 1)
    gi, goff, gw, gb = ops.deform_conv3d_backward_cl(parity.to_cl(x), off, w, parity.to_cl(go), 1, 1)
    return [parity.from_cl(out), parity.from_cl(gi), goff, gw, gb]


NAMES = ("output", "grad_input", "grad_offset", "grad_weight", "grad_bias")


def _compare(tag, got, ref, fwd_atol=parity.FWD_ATOL, rtol=parity.BWD_RTOL):
    parity.assert_close(f"{tag} {NAMES[0]}", got[0], ref[0], atol=fwd_atol)
    for n, a, r in zip(NAMES[1:], got[1:], ref[1:]):
        parity.assert_close(f"{tag} {n}", a, r, rtol=rtol)


def _is_fast_path(case):
    B, C, Cout, dims, k, s, p, d, g, dg, *_ = case
    return (k, s, p, d, g, dg) == (3, 1, 1, 1, 1, 1) and C == Cout and C % 32 == 0


@pytest.mark.parametrize("name", list(ref_cases.SMALL) + list(ref_cases.FULL))
def test_reference_native_op_vs_oracle_and_hip(name):
    case = {**ref_cases.SMALL, **ref_cases.FULL}[name]
    t = ref_cases.make(case)
    ref = ref_cases.run_ref(t, DEV)
    q1 = t["p"][1] != t["p"][2]   # Q1 (cuh:447) shows wherever pad_h != pad_w
    # (1) the C oracle restates the reference: literal Q1 variant is what the reference computes
    _compare("oracle vs reference", _oracle_all(t, q1_literal=True), ref)
    # (2) the HIP kernels vs the reference.  For pad_h != pad_w the product uses pad_w consistently (the reference's
    #     grad_input is shifted by its own slip there), so that one case checks everything except grad_input.
    hip = _hip_all(t)
    if q1:
        hip[1] = ref[1]
    _compare("hip general vs reference", hip, ref)
    if _is_fast_path(case):
        _compare("hip channels-last vs reference", _hip_cl_all(t), ref)


def test_reference_rejects_what_we_reject():
    """Error behaviour at the boundary (deform_conv_cuda.cu:41-76): non-contiguous input, wrong kernel dims, bad im2col_step."""
    from deformablelka_amd import D3D as ours
    from oracle import ref
    theirs = ref.D3D()
    t = ref_cases.make(ref_cases.SMALL["k3_normal"])
    x, w, b, off = (t[k].to(DEV) for k in ("x", "w", "b", "off"))
    args = (3, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 64)
    for mod in (theirs, ours):
        with pytest.raises(RuntimeError):
            mod.deform_conv_forward(x.transpose(3, 4), w, b, off, *args)                       # cu:41
        with pytest.raises(RuntimeError):
            mod.deform_conv_forward(x, w, b, off, 5, 3, 3, *args[3:])                          # cu:72-73
        with pytest.raises(RuntimeError):
            mod.deform_conv_forward(x.repeat(3, 1, 1, 1, 1)[:3], w, b, off.repeat(3, 1, 1, 1, 1)[:3], *args[:-1], 2)   # cu:61-63: 3 % 2
        with pytest.raises(RuntimeError):
            mod.deform_conv_forward(x.cpu(), w.cpu(), b.cpu(), off.cpu(), *args)               # deform_conv.h:46 / cu:44
