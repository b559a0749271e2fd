"""-m gpu: DLKA_F64 on the general NCDHW operators — the second type of the reference's dispatch (AT_DISPATCH_FLOATING_TYPES: float, double,
3D/dcn/src/cuda/deform_conv_cuda.cu:96,233).
  * the reference's OWN op compiled for double (oracle/_ref/D3D.so, built unmodified from /root/reference) against the product in double: forward and all four
    gradients to 1e-10 of max|ref| (cases incl. groups, deformable groups, stride, dilation, integer / out-of-volume offsets; the Q1 case excluded: there the
    reference's grad_input uses pad_h for pad_w, cuh:447, a documented deviation);
  * torch.autograd.gradcheck THROUGH the product (DeformConvFunction.apply at the reference's smoke configuration 3D/dcn/test.py:16-22 shrunk; the 2-D operator;
    the plain conv3d) — a check that does not depend on the builder's oracle."""
import pytest
import torch

from tests import f64_checks, ref_cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def backends():
    from deformablelka_amd import _lib
    _lib._set_backend_for_tests(None)
    assert torch.cuda.is_available()
    _lib.get_lib()
    yield


@pytest.mark.parametrize("name", [k for k in ref_cases.SMALL if k != "q1_pad_h_ne_pad_w"] + ["params_py_k3", "stage2", "stage3"])
def test_product_double_equals_the_references_own_op_in_double(name):
    from oracle import ref
    if not ref.available():
        pytest.fail("oracle/_ref/D3D.so is missing — run __graft_entry__.build() where /root/reference is mounted")
    from deformablelka_amd import ops
    case = ref_cases.SMALL.get(name) or ref_cases.FULL[name]
    t = ref_cases.make(case)
    x, w, b, off, go = (t[k].double().to(DEV).contiguous() for k in ("x", "w", "b", "off", "go"))
    k3 = tuple(w.shape[2:5])
    r_out = ref.deform_conv3d_forward(x, w, b, off, t["s"], t["p"], t["d"], t["g"], t["dg"], t["step"])
    r_g = ref.deform_conv3d_backward(x, w, b, off, go, t["s"], t["p"], t["d"], t["g"], t["dg"], t["step"])
    out = ops.deform_conv3d_forward(x, w, b, off, k3, t["s"], t["p"], t["d"], t["g"], t["dg"], t["step"])
    g = ops.deform_conv3d_backward(x, w, b, off, go, k3, t["s"], t["p"], t["d"], t["g"], t["dg"], t["step"])
    assert out.dtype == torch.float64 and all(q.dtype == torch.float64 for q in g)
    for nm, a_, r_ in zip(("output", "grad_input", "grad_offset", "grad_weight", "grad_bias"), [out, *g], [r_out, *r_g]):
        if nm == "grad_input" and t["p"][1] != t["p"][2]:
            continue   # Q1: the reference's col2im forwards pad_h in place of pad_w (cuh:447); the product computes the consistent gradient (tests/test_ref_d3d_gpu.py)
        err = float((a_ - r_).abs().max() / r_.abs().max().clamp_min(1e-30))
        assert err <= 1e-10, f"{name} {nm}: {err:.3e} of max|ref| (double)"


def test_gradcheck_through_the_product_deform_conv3d():
    f64_checks.gradcheck_deform_conv3d(DEV)


def test_gradcheck_through_the_product_deform_conv2d():
    f64_checks.gradcheck_deform_conv2d(DEV)


def test_gradcheck_through_the_product_conv3d():
    f64_checks.gradcheck_conv3d(DEV)


def test_double_is_refused_by_the_fast_paths():
    f64_checks.fast_paths_refuse_double(DEV)
