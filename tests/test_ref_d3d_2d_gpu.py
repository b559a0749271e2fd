"""-m gpu: pins the 2-D parity chain to the REFERENCE'S OWN ARITHMETIC.

torchvision (the reference's 2-D deformable op, `torchvision==0.12.0`, 2D/requirements.txt:69) is neither vendored nor installed, but the
reference's own 3-D op computes exactly the 2-D operator when called with a depth axis of size 1, kd = 1, pad_d = 0 and zero depth offsets
(tests/ref_cases.py: qd = 0 -> floor 0, ld = 0, upper-depth corner dropped; deform_im2col_cuda.cuh:26-72,245-259).  oracle/_ref/D3D.so — the
reference's 3D/dcn/src compiled unmodified — therefore produces reference-arithmetic vectors for the depthwise 5x5 / 7x7-dilation-3
convs of the 2-D D-LKA block, and here they hold, on the MI355X:
  (1) the 2-D C oracle (restatement of torchvision's kernel, oracle/dlka_oracle_impl.h),
  (2) the general 2-D HIP kernels (deform_conv.hip),
  (3) the channels-last depthwise fast path (cl_ddw2d.hip) the 2-D block actually runs.
The only torchvision line that stays restated is its UNGUARDED coordinate weight, which differs from D3D's at q == -1 exactly; those
offset channels are excluded here (`q_minus_one_mask`) and pinned to the restatement in tests/test_oracle_deform2d_conv.py."""
import pytest
import torch

from tests import parity, ref_cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = ("output", "grad_input", "grad_offset", "grad_weight")


@pytest.fixture(scope="module", autouse=True)
def backends(oracle):
    from deformablelka_amd import _lib
    from oracle import ref
    _lib._set_backend_for_tests(None)
    assert torch.cuda.is_available()
    _lib.get_lib()
    if not ref.available():
        pytest.fail("oracle/_ref/D3D.so is missing — run __graft_entry__.build() where /root/reference is mounted")
    ref.D3D()
    yield


def _compare(tag, got, ref, edge):
    parity.assert_close(f"{tag} output", got[0], ref[0], atol=parity.FWD_ATOL)
    parity.assert_close(f"{tag} grad_input", got[1], ref[1], rtol=parity.BWD_RTOL)
    keep = (~edge).to(ref[2].dtype)
    parity.assert_close(f"{tag} grad_offset", got[2].cpu() * keep, ref[2] * keep, rtol=parity.BWD_RTOL)
    parity.assert_close(f"{tag} grad_weight", got[3], ref[3], rtol=parity.BWD_RTOL)


def _is_fast(case):
    B, C, Cout, H, W, k, s, p, d, g, og, mode = case
    return g == C == Cout and C % 32 == 0 and og == 1 and s == 1 and 2 * p == d * (k[0] - 1)


@pytest.mark.parametrize("name", list(ref_cases.CASES_2D))
def test_reference_op_pins_the_2d_oracle_and_kernels(name, oracle):
    from deformablelka_amd import ops
    case = ref_cases.CASES_2D[name]
    t = ref_cases.make2d(case)
    ref, gd = ref_cases.run_ref2d(t, DEV)
    edge = ref_cases.q_minus_one_mask(t)
    x, off, w, go, s, p, d = t["x"], t["off"], t["w"], t["go"], t["s"], t["p"], t["d"]
    # the depth axis carries no information in the embedding: D3D's depth-offset gradient is whatever its formula gives for ld = 0, and
    # nothing here depends on it; but the zero-size depth must not have leaked into the in-plane results: out is finite and non-trivial
    assert torch.isfinite(ref[0]).all() and ref[0].abs().max() > 0
    # (1) 2-D C oracle vs the reference op
    o_out = oracle.deform_conv2d_forward(x, off, w, None, s, p, d)
    o_gi, o_go, o_gw, _ = oracle.deform_conv2d_backward(x, off, w, go, s, p, d)
    _compare("2-D oracle vs reference op", [o_out, o_gi, o_go, o_gw], ref, edge)
    # (2) general 2-D HIP kernels vs the reference op
    xd, od, wd, god = (v.to(DEV) for v in (x, off, w, go))
    h_out = ops.deform_conv2d_forward(xd, od, wd, None, s, p, d)
    h_gi, h_go, h_gw, _ = ops.deform_conv2d_backward(xd, od, wd, god, s, p, d)
    _compare("general 2-D kernels vs reference op", [h_out, h_gi, h_go, h_gw], ref, edge)
    # ... and at q == -1 the kernels follow the restated torchvision rule
    if bool(edge.any()):
        parity.assert_close("general 2-D kernels grad_offset incl. q == -1 vs oracle", h_go, o_go, rtol=parity.BWD_RTOL)
    # index parity through every 2-D site of the rule
    parity.check_index2d(DEV, off, t["H"], t["W"], t["k"], s, p, d, t["og"])
    # (3) channels-last depthwise fast path
    if _is_fast(case):
        cl = lambda v: v.permute(0, 2, 3, 1).contiguous()
        uncl = lambda v: v.permute(0, 3, 1, 2).contiguous()
        f_out = ops.deform_dwconv2d_forward_cl(cl(xd), od, wd, p, d)
        f_gx, f_go, f_gw = ops.deform_dwconv2d_backward_cl(cl(xd), od, wd, cl(god), p, d)
        _compare("cl_ddw2d fast path vs reference op", [uncl(f_out), uncl(f_gx), f_go, f_gw], ref, edge)
        if bool(edge.any()):
            parity.assert_close("cl_ddw2d grad_offset incl. q == -1 vs oracle", f_go, o_go, rtol=parity.BWD_RTOL)
