"""CPU: what pins the 2-D (torchvision-semantics) oracle and kernels to the reference's arithmetic.

(a) tests/golden/d3d_reference_vectors_2d.pt — outputs of the REFERENCE'S OWN op (oracle/_ref/D3D.so) on the D = 1 embedding of the 2-D
    cases (tests/ref_cases.py; recorded on an MI355X by tests/golden/make_ref_golden.py): the 2-D C oracle, the general 2-D kernel sources
    and the channels-last depthwise fast path (both on the emulator) must reproduce them;
(b) independent of the fixture: the 3-D C oracle — itself pinned to the reference op — evaluated on the embedding equals the 2-D C oracle;
(c) the one torchvision line that cannot be pinned this way, the UNGUARDED coordinate weight at q == -1, is checked against its restated
    rule on a constructed case."""
import os

import pytest
import torch

from tests import parity, ref_cases

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "d3d_reference_vectors_2d.pt")
BLOB = torch.load(PATH, weights_only=True) if os.path.exists(PATH) else {}
NAMES = [k for k in BLOB if k != "_meta"]


def _compare(tag, got, ref, edge):
    parity.assert_close(f"{tag} output", got[0], ref[0], atol=parity.FWD_ATOL)
    parity.assert_close(f"{tag} grad_input", got[1], ref[1], rtol=parity.BWD_RTOL)
    keep = (~edge).to(ref[2].dtype)
    parity.assert_close(f"{tag} grad_offset", got[2] * keep, ref[2] * keep, rtol=parity.BWD_RTOL)
    parity.assert_close(f"{tag} grad_weight", got[3], ref[3], rtol=parity.BWD_RTOL)


def _t(rec):
    B, C, Cout, H, W, k, s, p, d, g, og, mode = rec["case"]
    return dict(x=rec["x"], off=rec["off"], w=rec["w"], go=rec["go"], k=tuple(k), s=s, p=p, d=d, g=g, og=og, H=H, W=W)


def test_fixture_is_present_and_comes_from_the_reference_op():
    assert NAMES, "tests/golden/d3d_reference_vectors_2d.pt is missing (tests/golden/make_ref_golden.py, GPU box)"
    assert "oracle/_ref/D3D.so" in BLOB["_meta"]["source"] and "D = 1" in BLOB["_meta"]["embedding"]
    assert {"dlka_dw5", "dlka_dw7_dil3", "dense3"} <= set(NAMES)


@pytest.mark.parametrize("name", NAMES)
def test_2d_oracle_reproduces_reference_op_vectors(name, oracle):
    rec = BLOB[name]
    t = _t(rec)
    out = oracle.deform_conv2d_forward(t["x"], t["off"], t["w"], None, t["s"], t["p"], t["d"])
    gi, go, gw, _ = oracle.deform_conv2d_backward(t["x"], t["off"], t["w"], t["go"], t["s"], t["p"], t["d"])
    _compare(f"{name}: 2-D oracle vs reference op", [out, gi, go, gw], [rec[k] for k in ("out", "grad_input", "grad_offset", "grad_weight")],
             ref_cases.q_minus_one_mask(t))


@pytest.mark.parametrize("name", list(ref_cases.CASES_2D))
def test_3d_oracle_on_the_embedding_equals_2d_oracle(name, oracle):
    t = ref_cases.make2d(ref_cases.CASES_2D[name])
    e = ref_cases.embed2d(t)
    out3 = oracle.deform_conv3d_forward(e["x"], e["w"], e["b"], e["off"], e["s"], e["p"], e["d"], e["g"], e["dg"], e["step"])
    g3 = oracle.deform_conv3d_backward(e["x"], e["w"], e["b"], e["off"], e["go"], e["s"], e["p"], e["d"], e["g"], e["dg"], e["step"], q1_literal=False)
    ref, _ = ref_cases.project2d(t, [out3, *g3])
    out = oracle.deform_conv2d_forward(t["x"], t["off"], t["w"], None, t["s"], t["p"], t["d"])
    gi, go, gw, _ = oracle.deform_conv2d_backward(t["x"], t["off"], t["w"], t["go"], t["s"], t["p"], t["d"])
    _compare(f"{name}: 2-D oracle vs 3-D oracle on the embedding", [out, gi, go, gw], ref, ref_cases.q_minus_one_mask(t))


def test_unguarded_coordinate_weight_at_q_minus_one(oracle):
    """torchvision get_coordinate_weight has per-corner bounds but no guard: a sample at qy == -1 exactly contributes NOTHING to the output
    (bilinear_interpolate returns 0 for h <= -1) yet its offset gradient sees the row y = 0 with weight +1 (d/dy of the interpolant between the
    virtual row -1 = 0 and row 0).  D3D zeroes that gradient (deform_im2col_cuda.cuh:391-394) — the one place the two operators differ."""
    H = W = 4
    x = torch.arange(1.0, 1 + H * W).reshape(1, 1, H, W)
    w = torch.ones(1, 1, 1, 1)
    off = torch.zeros(1, 2, H, W)
    off[0, 0, 0, :] = -1.0                      # output row 0 samples qy = 0 - 1 = -1 exactly
    go = torch.ones(1, 1, H, W)
    out = oracle.deform_conv2d_forward(x, off, w, None, 1, 0, 1)
    assert torch.equal(out[0, 0, 0], torch.zeros(W)) and torch.equal(out[0, 0, 1:], x[0, 0, 1:])
    gi, goff, gw, _ = oracle.deform_conv2d_backward(x, off, w, go, 1, 0, 1)
    assert torch.allclose(goff[0, 0, 0], x[0, 0, 0])          # d/dy = +1 * x[row 0] - 1 * (row -1 = 0)
    assert torch.equal(gi[0, 0, 0], torch.zeros(W))           # the guarded sample scatters nothing
    e = ref_cases.embed2d(dict(x=x, off=off, w=w, go=go, k=(1, 1), s=1, p=0, d=1, g=1, og=1, H=H, W=W))
    g3 = oracle.deform_conv3d_backward(e["x"], e["w"], e["b"], e["off"], e["go"], e["s"], e["p"], e["d"], 1, 1, 64, q1_literal=False)
    assert torch.equal(g3[1][0, 1, 0, 0], torch.zeros(W))     # D3D: guarded -> 0


@pytest.fixture()
def emu_backend():
    from deformablelka_amd import _lib
    from tests import emu
    _lib._set_backend_for_tests(emu.load())
    yield
    _lib._set_backend_for_tests(None)


@pytest.mark.parametrize("name", [n for n in NAMES if "dw7" not in n or "integer" in n])   # (one 49-tap case is enough on the fiber emulator)
def test_kernel_sources_on_emulator_reproduce_reference_op_vectors_2d(name, emu_backend, oracle):
    from deformablelka_amd import ops
    rec = BLOB[name]
    t = _t(rec)
    ref = [rec[k] for k in ("out", "grad_input", "grad_offset", "grad_weight")]
    edge = ref_cases.q_minus_one_mask(t)
    x, off, w, go, s, p, d = t["x"], t["off"], t["w"], t["go"], t["s"], t["p"], t["d"]
    out = ops.deform_conv2d_forward(x, off, w, None, s, p, d)
    gi, goff, gw, _ = ops.deform_conv2d_backward(x, off, w, go, s, p, d)
    _compare(f"{name}: general 2-D kernels (emulator) vs reference op", [out, gi, goff, gw], ref, edge)
    parity.check_index2d("cpu", off, t["H"], t["W"], t["k"], s, p, d, t["og"])
    B, C, Cout, H, W, k, *_ = rec["case"]
    if t["g"] == C == Cout and C % 32 == 0 and t["og"] == 1 and s == 1:
        cl = lambda v: v.permute(0, 2, 3, 1).contiguous()
        uncl = lambda v: v.permute(0, 3, 1, 2).contiguous()
        f_out = ops.deform_dwconv2d_forward_cl(cl(x), off, w, p, d)
        f_gx, f_go, f_gw = ops.deform_dwconv2d_backward_cl(cl(x), off, w, cl(go), p, d)
        _compare(f"{name}: cl_ddw2d fast path (emulator) vs reference op", [uncl(f_out), uncl(f_gx), f_go, f_gw], ref, edge)
        if bool(edge.any()):
            o_go = oracle.deform_conv2d_backward(x, off, w, go, s, p, d)[1]
            parity.assert_close("cl_ddw2d grad_offset incl. q == -1 vs oracle", f_go, o_go, rtol=parity.BWD_RTOL)
