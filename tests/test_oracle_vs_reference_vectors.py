"""CPU: what pins the oracle (and, through the emulator, the kernel sources) to the REFERENCE'S OWN ARITHMETIC.

tests/golden/d3d_reference_vectors.pt holds inputs and outputs of the reference's native op — oracle/_ref/D3D.so, i.e.
3D/dcn/src/{vision.cpp,cuda/deform_conv_cuda.cu,cuda/deform_im2col_cuda.cuh} compiled unmodified by oracle/ref.mk and run on an
MI355X by tests/golden/make_ref_golden.py.  Here:
  * the C oracle (oracle/dlka_oracle_impl.h) must reproduce every recorded output — forward 1e-4 abs (measured ~1e-6), gradients
    1e-3 rel (the reference's col2im accumulates with fp32 atomics, cuh:326-328) — with its LITERAL Q1 variant, which the case
    with pad_h != pad_w tells apart from the consistent one;
  * the product's kernel sources, compiled for the host against tests/emu, must reproduce them too (small cases)."""
import os

import pytest
import torch

from tests import parity

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "d3d_reference_vectors.pt")
BLOB = torch.load(PATH, weights_only=True)
NAMES = [k for k in BLOB if k != "_meta"]
OUT = ("out", "grad_input", "grad_offset", "grad_weight", "grad_bias")


def _geometry(rec):
    B, C, Cout, dims, k, s, p, d, g, dg, mode, scale, step = rec["case"]
    t3 = lambda v: (v,) * 3 if isinstance(v, int) else tuple(v)
    return t3(k), t3(s), t3(p), t3(d), g, dg, step


def test_vectors_come_from_the_reference_native_op():
    assert "oracle/_ref/D3D.so" in BLOB["_meta"]["source"] and len(NAMES) >= 10


@pytest.mark.parametrize("name", NAMES)
def test_c_oracle_reproduces_reference_outputs(name, oracle):
    rec = BLOB[name]
    k, s, p, d, g, dg, step = _geometry(rec)
    out = oracle.deform_conv3d_forward(rec["x"], rec["w"], rec["b"], rec["off"], s, p, d, g, dg, step)
    grads = oracle.deform_conv3d_backward(rec["x"], rec["w"], rec["b"], rec["off"], rec["go"], s, p, d, g, dg, step, q1_literal=True)
    parity.assert_close(f"{name} out", out, rec["out"], atol=parity.FWD_ATOL)
    for n, got in zip(OUT[1:], grads):
        parity.assert_close(f"{name} {n}", got, rec[n], rtol=parity.BWD_RTOL)


def test_q1_is_real_in_the_reference(oracle):
    """deformable_col2im_cuda forwards pad_h where pad_w belongs (cuh:447): the reference's recorded grad_input matches the
    literal restatement and NOT the consistent one when pad_h != pad_w.  (The product computes the consistent variant —
    DESIGN.md §3 — which is the true gradient; the D-LKA path has pad_h == pad_w everywhere.)"""
    rec = BLOB["q1_pad_h_ne_pad_w"]
    k, s, p, d, g, dg, step = _geometry(rec)
    assert p[1] != p[2]
    lit = oracle.deform_conv3d_backward(rec["x"], rec["w"], rec["b"], rec["off"], rec["go"], s, p, d, g, dg, step, q1_literal=True)[0]
    con = oracle.deform_conv3d_backward(rec["x"], rec["w"], rec["b"], rec["off"], rec["go"], s, p, d, g, dg, step, q1_literal=False)[0]
    assert parity.rel_err(lit, rec["grad_input"]) < 1e-4
    assert parity.rel_err(con, rec["grad_input"]) > 1e-2


@pytest.fixture()
def emu_backend():
    from deformablelka_amd import _lib
    from tests import emu
    _lib._set_backend_for_tests(emu.load())
    yield
    _lib._set_backend_for_tests(None)


@pytest.mark.parametrize("name", [n for n in NAMES if not n.startswith("k5")])   # k5: 125 taps on the fiber emulator is slow
def test_kernel_sources_on_emulator_reproduce_reference_outputs(name, emu_backend):
    from deformablelka_amd import ops
    rec = BLOB[name]
    k, s, p, d, g, dg, step = _geometry(rec)
    out = ops.deform_conv3d_forward(rec["x"], rec["w"], rec["b"], rec["off"], k, s, p, d, g, dg, step)
    gi, goff, gw, gb = ops.deform_conv3d_backward(rec["x"], rec["w"], rec["b"], rec["off"], rec["go"], k, s, p, d, g, dg, step)
    parity.assert_close(f"{name} out", out, rec["out"], atol=parity.FWD_ATOL)
    if p[1] == p[2]:   # Q1: with pad_h != pad_w the reference's grad_input carries its own slip
        parity.assert_close(f"{name} grad_input", gi, rec["grad_input"], rtol=parity.BWD_RTOL)
    parity.assert_close(f"{name} grad_offset", goff, rec["grad_offset"], rtol=parity.BWD_RTOL)
    parity.assert_close(f"{name} grad_weight", gw, rec["grad_weight"], rtol=parity.BWD_RTOL)
    parity.assert_close(f"{name} grad_bias", gb, rec["grad_bias"], rtol=parity.BWD_RTOL)
