"""CPU: the oracle's block compositions (oracle/blocks.py) reproduce the vectors the reference's own Python modules
produced (tests/golden/make_golden.py).  Pins the block-level restatement."""
import torch

from oracle import blocks
from tests.golden_checks import gold
from tests.parity import assert_close


def _strip(sd):
    return {k: v for k, v in sd.items()}


def test_lka3d_attention_tokens_matches_reference_module(oracle):
    c = gold()["LKA_Attention3d_deform"]
    x, B, C, H, W, D = c["inputs"]
    P = {k: v.clone().requires_grad_(True) for k, v in c["state_dict"].items()}
    xr = x.clone().requires_grad_(True)
    y = blocks.lka3d_attention_tokens(xr, P, B, C, H, W, D)
    assert_close("y", y, c["output"], atol=1e-5)
    y.backward(c["grad_output"])
    assert_close("gx", xr.grad, c["grad_inputs"][0], rtol=1e-4)
    for k, g in c["grad_params"].items():
        if g is not None and g.abs().max() > 0:
            assert_close(k, P[k].grad, g, rtol=1e-4)


def test_lka2d_attention_matches_reference_module(oracle):
    c = gold()["deformable_LKA_Attention"]
    (x,) = c["inputs"]
    P = {k: v.clone().requires_grad_(True) for k, v in c["state_dict"].items()}
    xr = x.clone().requires_grad_(True)
    y = blocks.lka2d_attention(xr, P)
    assert_close("y", y, c["output"], atol=1e-5)
    y.backward(c["grad_output"])
    assert_close("gx", xr.grad, c["grad_inputs"][0], rtol=1e-4)
    for k, g in c["grad_params"].items():
        if g is not None and g.abs().max() > 0:
            assert_close(k, P[k].grad, g, rtol=1e-4)


def test_zero_offset_pack_equals_conv3d(oracle):
    """Known answer through the reference module: fresh DeformConvPack == nn.Conv3d with the same weights."""
    c = gold()["DeformConvPack_k5_dw_zero"]
    sd = c["state_dict"]
    ref = torch.nn.functional.conv3d(c["inputs"][0], sd["weight"], sd["bias"], 1, 2, 1, 4)
    assert_close("y", c["output"], ref, atol=2e-5)
