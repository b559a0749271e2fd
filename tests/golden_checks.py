"""Replays tests/golden/reference_modules.pt (made by the reference's own Python modules, see make_golden.py) on
the deformablelka_amd modules: state_dict must load with strict=True (checkpoint compatibility), forward
within 1e-4, backward within 1e-3 relative."""
import os

import torch

import deformablelka_amd as dk
from tests.parity import assert_close

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_modules.pt")
GOLD_NETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_nets.pt")   # tests/golden/make_golden_nets.py
_cache = {}


def gold():
    if "g" not in _cache:
        _cache["g"] = torch.load(GOLD, weights_only=False)
        _cache["g"].update({k: v for k, v in gold_nets().items() if isinstance(v, dict) and "state_dict" in v and "inputs" in v})
    return _cache["g"]


def gold_nets():
    if "n" not in _cache:
        _cache["n"] = torch.load(GOLD_NETS, weights_only=False)
    return _cache["n"]


def build(name, case):
    if name.startswith("DeformConvPack_d_"):
        return dk.DeformConvPack_d(**case["ctor"])
    if name == "DeformConvPack_experimental":
        return dk.DeformConvPack_experimental(**case["ctor"])
    if name == "DeformConvPack_Depth":
        return dk.DeformConvPack_Depth(**case["ctor"])
    if name.startswith("DeformConvPack_"):
        return dk.DeformConvPack(**case["ctor"])
    if name.startswith("DeformConv_"):
        return dk.DeformConv(**case["ctor"])
    if name == "LKA3d_deform":
        return dk.LKA3d_deform(case["inputs"][0].shape[1])
    if name == "LKA_Attention3d_deform":
        return dk.LKA_Attention3d_deform(case["inputs"][2])
    if name.startswith("TransformerBlock_3D_single_deform_LKA"):
        return dk.TransformerBlock_3D_single_deform_LKA(**case["ctor"])
    if name.startswith("UnetResBlock"):
        C = case["inputs"][0].shape[1]
        return dk.UnetResBlock(3, C, C, kernel_size=3, stride=1, norm_name="batch")
    if name == "DeformConv2d_k5_dw":
        return dk.DeformConv2dPack(6, kernel_size=(5, 5), padding=2, groups=6)
    if name == "deformable_LKA_Attention":
        return dk.deformable_LKA_Attention(case["inputs"][0].shape[1])
    if name == "deformableLKABlock":
        return dk.deformableLKABlock(**case["ctor"])
    if name.startswith("MyDecoderLayer"):
        return dk.MyDecoderLayer(**case["ctor"])
    raise KeyError(name)


def replay(name, dev, fwd_atol=1e-4, bwd_rtol=1e-3):
    case = gold()[name]
    m = build(name, case)
    m.load_state_dict(case["state_dict"], strict=True)
    m = m.to(dev)
    m.train(not name.endswith("_eval"))
    xs = [t.to(dev).clone().requires_grad_(True) if torch.is_tensor(t) and t.is_floating_point() else t for t in case["inputs"]]
    if case.get("rng_seed") is not None:   # dropout noise: the reference run drew it from the CPU generator; draw there on any device
        torch.manual_seed(case["rng_seed"])
        draw = m._draw_drop_mask
        m._draw_drop_mask = lambda B, C, dtype, device: draw(B, C, dtype, "cpu").to(device)
    y = m(*xs)
    assert_close(name + " output", y, case["output"], atol=fwd_atol)
    for k, v in case.get("state_after", {}).items():
        got = m.state_dict()[k]
        if v.is_floating_point():
            assert_close(f"{name} buffer {k}", got, v, rtol=1e-4)
        else:
            assert torch.equal(got.cpu(), v), k
    y.backward(case["grad_output"].to(dev))
    for i, (x, g) in enumerate(zip(xs, case["grad_inputs"])):
        if g is not None:
            assert_close(f"{name} grad_input[{i}]", x.grad, g, rtol=bwd_rtol)
    for k, p in m.named_parameters():
        g = case["grad_params"][k]
        if g is None:
            continue
        if g.abs().max() == 0:
            assert p.grad is None or p.grad.abs().max().item() < 1e-6, k
        else:
            assert p.grad is not None, k
            assert_close(f"{name} grad {k}", p.grad, g, rtol=bwd_rtol)
