"""The zero-edit route of INTEGRATION.md §2a, executed: the REFERENCE'S OWN Python files (3D/dcn/functions/deform_conv_func.py,
3D/dcn/modules/deform_conv.py, imported from /root/reference) run on top of this repo's ``D3D`` shim after
``install_reference_aliases()``, and a reference-constructed TransformerBlock_3D_single_deform_LKA's state_dict loads strictly
into the repo's module.  /root/reference exists only in the build container (not on the GPU box), so the native side here is
the host build of the kernel sources on the wavefront emulator (tests/emu) — same C-ABI, same Python host code.
Each scenario runs in a fresh interpreter: they rewrite sys.modules."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF + "/3D/dcn"), reason="/root/reference is not mounted here")


def _run(body):
    env = dict(os.environ, PYTHONPATH=ROOT)
    code = textwrap.dedent(PRELUDE) + textwrap.dedent(body)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    return r.stdout


PRELUDE = """
import sys, torch
import deformablelka_amd as dk
from deformablelka_amd import _lib
from tests import emu, parity
import oracle
_lib._set_backend_for_tests(emu.load())
REF = "/root/reference"
"""


@needs_ref
def test_reference_import_paths():
    """3D/dcn/functions/deform_conv_func.py:13 does ``import D3D``; with the alias in place that is this repo's shim, and the
    reference's DeformConvFunction / DeformConvPack / DeformConv run unmodified — the two smoke examples of 3D/dcn/test.py:62-86
    (example_dconv, example_dconv_offset; k=5 p=2, shrunk volume) against the oracle."""
    out = _run("""
    dk.install_reference_aliases(names=("D3D",))
    sys.path.insert(0, REF + "/3D/dcn")
    import types   # fvcore (a FLOP counter the reference imports at the bottom of modules/deform_conv.py:328) is not installed
    fv, fvnn = types.ModuleType("fvcore"), types.ModuleType("fvcore.nn")
    fvnn.FlopCountAnalysis = object; fv.nn = fvnn
    sys.modules["fvcore"], sys.modules["fvcore.nn"] = fv, fvnn
    from functions import deform_conv_func as ref_func          # the reference's file
    from modules import deform_conv as ref_mod                   # the reference's file
    assert ref_func.__file__.startswith(REF) and ref_mod.__file__.startswith(REF)
    import D3D
    assert D3D is dk.D3D and ref_func.D3D is dk.D3D
    B, inC, outC, T, H, W, k = 2, 4, 6, 6, 5, 7, 5
    torch.manual_seed(0)

    def check(tag, run_ref_module, params, x, off_in):
        # reference module on the HIP-kernel sources (emulator) ...
        xs = x.clone().requires_grad_(True)
        y = run_ref_module(xs)
        target = torch.empty_like(y).uniform_(-0.01, 0.01)      # 3D/dcn/test.py:69-72
        (target - y).mean().backward()
        # ... against the oracle with the same parameters
        P = {n: p.detach().clone().requires_grad_(True) for n, p in params.items()}
        xr = x.clone().requires_grad_(True)
        off = off_in if off_in is not None else torch.nn.functional.conv3d(xr, P["conv_offset.weight"], P["conv_offset.bias"], padding=2)
        yr = oracle.DeformConv3dFunction.apply(xr, off, P["weight"], P["bias"], 1, 2, 1, 1, 1, 64)
        (target - yr).mean().backward()
        parity.assert_close(tag + " y", y, yr.detach(), atol=1e-4)
        parity.assert_close(tag + " gx", xs.grad, xr.grad, rtol=1e-3)
        for n, p in params.items():
            if P[n].grad is not None and P[n].grad.abs().max() > 0:
                parity.assert_close(tag + " grad " + n, p.grad, P[n].grad, rtol=1e-3)
        print(tag, "ok")

    x = torch.randn(B, inC, T, H, W)
    dcn = ref_mod.DeformConvPack(inC, outC, kernel_size=[k, k, k], stride=[1, 1, 1], padding=[2, 2, 2])   # example_dconv
    with torch.no_grad():
        dcn.conv_offset.weight.normal_(0, 0.05)      # fresh modules predict zero offsets (deform_conv.py:86-88)
    check("example_dconv", lambda t: dcn(t), dict(dcn.named_parameters()), x, None)
    off = torch.randn(B, k * k * k * 3, T, H, W)
    dc = ref_mod.DeformConv(inC, outC, kernel_size=[k, k, k], stride=[1, 1, 1], padding=[2, 2, 2], dilation=[1, 1, 1])   # example_dconv_offset
    check("example_dconv_offset", lambda t: dc(t, off), dict(dc.named_parameters()), x, off)
    """)
    assert "example_dconv ok" in out and "example_dconv_offset ok" in out


def test_aliases_serve_the_reference_scripts_imports():
    """3D/dcn/test.py:11-12 — ``from modules.deform_conv import DeformConv, _DeformConv, DeformConvPack, DeformConv_d, DeformConvPack_d``
    resolves to this package once the aliases are installed (the route for scripts that do not ship the reference's files)."""
    _run("""
    dk.install_reference_aliases()
    from modules.deform_conv import DeformConv, _DeformConv, DeformConvPack, DeformConv_d, DeformConvPack_d
    from functions.deform_conv_func import DeformConvFunction
    import D3D
    assert DeformConvPack is dk.DeformConvPack and DeformConv_d is dk.DeformConv_d and DeformConvFunction is dk.DeformConvFunction
    m = DeformConvPack_d(4, 4, kernel_size=3, stride=1, padding=1, dimension="TW")
    y = m(torch.randn(1, 4, 5, 5, 5))
    assert y.shape == (1, 4, 5, 5, 5)
    """)


@needs_ref
def test_reference_constructed_block_state_dict_loads_strictly():
    """A TransformerBlock_3D_single_deform_LKA built by the REFERENCE'S class (transformerblock.py:570-615; MONAI factories restated
    as in tests/golden/make_golden.py) hands its state_dict to the repo's module with strict=True, and the output is a contiguous
    NCDHW tensor like the reference's (transformerblock.py:626-630)."""
    _run("""
    sys.path.insert(0, "tests/golden")
    import make_golden
    tb, dc, dcn_mod, lka2d = make_golden._import_reference()
    torch.manual_seed(1)
    C, H, W, D = 32, 4, 4, 4
    ref = tb.TransformerBlock_3D_single_deform_LKA(input_size=H * W * D, hidden_size=C, proj_size=C, num_heads=4, dropout_rate=0.1, pos_embed=True)
    ours = dk.TransformerBlock_3D_single_deform_LKA(input_size=H * W * D, hidden_size=C, proj_size=C, num_heads=4, dropout_rate=0.1, pos_embed=True)
    missing = ours.load_state_dict(ref.state_dict(), strict=True)
    assert set(ref.state_dict()) == set(ours.state_dict())
    for k, v in ref.state_dict().items():
        assert ours.state_dict()[k].shape == v.shape, k
    ref.eval(); ours.eval()
    x = torch.randn(1, C, H, W, D)
    y, yr = ours(x), ref(x)
    assert y.is_contiguous() and y.shape == yr.shape
    parity.assert_close("block eval y", y, yr.detach(), atol=1e-4)
    y.view(1, -1)          # a downstream .view() must work, as it does on the reference's output
    """)


@needs_ref
def test_reference_synapse_transformerblock_runs_unmodified_on_the_hip_path():
    """INTEGRATION.md §2a for the NETS: the reference's own 3D/d_lka_former/network_architecture/synapse/transformerblock.py — imported from
    /root/reference, not edited — finds ``DeformConvPack`` where it imports it from (``d_lka_former.network_architecture.synapse.deform_conv``,
    transformerblock.py:568) = this package's module, which runs the HIP kernel sources (emulator here).  Only the absent third-party packages
    (MONAI factories, fvcore) are restated; ``import torchvision`` (:421) is served by the opt-in stand-in.  The reference's
    LKA_Attention3d_deform and its whole TransformerBlock_3D_single_deform_LKA are compared with the oracle."""
    out = _run("""
    import types, importlib
    sys.path.insert(0, "tests/golden")
    import make_golden
    make_golden._install_stubs(native=False)                     # MONAI / fvcore only — NOT the oracle-backed D3D / torchvision stubs
    dk.install_reference_aliases(names=("D3D",) + dk.NET_ALIASES, torchvision_ops=True)
    sys.path.insert(0, REF + "/3D")
    for name in ("d_lka_former", "d_lka_former.network_architecture", "d_lka_former.network_architecture.synapse"):
        m = types.ModuleType(name)                               # (the package's real __init__ star-imports the trainers: nnU-Net, batchgenerators ...)
        m.__path__ = [REF + "/3D/" + name.replace(".", "/")]
        sys.modules[name] = m
    tb = importlib.import_module("d_lka_former.network_architecture.synapse.transformerblock")
    assert tb.__file__.startswith(REF)
    assert tb.DeformConvPack is dk.DeformConvPack, "the reference's transformerblock.py did not pick up the HIP-backed DeformConvPack"
    from oracle import blocks
    torch.manual_seed(0)
    B, C, H, W, D = 2, 32, 4, 5, 6
    m = tb.LKA_Attention3d_deform(C)                             # the REFERENCE's class
    blocks.randomize_offsets_(m, std=0.05)
    x = torch.randn(B, H * W * D, C)
    gy = torch.randn(B, H * W * D, C)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    yr = blocks.lka3d_attention_tokens(xr, P, B, C, H, W, D)
    yr.backward(gy)
    xs = x.clone().requires_grad_(True)
    y = m(xs, B, C, H, W, D)
    y.backward(gy)
    parity.assert_close("ref LKA_Attention3d_deform y", y, yr.detach(), atol=1e-4)
    parity.assert_close("ref LKA_Attention3d_deform gx", xs.grad, xr.grad, rtol=1e-3)
    for k, p in m.named_parameters():
        if P[k].grad is not None and P[k].grad.abs().max() > 0:
            parity.assert_close("ref LKA_Attention3d_deform grad " + k, p.grad, P[k].grad, rtol=1e-3)
    print("lka3d ok")
    blk = tb.TransformerBlock_3D_single_deform_LKA(input_size=H * W * D, hidden_size=C, proj_size=C, num_heads=4, dropout_rate=0.1, pos_embed=True)
    blocks.randomize_offsets_(blk, std=0.05)
    with torch.no_grad():
        blk.gamma.normal_(0.5, 0.2)
    blk.eval()
    xv = torch.randn(1, C, H, W, D)
    Pb = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    parity.assert_close("ref wrapper block y", blk(xv), blocks.transformer_block_3d(xv, Pb, False, None), atol=1e-4)
    print("tblock ok")
    """)
    assert "lka3d ok" in out and "tblock ok" in out


@needs_ref
def test_reference_2d_deformable_lka_runs_unmodified_on_the_hip_path():
    """2D/deformable_LKA/deformable_LKA.py — the reference's file, not edited — with ``torchvision.ops.DeformConv2d`` (:18, torchvision is not
    installed here) served by install_reference_aliases(torchvision_ops=True): its deformable_LKA_Attention against the oracle block."""
    out = _run("""
    sys.path.insert(0, "tests/golden")
    import make_golden
    make_golden._install_third_party_stubs()                     # fvcore (a FLOP counter the file imports at :160) is not installed; nothing native
    dk.install_reference_aliases(names=(), torchvision_ops=True)
    import torchvision
    assert torchvision.ops.DeformConv2d is dk.DeformConv2d
    sys.path.insert(0, REF + "/2D/deformable_LKA")
    import deformable_LKA as ref2d
    assert ref2d.__file__.startswith(REF)
    from oracle import blocks
    torch.manual_seed(0)
    B, C, H, W = 2, 32, 9, 8
    m = ref2d.deformable_LKA_Attention(C)                        # the REFERENCE's class
    blocks.randomize_offsets_(m, std=0.03)
    x = torch.randn(B, C, H, W)
    gy = torch.randn(B, C, H, W)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    yr = blocks.lka2d_attention(xr, P)
    yr.backward(gy)
    xs = x.clone().requires_grad_(True)
    y = m(xs)
    y.backward(gy)
    parity.assert_close("ref deformable_LKA_Attention y", y, yr.detach(), atol=1e-4)
    parity.assert_close("ref deformable_LKA_Attention gx", xs.grad, xr.grad, rtol=1e-3)
    for k, p in m.named_parameters():
        if P[k].grad is not None and P[k].grad.abs().max() > 0:
            parity.assert_close("ref deformable_LKA_Attention grad " + k, p.grad, P[k].grad, rtol=1e-3)
    print("lka2d ok")
    """)
    assert "lka2d ok" in out


def test_net_alias_names_resolve_to_this_package():
    """The pancreas net's path (3D/pancreas_code/networks/d_lka_former/transformerblock.py:569) and the Synapse / ACDC one resolve to this package's
    modules once registered; an unknown path is an error, not a silent no-op."""
    _run("""
    dk.install_reference_aliases(names=dk.NET_ALIASES)
    import importlib
    for n in dk.NET_ALIASES:
        m = importlib.import_module(n)
        assert m.__name__.startswith("deformablelka_amd."), (n, m.__name__)
    from networks.d_lka_former.deform_conv import DeformConvPack, DeformConvPack_Depth
    from d_lka_former.network_architecture.synapse.deform_conv_func import DeformConvFunction
    assert DeformConvPack is dk.DeformConvPack and DeformConvFunction is dk.DeformConvFunction
    try:
        dk.install_reference_aliases(names=("no.such.module",))
    except KeyError:
        pass
    else:
        raise AssertionError("unknown alias accepted")
    """)


def test_deform_b16_switch_off_keeps_the_fp32_input_mfma_kernels_correct():
    """DLKA_DEFORM_B16=0 (read once per process: hence a subprocess) restores the round-3 kernels — fp32-input MFMA in the deformable conv's forward and
    backward contractions for both activation types.  They are the A/B reference of profiles/r05_notes.md; this keeps them under test."""
    env_body = """
    import os
    assert os.environ.get("DLKA_DEFORM_B16") == "0"
    parity.check_lka3d_tokens("cpu", 1, 32, (3, 4, 5), offset_std=0.3)
    parity.check_lka3d_tokens("cpu", 1, 64, (2, 3, 4), offset_std=0.3)
    parity.check_lka3d_tokens_bf16("cpu", 1, 32, (4, 4, 8))
    print("b16 off ok")
    """
    env = dict(os.environ, PYTHONPATH=ROOT, DLKA_DEFORM_B16="0")
    code = textwrap.dedent(PRELUDE) + textwrap.dedent(env_body)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "b16 off ok" in r.stdout, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
