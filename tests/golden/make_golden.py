"""Generates tests/golden/*.pt by running the REFERENCE'S OWN PYTHON MODULES (imported from /root/reference) on CPU.

The reference's native op cannot run here (``D3D`` is a CUDA-only extension whose CPU branch throws —
3D/dcn/src/cpu/deform_cpu.cpp:28,53 — and torchvision is not installed), so the two native leaves are stubbed:
  * ``D3D.deform_conv_forward/backward``  -> the C oracle (oracle/dlka_oracle.c)
  * ``torchvision.ops.DeformConv2d``       -> a module with torchvision's parameter names backed by the C oracle
Everything above the leaves — DeformConvFunction, DeformConvPack(_d), LKA3d_deform, LKA_Attention3d_deform,
DeformConv, deformable_LKA, deformable_LKA_Attention — is the reference's code, executed unmodified.  The vectors
therefore pin (a) the module semantics (parameter names, quirks Q2-Q7, op order, token permutes) against the real
reference and (b) the operator against the oracle.  Run:  python tests/golden/make_golden.py
"""
import math
import os
import sys
import types

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

import oracle  # noqa: E402
from oracle.blocks import randomize_offsets_  # noqa: E402


# ---- stubs for the native leaves and for absent third-party packages --------------------------------------
def _install_stubs(native=True):
    """native=False: only the absent THIRD-PARTY packages (fvcore, MONAI) are stubbed — the native leaves (D3D, torchvision.ops) are left to the
    caller (tests/test_reference_import_paths.py routes them to the product's HIP kernels with install_reference_aliases)."""
    if native:
        _install_native_stubs()
    _install_third_party_stubs()


def _install_native_stubs():
    d3d = types.ModuleType("D3D")

    def fwd(input, weight, bias, offset, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw, group, dg, step):
        return oracle.deform_conv3d_forward(input, weight, bias, offset, (sd, sh, sw), (pd, ph, pw), (dd, dh, dw), group, dg, step)

    def bwd(input, weight, bias, offset, grad_output, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw, group, dg, step):
        return list(oracle.deform_conv3d_backward(input, weight, bias, offset, grad_output.contiguous(), (sd, sh, sw),
                                                  (pd, ph, pw), (dd, dh, dw), group, dg, step, q1_literal=True))

    d3d.deform_conv_forward, d3d.deform_conv_backward = fwd, bwd
    sys.modules["D3D"] = d3d

    tv = types.ModuleType("torchvision")
    tv_ops = types.ModuleType("torchvision.ops")

    class DeformConv2d(nn.Module):  # torchvision 0.12 parameter names / init
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
            super().__init__()
            k = kernel_size if isinstance(kernel_size, tuple) else (kernel_size, kernel_size)
            self.stride, self.padding, self.dilation = stride, padding, dilation
            self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *k))
            nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
            if bias:
                self.bias = nn.Parameter(torch.zeros(out_channels))
            else:
                self.register_parameter("bias", None)

        def forward(self, input, offset, mask=None):
            assert mask is None
            return oracle.DeformConv2dFunction.apply(input, offset, self.weight, self.bias, self.stride, self.padding, self.dilation)

    tv_ops.DeformConv2d = DeformConv2d
    tv.ops = tv_ops
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = tv_ops


def _install_third_party_stubs():
    fv = types.ModuleType("fvcore")
    fvnn = types.ModuleType("fvcore.nn")
    fvnn.FlopCountAnalysis = object
    fv.nn = fvnn
    sys.modules["fvcore"] = fv
    sys.modules["fvcore.nn"] = fvnn

    # MONAI (the reference pins monai==0.8.1 in its requirements; not installed here, not vendored) supplies the three factories
    # UnetResBlock is built from (dynunet_block.py:7-9).  Their published behaviour for the arguments the reference passes:
    #   Convolution(3, cin, cout, strides, kernel_size, bias=False, conv_only=True, padding=p) -> nn.Sequential with ONE child
    #       "conv" = nn.Conv3d(cin, cout, kernel_size, strides, p, bias=False)      (state_dict key "<name>.conv.weight")
    #   get_norm_layer("batch", 3, C)  -> nn.BatchNorm3d(C)  (torch defaults: eps 1e-5, momentum 0.1, affine, running stats)
    #   get_act_layer(("leakyrelu", {"inplace": True, "negative_slope": 0.01})) -> nn.LeakyReLU(0.01, inplace=True)
    for name in ("monai", "monai.networks", "monai.networks.blocks", "monai.networks.blocks.convolutions",
                 "monai.networks.layers", "monai.networks.layers.factories", "monai.networks.layers.utils"):
        sys.modules[name] = types.ModuleType(name)

    class Convolution(nn.Sequential):
        def __init__(self, spatial_dims, in_channels, out_channels, strides=1, kernel_size=3, act=None, norm=None, dropout=None, bias=True,
                     conv_only=False, is_transposed=False, padding=None, output_padding=None):
            super().__init__()
            assert spatial_dims == 3 and conv_only and not is_transposed and dropout is None
            self.add_module("conv", nn.Conv3d(in_channels, out_channels, kernel_size, strides, padding, bias=bias))

    sys.modules["monai.networks.blocks.convolutions"].Convolution = Convolution

    class _AnyAttr:
        def __getattr__(self, n):
            return n
    sys.modules["monai.networks.layers.factories"].Act = _AnyAttr()
    sys.modules["monai.networks.layers.factories"].Norm = _AnyAttr()

    def get_act_layer(name):
        kind, kw = name
        assert kind.lower() == "leakyrelu"
        return nn.LeakyReLU(**kw)

    def get_norm_layer(name, spatial_dims, channels):
        assert name == "batch" and spatial_dims == 3
        return nn.BatchNorm3d(channels)

    sys.modules["monai.networks.layers.utils"].get_act_layer = get_act_layer
    sys.modules["monai.networks.layers.utils"].get_norm_layer = get_norm_layer


def _import_reference():
    _install_stubs()
    # the 3D package's __init__ does ``from . import *`` over heavy subpackages; register bare namespace packages instead
    for name, path in (("d_lka_former", f"{REF}/3D/d_lka_former"),
                       ("d_lka_former.network_architecture", f"{REF}/3D/d_lka_former/network_architecture"),
                       ("d_lka_former.network_architecture.synapse", f"{REF}/3D/d_lka_former/network_architecture/synapse")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    import importlib
    tb = importlib.import_module("d_lka_former.network_architecture.synapse.transformerblock")
    dc = importlib.import_module("d_lka_former.network_architecture.synapse.deform_conv")
    sys.path.insert(0, f"{REF}/3D/dcn")
    dcn_mod = importlib.import_module("modules.deform_conv")          # 3D/dcn/modules/deform_conv.py (has the _d variants)
    sys.path.insert(0, f"{REF}/2D/deformable_LKA")
    lka2d = importlib.import_module("deformable_LKA")                  # 2D/deformable_LKA/deformable_LKA.py
    return tb, dc, dcn_mod, lka2d


def _run(module, inputs, seed, rng_seed=None):
    """forward + backward with a fixed grad_output; returns everything needed to replay on another implementation.
    rng_seed: torch.manual_seed right before the forward (modules that draw dropout noise)."""
    xs = [t.clone().requires_grad_(True) if torch.is_tensor(t) and t.is_floating_point() else t for t in inputs]
    state_before = {k: v.detach().clone() for k, v in module.state_dict().items()}
    if rng_seed is not None:
        torch.manual_seed(rng_seed)
    y = module(*xs)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(seed))
    y.backward(gy)
    return {
        "state_dict": state_before, "rng_seed": rng_seed,
        # what the forward itself changed (BatchNorm running statistics in training mode)
        "state_after": {k: v.detach().clone() for k, v in module.state_dict().items() if not torch.equal(v, state_before[k])},
        "inputs": [t.detach().clone() if torch.is_tensor(t) else t for t in inputs],
        "output": y.detach().clone(), "grad_output": gy,
        "grad_inputs": [t.grad.detach().clone() if torch.is_tensor(t) and t.grad is not None else None for t in xs],
        "grad_params": {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in module.named_parameters()},
    }


def main():
    tb, dc, dcn_mod, lka2d = _import_reference()
    torch.manual_seed(0)
    gold = {}

    # (1) DeformConvPack 3^3 dense — the hot-path configuration (transformerblock.py:639)
    torch.manual_seed(1)
    m = dc.DeformConvPack(in_channels=8, out_channels=8, kernel_size=(3, 3, 3), stride=1, padding=1)
    randomize_offsets_(m, std=0.08)
    gold["DeformConvPack_k3"] = _run(m, [torch.randn(2, 8, 6, 7, 5)], 11)
    gold["DeformConvPack_k3"]["ctor"] = dict(in_channels=8, out_channels=8, kernel_size=(3, 3, 3), stride=1, padding=1)

    # (2) DeformConvPack 5^3 depthwise, zero offsets as constructed (3D/dcn/test.py:28) — known answer == conv3d
    torch.manual_seed(2)
    m = dc.DeformConvPack(in_channels=4, out_channels=4, kernel_size=(5, 5, 5), stride=1, padding=2, groups=4)
    gold["DeformConvPack_k5_dw_zero"] = _run(m, [torch.randn(1, 4, 6, 6, 6)], 12)
    gold["DeformConvPack_k5_dw_zero"]["ctor"] = dict(in_channels=4, out_channels=4, kernel_size=(5, 5, 5), stride=1, padding=2, groups=4)

    # (3) DeformConv (explicit offsets), bias=False quirk Q2: bias still added
    torch.manual_seed(3)
    m = dc.DeformConv(6, 4, (3, 3, 3), 1, 1, dilation=1, groups=2, deformable_groups=2, bias=False)
    off = torch.randn(2, 2 * 81, 5, 4, 6) * 1.5
    gold["DeformConv_g2_dg2_nobias"] = _run(m, [torch.randn(2, 6, 5, 4, 6), off], 13)
    gold["DeformConv_g2_dg2_nobias"]["ctor"] = dict(in_channels=6, out_channels=4, kernel_size=(3, 3, 3), stride=1, padding=1,
                                                     dilation=1, groups=2, deformable_groups=2, bias=False)

    # (4) DeformConvPack_d 'TW' and 'H' with k=3 (the only K for which the reference's hard-coded 81 works, Q7)
    for dim in ("TW", "HW", "H"):  # "THW" raises UnboundLocalError in the reference (no branch for length 3, deform_conv.py:268-318)
        torch.manual_seed(4)
        m = dcn_mod.DeformConvPack_d(4, 4, kernel_size=(3, 3, 3), stride=1, padding=1, dimension=dim)
        randomize_offsets_(m, std=0.1)
        key = f"DeformConvPack_d_{dim}"
        gold[key] = _run(m, [torch.randn(1, 4, 5, 6, 4)], 14)
        gold[key]["ctor"] = dict(in_channels=4, out_channels=4, kernel_size=(3, 3, 3), stride=1, padding=1, dimension=dim)

    # (5) DeformConvPack_Depth (synapse/deform_conv.py:113-158)
    torch.manual_seed(5)
    m = dc.DeformConvPack_Depth(4, 4, kernel_size=(3, 3, 3), stride=1, padding=1)
    with torch.no_grad():
        m.conv_offset.weight.normal_(0, 0.1)
        m.conv_1x1.weight.normal_(0, 0.1)
    gold["DeformConvPack_Depth"] = _run(m, [torch.randn(1, 4, 5, 5, 6)], 15)
    gold["DeformConvPack_Depth"]["ctor"] = dict(in_channels=4, out_channels=4, kernel_size=(3, 3, 3), stride=1, padding=1)

    # (5b) DeformConvPack_experimental (3D/dcn/modules/deform_conv.py:103-139): 1x1x1 channel_adjust C -> 3K, then a DEPTHWISE conv_offset on 3K channels
    torch.manual_seed(55)
    m = dcn_mod.DeformConvPack_experimental(4, 6, kernel_size=(3, 3, 3), stride=1, padding=1)
    with torch.no_grad():
        m.conv_offset.weight.normal_(0, 0.15)
        m.conv_offset.bias.normal_(0, 0.05)
    gold["DeformConvPack_experimental"] = _run(m, [torch.randn(2, 4, 5, 6, 4)], 155)
    gold["DeformConvPack_experimental"]["ctor"] = dict(in_channels=4, out_channels=6, kernel_size=(3, 3, 3), stride=1, padding=1)

    # (6) LKA3d_deform and the full LKA_Attention3d_deform on tokens (transformerblock.py:634-673)
    torch.manual_seed(6)
    m = tb.LKA3d_deform(4)
    randomize_offsets_(m, std=0.05)
    gold["LKA3d_deform"] = _run(m, [torch.randn(1, 4, 6, 5, 7)], 16)
    torch.manual_seed(7)
    B, C, H, W, D = 2, 8, 4, 5, 6
    m = tb.LKA_Attention3d_deform(C)
    randomize_offsets_(m, std=0.05)
    gold["LKA_Attention3d_deform"] = _run(m, [torch.randn(B, H * W * D, C), B, C, H, W, D], 17)

    # (7) 2-D: DeformConv, deformable_LKA_Attention (2D/deformable_LKA/deformable_LKA.py)
    torch.manual_seed(8)
    m = lka2d.DeformConv(6, kernel_size=(5, 5), padding=2, groups=6)
    randomize_offsets_(m, std=0.05)
    gold["DeformConv2d_k5_dw"] = _run(m, [torch.randn(2, 6, 9, 8)], 18)
    torch.manual_seed(9)
    m = lka2d.deformable_LKA_Attention(6)
    randomize_offsets_(m, std=0.03)
    gold["deformable_LKA_Attention"] = _run(m, [torch.randn(2, 6, 12, 11)], 19)

    # (8) the wrapper block TransformerBlock_3D_single_deform_LKA (transformerblock.py:570-630) and its UnetResBlock
    def _liven(m):   # the constructor's gamma = 1e-6 / pos_embed = 0 / BN = identity would hide most of the block
        with torch.no_grad():
            m.gamma.normal_(0.5, 0.2)
            if m.pos_embed is not None:
                m.pos_embed.normal_(0, 0.5)
            m.norm.weight.normal_(1.0, 0.2)
            m.norm.bias.normal_(0, 0.2)
            for bn in (m.conv51.norm1, m.conv51.norm2):
                bn.weight.normal_(1.0, 0.2)
                bn.bias.normal_(0, 0.2)
                bn.running_mean.normal_(0, 0.3)
                bn.running_var.uniform_(0.5, 1.5)
        randomize_offsets_(m, std=0.05)

    B, C, H, W, D = 2, 32, 4, 5, 6
    torch.manual_seed(20)
    m = tb.TransformerBlock_3D_single_deform_LKA(input_size=H * W * D, hidden_size=C, proj_size=32, num_heads=4, dropout_rate=0.1, pos_embed=True)
    _liven(m)
    m.train()
    gold["TransformerBlock_3D_single_deform_LKA_train"] = _run(m, [torch.randn(B, C, H, W, D)], 30, rng_seed=1234)
    gold["TransformerBlock_3D_single_deform_LKA_train"]["ctor"] = dict(input_size=H * W * D, hidden_size=C, proj_size=32, num_heads=4,
                                                                       dropout_rate=0.1, pos_embed=True)
    torch.manual_seed(21)
    m = tb.TransformerBlock_3D_single_deform_LKA(input_size=H * W * D, hidden_size=C, proj_size=32, num_heads=4, dropout_rate=0.1, pos_embed=False)
    _liven(m)
    m.eval()
    gold["TransformerBlock_3D_single_deform_LKA_eval"] = _run(m, [torch.randn(1, C, H, W, D)], 31)
    gold["TransformerBlock_3D_single_deform_LKA_eval"]["ctor"] = dict(input_size=H * W * D, hidden_size=C, proj_size=32, num_heads=4,
                                                                      dropout_rate=0.1, pos_embed=False)
    torch.manual_seed(22)
    from d_lka_former.network_architecture.dynunet_block import UnetResBlock
    m = UnetResBlock(3, C, C, kernel_size=3, stride=1, norm_name="batch")
    with torch.no_grad():
        for bn in (m.norm1, m.norm2):
            bn.weight.normal_(1.0, 0.2)
            bn.bias.normal_(0, 0.2)
    m.train()
    gold["UnetResBlock_train"] = _run(m, [torch.randn(2, C, 5, 4, 6)], 32)

    path = os.path.join(OUT, "reference_modules.pt")
    torch.save(gold, path)
    print("wrote", path, os.path.getsize(path), "bytes;", len(gold), "cases")


if __name__ == "__main__":
    main()
