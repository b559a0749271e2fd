"""Generates tests/golden/reference_nets.pt from the REFERENCE'S OWN network classes (imported from /root/reference, CPU):

  D_LKA_Former_keys        state_dict keys / shapes of the reference's D_LKA_Former(trans_block=TransformerBlock_3D_single_deform_LKA)
                           (3D/d_lka_former/network_architecture/synapse/d_lka_former_synapse.py) — checkpoint compatibility of the assembly.
  D_LKA_Former_plumbing    the same class with a CHEAP stand-in transformer block (both implementations take ``trans_block`` as an
                           argument), run at the full 64x128x128 patch: pins everything AROUND the D-LKA blocks — stem, down-sampling,
                           token reshapes, transposed convs, skip additions, encoder1 / decoder2 conv blocks, deep-supervision heads — at
                           full size.  (The blocks themselves are pinned by reference_modules.pt and the oracle; the full net with real
                           blocks is 21 oracle blocks at full size: minutes of CPU per pass, and its pos_embed sizes are hard-coded to this
                           patch, so it cannot be shrunk.)
  deformableLKABlock, MyDecoderLayer(_last)   2D/networks/MaxViT_deform_LKA.py:142-189,545-620, small sizes, native leaf = the C oracle.

Absent third-party packages are stubbed with their published behaviour for the calls the reference makes (see make_golden.py; here
additionally: timm ``DropPath`` / ``trunc_normal_``, monai ``optional_import``, transposed ``Convolution``, group / instance norm,
batchgenerators ``pad_nd_image`` (never called), the MaxViT encoder module (never constructed)).  Run: python tests/golden/make_golden_nets.py"""
import importlib
import os
import sys
import types

import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"

import make_golden  # noqa: E402
from oracle.blocks import randomize_offsets_  # noqa: E402


class LiteBlock(nn.Module):
    """Cheap transformer-block stand-in with the reference's constructor signature (model_components.py:36-37,130-131): tokens + pos_embed,
    a per-token linear map, back to the volume, plus a 1x1x1 conv.  Used on BOTH sides of the plumbing comparison."""

    def __init__(self, input_size, hidden_size, proj_size, num_heads, dropout_rate=0.0, pos_embed=False):
        super().__init__()
        # a deterministic, NON-persistent position table (a learned (1, N, C) parameter per block would put 25 MB into the fixture)
        pos = 0.1 * torch.sin(torch.arange(input_size, dtype=torch.float32)[:, None] * 0.01 + torch.arange(hidden_size, dtype=torch.float32)[None, :])
        self.register_buffer("pos_embed", pos[None] if pos_embed else None, persistent=False)
        self.lin = nn.Linear(hidden_size, hidden_size)
        self.conv = nn.Conv3d(hidden_size, hidden_size, 1)

    def forward(self, x):
        B, C, H, W, D = x.shape
        t = x.reshape(B, C, H * W * D).permute(0, 2, 1)
        if self.pos_embed is not None:
            t = t + self.pos_embed
        t = t + torch.tanh(self.lin(t))
        v = t.reshape(B, H, W, D, C).permute(0, 4, 1, 2, 3)
        return v + self.conv(v)


def install_net_stubs():
    make_golden._install_stubs()
    # ---- monai: general Convolution (transposed too), norm factory with group / instance, optional_import ----
    class Convolution(nn.Sequential):
        def __init__(self, spatial_dims, in_channels, out_channels, strides=1, kernel_size=3, act=None, norm=None, dropout=None, bias=True,
                     conv_only=False, is_transposed=False, padding=None, output_padding=None):
            super().__init__()
            assert spatial_dims == 3 and conv_only and dropout in (None, 0.0)
            if is_transposed:
                conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, strides, padding, output_padding, bias=bias)
            else:
                conv = nn.Conv3d(in_channels, out_channels, kernel_size, strides, padding, bias=bias)
            self.add_module("conv", conv)
    sys.modules["monai.networks.blocks.convolutions"].Convolution = Convolution

    def get_norm_layer(name, spatial_dims=3, channels=1):
        kind, kw = (name, {}) if isinstance(name, str) else (name[0], dict(name[1]) if len(name) > 1 else {})
        kind = kind.lower()
        if kind == "group":
            return nn.GroupNorm(num_channels=channels, **kw)
        if kind == "instance":
            return nn.InstanceNorm3d(channels, **kw)
        assert kind == "batch"
        return nn.BatchNorm3d(channels, **kw)
    sys.modules["monai.networks.layers.utils"].get_norm_layer = get_norm_layer
    mu = types.ModuleType("monai.utils")

    def optional_import(name):
        try:
            return importlib.import_module(name), True
        except ImportError:
            return None, False
    mu.optional_import = optional_import
    sys.modules["monai.utils"] = mu
    # ---- timm ----
    for n in ("timm", "timm.models", "timm.models.layers"):
        sys.modules[n] = types.ModuleType(n)

    class DropPath(nn.Module):
        def __init__(self, p=0.):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert self.p == 0. or not self.training
            return x
    sys.modules["timm.models.layers"].DropPath = DropPath
    sys.modules["timm.models.layers"].trunc_normal_ = nn.init.trunc_normal_
    sys.modules["timm.models.layers"].to_2tuple = lambda v: (v, v)
    # ---- batchgenerators (pad_nd_image: imported by neural_network.py, used only by the tiled predictor) ----
    for n in ("batchgenerators", "batchgenerators.augmentations", "batchgenerators.augmentations.utils"):
        sys.modules[n] = types.ModuleType(n)
    sys.modules["batchgenerators.augmentations.utils"].pad_nd_image = None


def import_reference_3d():
    install_net_stubs()
    for name, path in (("d_lka_former", f"{REF}/3D/d_lka_former"), ("d_lka_former.network_architecture", f"{REF}/3D/d_lka_former/network_architecture"),
                       ("d_lka_former.network_architecture.synapse", f"{REF}/3D/d_lka_former/network_architecture/synapse"),
                       ("d_lka_former.utilities", f"{REF}/3D/d_lka_former/utilities")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    net = importlib.import_module("d_lka_former.network_architecture.synapse.d_lka_former_synapse")
    tb = importlib.import_module("d_lka_former.network_architecture.synapse.transformerblock")
    return net, tb


def import_reference_2d():
    install_net_stubs()
    for name, path in (("networks", f"{REF}/2D/networks"), ("networks.merit_lib", f"{REF}/2D/networks/merit_lib"), ("deformable_LKA", f"{REF}/2D/deformable_LKA")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    enc = types.ModuleType("networks.merit_lib.networks")     # the MaxViT encoder (timm-based; never constructed here)
    enc.MaxViT4Out_Small = None
    sys.modules["networks.merit_lib.networks"] = enc
    return importlib.import_module("networks.MaxViT_deform_LKA")


def subsample(t, step=8):
    return t[..., ::step, ::step, ::step].clone()


def main():
    gold = {}
    net_mod, tb = import_reference_3d()
    # (a) keys / shapes of the real assembly
    torch.manual_seed(0)
    ref = net_mod.D_LKA_Former(in_channels=1, out_channels=14, img_size=[64, 128, 128], feature_size=16, num_heads=4, depths=[3, 3, 3, 3],
                               dims=[32, 64, 128, 256], do_ds=True, trans_block=tb.TransformerBlock_3D_single_deform_LKA,
                               skip_connections=[True, True, True, True])
    gold["D_LKA_Former_keys"] = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    gold["D_LKA_Former_params"] = sum(p.numel() for p in ref.parameters())
    print("D_LKA_Former:", len(gold["D_LKA_Former_keys"]), "entries,", gold["D_LKA_Former_params"], "parameters")
    del ref
    # (b) plumbing at full size with the cheap block on both sides
    torch.manual_seed(1)
    # (quarter widths — feature_size 4, dims 8..64, hidden_size 64 — keep the fixture small; the dataflow and every spatial size are the
    # reference's)
    ref = net_mod.D_LKA_Former(in_channels=1, out_channels=14, img_size=[64, 128, 128], feature_size=4, hidden_size=64, num_heads=4,
                               depths=[3, 3, 3, 3], dims=[8, 16, 32, 64], do_ds=True, trans_block=LiteBlock, skip_connections=[True, True, True, True])
    ref.eval()
    x = torch.randn(1, 1, 64, 128, 128, generator=torch.Generator().manual_seed(77))
    with torch.no_grad():
        outs = ref(x)
    gold["D_LKA_Former_plumbing"] = {
        "state_dict": {k: v.detach().clone() for k, v in ref.state_dict().items()}, "input_seed": 77,
        "ctor": dict(in_channels=1, out_channels=14, img_size=[64, 128, 128], feature_size=4, hidden_size=64, num_heads=4, depths=[3, 3, 3, 3],
                     dims=[8, 16, 32, 64], do_ds=True, skip_connections=[True, True, True, True]),
        "out_shapes": [tuple(o.shape) for o in outs], "out_sub": [subsample(o) for o in outs],
        "out_mean": [float(o.double().mean()) for o in outs], "out_abs_mean": [float(o.double().abs().mean()) for o in outs]}
    print("plumbing:", [tuple(o.shape) for o in outs], gold["D_LKA_Former_plumbing"]["out_abs_mean"])
    del ref
    # (c) 2-D decoder pieces
    m2 = import_reference_2d()
    torch.manual_seed(40)
    blk = m2.deformableLKABlock(dim=8)
    randomize_offsets_(blk, std=0.05)
    with torch.no_grad():
        blk.layer_scale_1.normal_(0.5, 0.1)
        blk.layer_scale_2.normal_(0.5, 0.1)
    gold["deformableLKABlock"] = make_golden._run(blk, [torch.randn(2, 7 * 6, 8), 7, 6], 50)
    gold["deformableLKABlock"]["ctor"] = dict(dim=8)
    for last in (False, True):
        torch.manual_seed(41 + int(last))
        lay = m2.MyDecoderLayer((5, 6), [8, 8, 8, 8, 8], 1, "mix_skip", n_class=3, is_last=last)
        randomize_offsets_(lay, std=0.05)
        key = "MyDecoderLayer_last" if last else "MyDecoderLayer"
        gold[key] = make_golden._run(lay, [torch.randn(2, 30, 8), torch.randn(2, 5, 6, 8)], 51 + int(last))
        gold[key]["ctor"] = dict(input_size=(5, 6), in_out_chan=[8, 8, 8, 8, 8], head_count=1, token_mlp_mode="mix_skip", n_class=3, is_last=last)
    torch.manual_seed(43)
    lay = m2.MyDecoderLayer((5, 6), [8, 8, 8, 8, 8], 1, "mix_skip", n_class=3)
    gold["MyDecoderLayer_noskip"] = make_golden._run(lay, [torch.randn(2, 30, 8)], 53)
    gold["MyDecoderLayer_noskip"]["ctor"] = gold["MyDecoderLayer"]["ctor"]
    path = os.path.join(HERE, "reference_nets.pt")
    torch.save(gold, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
