"""deformablelka_amd — MI355X-native Deformable Large-Kernel Attention (D-LKA) hot path.

Hand-written HIP (gfx950) kernels behind a C-ABI (``include/dlka.h``, ``libdlka_hip.so``), with a Python host side
that mirrors the reference's module / function interface for this path.  PyTorch supplies device memory, streams
and ``torch.distributed`` only.
"""
import sys

from . import D3D, _lib, nn_ops, ops  # noqa: F401
from .deformable_LKA import DeformConv as DeformConv2dPack  # noqa: F401  (2-D "DeformConv" of deformable_LKA.py)
from .deformable_LKA import deformable_LKA, deformable_LKA_Attention  # noqa: F401
from .functions.deform_conv_func import DeformConvFunction  # noqa: F401
from .modules.deform_conv import (DeformConv, DeformConv_d, DeformConvPack, DeformConvPack_d,  # noqa: F401
                                  DeformConvPack_Depth, DeformConvPack_experimental)
from .dynunet_block import UnetResBlock  # noqa: F401
from .transformerblock import LKA3d_deform, LKA_Attention3d_deform, TransformerBlock_3D_single_deform_LKA  # noqa: F401
from .tv_ops import DeformConv2d, deform_conv2d  # noqa: F401
from .network import D_LKA_Former, D_LKA_Former_Encoder, D_LKA_FormerUpBlock, UnetOutBlock  # noqa: F401
from .decoder2d import deformableLKABlock, MyDecoderLayer  # noqa: F401
from . import inference, training, dp  # noqa: F401

__all__ = ["DeformConv", "DeformConvPack", "DeformConvPack_experimental", "DeformConvPack_Depth", "DeformConv_d",
           "DeformConvPack_d", "DeformConvFunction", "LKA3d_deform", "LKA_Attention3d_deform", "TransformerBlock_3D_single_deform_LKA", "UnetResBlock", "DeformConv2dPack",
           "deformable_LKA", "deformable_LKA_Attention", "DeformConv2d", "deform_conv2d", "install_reference_aliases", "D_LKA_Former",
           "D_LKA_Former_Encoder", "D_LKA_FormerUpBlock", "UnetOutBlock", "deformableLKABlock", "MyDecoderLayer", "inference", "training", "dp"]


_ALIAS_DEFAULT = ("D3D", "functions.deform_conv_func", "modules.deform_conv")
# the import paths the reference's NETS use for the deformable conv (SURVEY §8b "import paths to honour"):
#   3D/d_lka_former/network_architecture/synapse/transformerblock.py:568, acdc/transformerblock.py:144   (Synapse / ACDC nets)
#   3D/pancreas_code/networks/d_lka_former/transformerblock.py:569                                        (pancreas net)
# and the modules those files import DeformConvFunction from (synapse/deform_conv.py:13, pancreas deform_conv.py)
NET_ALIASES = ("d_lka_former.network_architecture.synapse.deform_conv", "d_lka_former.network_architecture.synapse.deform_conv_func",
               "networks.d_lka_former.deform_conv", "networks.d_lka_former.deform_conv_func")


def install_reference_aliases(names=_ALIAS_DEFAULT, torchvision_ops=False):
    """Register this package's modules under the import paths the reference's scripts use, so that reference code runs UNCHANGED on top of the
    HIP kernels.  Opt-in; nothing is aliased at import time.

    names — any of: ``D3D`` (deform_conv_func.py:13 ``import D3D``), ``functions.deform_conv_func`` / ``modules.deform_conv`` (3D/dcn/test.py:11-12),
    and the leaf modules of ``NET_ALIASES``: ``d_lka_former.network_architecture.synapse.{deform_conv,deform_conv_func}``,
    ``networks.d_lka_former.{deform_conv,deform_conv_func}`` — what the reference's ``transformerblock.py`` files import ``DeformConvPack`` /
    ``DeformConvPack_Depth`` from.  Only the LEAF is registered in ``sys.modules``: the parent packages stay the reference's own, so its
    ``transformerblock.py``, ``model_components.py``, trainers ... are imported from the reference tree as they are and find this package's
    modules where they expect the deformable conv.  Call it BEFORE importing the reference's net.

    torchvision_ops — the 2-D net's native leaf is ``torchvision.ops.DeformConv2d`` / ``deform_conv2d`` (2D/deformable_LKA/deformable_LKA.py:3,18;
    torchvision 0.12, un-vendored).  True: if torchvision is importable, its ``ops.DeformConv2d`` / ``ops.deform_conv2d`` attributes are
    REPLACED by this package's (same constructor / call signatures, same parameter names: ``tv_ops.py``); if it is not installed, a minimal
    stand-in ``torchvision`` / ``torchvision.ops`` exposing exactly those two names is registered."""
    from . import functions, modules, tv_ops
    from .functions import deform_conv_func
    from .modules import deform_conv
    table = {
        "D3D": D3D,
        "functions": functions, "functions.deform_conv_func": deform_conv_func,
        "modules": modules, "modules.deform_conv": deform_conv,
    }
    leaves = {n: (deform_conv_func if n.endswith("deform_conv_func") else deform_conv) for n in NET_ALIASES}
    for n in names:
        if n in leaves:
            sys.modules[n] = leaves[n]
            continue
        parts = n.split(".")
        hit = False
        for i in range(1, len(parts) + 1):
            key = ".".join(parts[:i])
            if key in table:
                sys.modules[key] = table[key]
                hit = True
        if not hit:
            raise KeyError(f"install_reference_aliases: unknown import path {n!r}")
    if torchvision_ops:
        try:
            import torchvision
            import torchvision.ops as tvo
        except ImportError:
            import types
            torchvision = types.ModuleType("torchvision")
            tvo = types.ModuleType("torchvision.ops")
            torchvision.__doc__ = tvo.__doc__ = "deformablelka_amd stand-in: only ops.DeformConv2d / ops.deform_conv2d (torchvision is not installed)"
            torchvision.ops = tvo
            sys.modules["torchvision"], sys.modules["torchvision.ops"] = torchvision, tvo
        tvo.DeformConv2d, tvo.deform_conv2d = tv_ops.DeformConv2d, tv_ops.deform_conv2d
