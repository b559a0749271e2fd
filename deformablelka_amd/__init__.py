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


def install_reference_aliases(names=("D3D", "functions.deform_conv_func", "modules.deform_conv")):
    """Register this package's modules under the import paths the reference's scripts use
    (3D/dcn/test.py:11-12 ``from modules.deform_conv import ...``; deform_conv_func.py:13 ``import D3D``), so that
    reference code runs unchanged on top of the HIP kernels.  Opt-in; nothing is aliased at import time."""
    from . import functions, modules
    from .functions import deform_conv_func
    from .modules import deform_conv
    table = {
        "D3D": D3D,
        "functions": functions, "functions.deform_conv_func": deform_conv_func,
        "modules": modules, "modules.deform_conv": deform_conv,
    }
    for n in names:
        parts = n.split(".")
        for i in range(1, len(parts) + 1):
            key = ".".join(parts[:i])
            if key in table:
                sys.modules[key] = table[key]
