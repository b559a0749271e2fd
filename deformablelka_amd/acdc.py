"""The ACDC variant of the 3-D D-LKA modules — 3D/d_lka_former/network_architecture/acdc/transformerblock.py:140-262: same block, but the
depthwise pair of ``LKA3d_deform`` depends on the width (:213-237) and the net's stem is (1, 4, 4) (acdc/model_components.py:21), which is the
stem BASELINE.json config 5's 40x224x224 tiles divide through (stage shapes 40x56x56 / 20x28x28 / 10x14x14 / 5x7x7).  Same constructor / forward
signatures and ``state_dict`` keys as the reference classes; the fused entry points take the variant (include/dlka.h: DLKA_LKA3D_ACDC)."""
import torch.nn as nn

from . import transformerblock as _tb


class LKA3d_deform(_tb.LKA3d_deform):
    """acdc/transformerblock.py:209-253."""

    VARIANT = 1

    def _make_depthwise_pair(self, dim):
        if dim in (32, 64):
            kd, dd, pd, k0, p0 = (5, 7, 7), (3, 3, 3), (6, 9, 9), 5, 2
        elif dim == 128:
            kd, dd, pd, k0, p0 = (3, 5, 5), (1, 3, 3), (1, 6, 6), 5, 2
        elif dim == 256:
            kd, dd, pd, k0, p0 = 3, 1, 1, 3, 1
        else:
            raise ValueError("Unknown dim: {}".format(dim))      # :231
        return (nn.Conv3d(dim, dim, kernel_size=k0, padding=p0, groups=dim),
                nn.Conv3d(dim, dim, kernel_size=kd, stride=1, padding=pd, groups=dim, dilation=dd))


class LKA_Attention3d_deform(_tb.LKA_Attention3d_deform):
    """acdc/transformerblock.py:256-275."""
    GATING_UNIT = LKA3d_deform


class TransformerBlock_3D_single_deform_LKA(_tb.TransformerBlock_3D_single_deform_LKA):
    """acdc/transformerblock.py:140-206."""
    EPA_BLOCK = LKA_Attention3d_deform
