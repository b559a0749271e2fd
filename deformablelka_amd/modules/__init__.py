from .deform_conv import (DeformConv, DeformConvPack, DeformConvPack_experimental, DeformConvPack_Depth,  # noqa: F401
                          DeformConv_d, DeformConvPack_d, _DeformConv)
