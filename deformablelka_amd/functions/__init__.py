from .deform_conv_func import DeformConvFunction  # noqa: F401
