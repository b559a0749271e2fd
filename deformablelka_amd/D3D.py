"""Drop-in for the reference's pybind11 module ``D3D`` (3D/dcn/src/vision.cpp:4-7): the same two functions with
the same 19/20 positional arguments, so the reference's own ``deform_conv_func.py`` (3D/dcn/functions/
deform_conv_func.py:13,26,43) works unmodified when this module is importable as ``D3D``
(``deformablelka_amd.install_reference_aliases()`` registers it)."""
from . import ops


def deform_conv_forward(input, weight, bias, offset, kernel_d, kernel_h, kernel_w, stride_d, stride_h, stride_w,
                        pad_d, pad_h, pad_w, dilation_d, dilation_h, dilation_w, group, deformable_group, im2col_step):
    """3D/dcn/src/deform_conv.h:10-47 -> at::Tensor output [B, Cout, Do, Ho, Wo]."""
    return ops.deform_conv3d_forward(input, weight, bias, offset, (kernel_d, kernel_h, kernel_w),
                                     (stride_d, stride_h, stride_w), (pad_d, pad_h, pad_w),
                                     (dilation_d, dilation_h, dilation_w), group, deformable_group, im2col_step)


def deform_conv_backward(input, weight, bias, offset, grad_output, kernel_d, kernel_h, kernel_w, stride_d, stride_h,
                         stride_w, pad_d, pad_h, pad_w, dilation_d, dilation_h, dilation_w, group, deformable_group,
                         im2col_step):
    """3D/dcn/src/deform_conv.h:49-91 -> [grad_input, grad_offset, grad_weight, grad_bias]."""
    return list(ops.deform_conv3d_backward(input, weight, bias, offset, grad_output, (kernel_d, kernel_h, kernel_w),
                                           (stride_d, stride_h, stride_w), (pad_d, pad_h, pad_w),
                                           (dilation_d, dilation_h, dilation_w), group, deformable_group, im2col_step))
