"""nn.Module surface of the 3-D deformable convolution — constructor / forward signatures, parameter names,
initialisers and quirks of 3D/dcn/modules/deform_conv.py:15-326 and its in-package copy
3D/d_lka_former/network_architecture/synapse/deform_conv.py (which adds ``DeformConvPack_Depth``, :113-158).

Kept quirks: Q2 bias parameter always exists and is always added (``bias=False`` only freezes it);
Q3 ``conv_offset`` ignores ``dilation``; Q4 ``lr_mult`` is stored, never consumed; Q5 zero-initialised
``conv_offset``; Q7 the ``_d`` variants scatter ``len(dimension)*K`` predicted channels into the 3K layout.
The python-loop channel fills of the reference (deform_conv.py:190-228) are replaced by one strided copy —
same result, no per-tap kernel launches — and the hard-coded 81 is generalised to 3*dg*K (equal for K=27).
"""
import math

import torch
from torch import nn
from torch.nn import init
from torch.nn.modules.utils import _triple

from ..functions.deform_conv_func import DeformConvFunction


class DeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1, deformable_groups=1,
                 im2col_step=64, bias=True):
        super().__init__()
        if in_channels % groups != 0:
            raise ValueError('in_channels {} must be divisible by groups {}'.format(in_channels, groups))
        if out_channels % groups != 0:
            raise ValueError('out_channels {} must be divisible by groups {}'.format(out_channels, groups))
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _triple(kernel_size)
        self.stride = _triple(stride)
        self.padding = _triple(padding)
        self.dilation = _triple(dilation)
        self.groups = groups
        self.deformable_groups = deformable_groups
        self.im2col_step = im2col_step
        self.use_bias = bias
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        self.bias = nn.Parameter(torch.Tensor(out_channels))
        self.reset_parameters()
        if not self.use_bias:
            self.bias.requires_grad = False

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def _apply_op(self, input, offset):
        return DeformConvFunction.apply(input, offset, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                        self.groups, self.deformable_groups, self.im2col_step)

    def forward(self, input, offset):
        assert 3 * self.deformable_groups * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2] == \
            offset.shape[1]
        return self._apply_op(input, offset)


_DeformConv = DeformConvFunction.apply


class DeformConvPack(DeformConv):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1, deformable_groups=1,
                 im2col_step=64, bias=True, lr_mult=0.1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups,
                         im2col_step, bias)
        out_channels = self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        self.conv_offset = nn.Conv3d(self.in_channels, out_channels, kernel_size=self.kernel_size, stride=self.stride,
                                     padding=self.padding, bias=True)
        self.conv_offset.lr_mult = lr_mult
        self.init_offset()

    def init_offset(self):
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def _predict_offset(self, input):
        from ..nn_ops import conv3d  # hand-written HIP conv with autograd, replaces the cuDNN nn.Conv3d call
        c = self.conv_offset
        return conv3d(input, c.weight, c.bias, c.stride, c.padding, c.dilation, c.groups)

    def forward(self, input):
        offset = self._predict_offset(input)
        return self._apply_op(input, offset)


class DeformConvPack_experimental(DeformConv):
    """3D/dcn/modules/deform_conv.py:103-139."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1, deformable_groups=1,
                 im2col_step=64, bias=True, lr_mult=0.1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups,
                         im2col_step, bias)
        self.out_channels = self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        self.channel_adjust = nn.Conv3d(self.in_channels, self.out_channels, kernel_size=(1, 1, 1))
        self.conv_offset = nn.Conv3d(self.out_channels, self.out_channels, kernel_size=self.kernel_size, stride=self.stride,
                                     padding=self.padding, groups=self.out_channels, bias=True)
        self.conv_offset.lr_mult = lr_mult
        self.init_offset()

    def init_offset(self):
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, input):
        from ..nn_ops import conv3d
        a, c = self.channel_adjust, self.conv_offset
        adj_input = conv3d(input, a.weight, a.bias, a.stride, a.padding, a.dilation, a.groups)
        offset = conv3d(adj_input, c.weight, c.bias, c.stride, c.padding, c.dilation, c.groups)
        return self._apply_op(input, offset)


class DeformConvPack_Depth(DeformConv):
    """3D/d_lka_former/network_architecture/synapse/deform_conv.py:113-158: depthwise conv_offset then conv_1x1 -> 3K."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1, deformable_groups=1,
                 im2col_step=64, bias=True, lr_mult=0.1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups,
                         im2col_step, bias)
        out_channels = self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        self.conv_1x1 = nn.Conv3d(in_channels=self.in_channels, out_channels=out_channels, kernel_size=1)
        self.conv_offset = nn.Conv3d(in_channels, in_channels, kernel_size=self.kernel_size, stride=self.stride,
                                     padding=self.padding, groups=in_channels, bias=True)
        self.conv_offset.lr_mult = lr_mult
        self.init_offset()

    def init_offset(self):
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, input):
        from ..nn_ops import conv3d
        input = input.contiguous()
        c, p = self.conv_offset, self.conv_1x1
        offset = conv3d(input, c.weight, c.bias, c.stride, c.padding, c.dilation, c.groups)
        offset = conv3d(offset, p.weight, p.bias, p.stride, p.padding, p.dilation, p.groups)
        return self._apply_op(input, offset)


def _expand_partial_offsets(temp, dimension, n_taps):
    """Scatter ``len(dimension)`` predicted offset channels per tap into the (d,h,w)-per-tap layout with zeros on
    the frozen axes.  Equivalent to the python loops at 3D/dcn/modules/deform_conv.py:190-228 (T->d, H->h, W->w)."""
    axes = [a for a, name in enumerate("THW") if name in dimension]
    L = len(axes)
    B, c = temp.shape[:2]
    assert c == L * n_taps, (c, L, n_taps)
    if L == 3:
        return temp
    sp = temp.shape[2:]
    offset = temp.new_zeros((B, n_taps, 3) + tuple(sp))
    t = temp.reshape((B, n_taps, L) + tuple(sp))
    for j, a in enumerate(axes):
        offset[:, :, a] = t[:, :, j]
    return offset.reshape((B, n_taps * 3) + tuple(sp))


class DeformConv_d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dimension='THW', dilation=1, groups=1,
                 deformable_groups=1, im2col_step=64, bias=True):
        super().__init__()
        if in_channels % groups != 0:
            raise ValueError('in_channels {} must be divisible by groups {}'.format(in_channels, groups))
        if out_channels % groups != 0:
            raise ValueError('out_channels {} must be divisible by groups {}'.format(out_channels, groups))
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _triple(kernel_size)
        self.stride = _triple(stride)
        self.padding = _triple(padding)
        self.dilation = _triple(dilation)
        self.dimension = dimension
        self.length = len(dimension)
        self.groups = groups
        self.deformable_groups = deformable_groups
        self.im2col_step = im2col_step
        self.use_bias = bias
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        self.bias = nn.Parameter(torch.Tensor(out_channels))
        self.reset_parameters()
        if not self.use_bias:
            self.bias.requires_grad = False

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def _n_taps(self):
        return self.deformable_groups * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]

    def forward(self, input, temp):
        offset = _expand_partial_offsets(temp, self.dimension, self._n_taps())
        return DeformConvFunction.apply(input, offset, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                        self.groups, self.deformable_groups, self.im2col_step)


class DeformConvPack_d(DeformConv_d):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dimension='THW', dilation=1, groups=1,
                 deformable_groups=1, im2col_step=64, bias=True, lr_mult=0.1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dimension, dilation, groups,
                         deformable_groups, im2col_step, bias)
        self.dimension = dimension
        self.length = len(dimension)
        out_channels = self.deformable_groups * self.length * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        self.conv_offset = nn.Conv3d(self.in_channels, out_channels, kernel_size=self.kernel_size, stride=self.stride,
                                     padding=self.padding, bias=True)
        self.conv_offset.lr_mult = lr_mult
        self.init_offset()

    def init_offset(self):
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, input):
        from ..nn_ops import conv3d
        c = self.conv_offset
        temp = conv3d(input, c.weight, c.bias, c.stride, c.padding, c.dilation, c.groups)
        return super().forward(input, temp)
