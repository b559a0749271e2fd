"""``UnetResBlock`` with the constructor, forward and ``state_dict`` keys of
3D/d_lka_former/network_architecture/dynunet_block.py:12-80 for the configuration the D-LKA wrapper block uses
(``UnetResBlock(3, C, C, kernel_size=3, stride=1, norm_name="batch")``, transformerblock.py:612).

The reference builds the block from MONAI 0.8 factories (``Convolution(conv_only=True, bias=False)`` -> a Sequential whose
only child is ``conv``; ``get_norm_layer("batch")`` -> ``nn.BatchNorm3d``; ``get_act_layer(("leakyrelu", ...))`` ->
``nn.LeakyReLU(0.01)``), hence the parameter names ``conv1.conv.weight``, ``norm1.weight`` ... kept here.  Inside
``TransformerBlock_3D_single_deform_LKA`` the parameters are consumed by the fused ``dlka_tblock3d_*`` entry points; the
stand-alone ``forward`` below runs the same HIP kernels op by op on an NCDHW tensor.  The down-sampling variant
(``conv3`` / ``norm3``, in_channels != out_channels or stride != 1) belongs to the encoder stem, outside this path.
"""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops


class _Conv(nn.Sequential):
    """MONAI ``Convolution(..., conv_only=True)``: a Sequential with the single child ``conv``."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.add_module("conv", nn.Conv3d(cin, cout, k, stride=1, padding=(k - 1 + 1) // 2, bias=False))  # get_padding, dynunet_block.py:251-262


def bn_eval_stats(bn: nn.BatchNorm3d):
    """{mean, rstd, unused} of a BatchNorm in eval mode (running statistics)."""
    return torch.cat([bn.running_mean, torch.rsqrt(bn.running_var + bn.eps), bn.running_var]).float().contiguous()


def bn_update_running(*pairs):
    """nn.BatchNorm training-mode bookkeeping from the kernel's {mean, rstd, unbiased var}: ``bn_update_running(bn, stats)`` or, for several layers at
    once, ``bn_update_running((bn1, stats1), (bn2, stats2), ...)``.  All layers with a fixed momentum go through ONE ``_foreach_lerp_`` launch (running
    <- running + m (batch - running)) and one ``_foreach_add_`` for the step counters: the two norms of a wrapper block cost 2 launches instead of 10 —
    210 launches per trainer iteration of the full net otherwise, ~4 us each."""
    if len(pairs) == 2 and isinstance(pairs[0], nn.Module):
        pairs = (pairs,)
    by_m, counters = {}, []
    with torch.no_grad():
        for bn, stats in pairs:
            if not bn.track_running_stats or bn.running_mean is None:
                continue
            C = bn.num_features
            if bn.momentum is None:   # cumulative average: the factor depends on the counter (a device value)
                bn.num_batches_tracked += 1
                m = 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1 - m).add_(stats[:C], alpha=m)
                bn.running_var.mul_(1 - m).add_(stats[2 * C:3 * C], alpha=m)
                continue
            counters.append(bn.num_batches_tracked)
            dst, src = by_m.setdefault(float(bn.momentum), ([], []))
            dst += [bn.running_mean, bn.running_var]
            src += [stats[:C], stats[2 * C:3 * C]]
        if counters:
            torch._foreach_add_(counters, 1)
        for m, (dst, src) in by_m.items():
            torch._foreach_lerp_(dst, src, m)


class _UnetResBlockFn(Function):
    @staticmethod
    def forward(ctx, inp, training, st1, st2, eps, w1, g1, b1, w2, g2, b2):
        x = ops.ncdhw_to_ndhwc(inp)
        c1 = ops.conv3d_forward_cl(x, w1, None, padding=1)
        a1 = ops.batchnorm_cl_forward(c1, None, g1, b1, st1, training, eps[0])
        c2 = ops.conv3d_forward_cl(a1, w2, None, padding=1)
        r = ops.batchnorm_cl_forward(c2, x, g2, b2, st2, training, eps[1])
        ctx.training = training
        ctx.save_for_backward(x, c1, a1, c2, r, st1, st2, w1, g1, w2, g2)
        return ops.ndhwc_to_ncdhw(r)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, c1, a1, c2, r, st1, st2, w1, g1, w2, g2 = ctx.saved_tensors
        g = ops.ncdhw_to_ndhwc(gy)
        g_c2, g_res, gg2, gb2 = ops.batchnorm_cl_backward(g, c2, r, g2, st2, ctx.training, with_res=True)
        g_a1, gw2, _ = ops.conv3d_backward_cl(a1, w2, g_c2, padding=1)
        g_c1, _, gg1, gb1 = ops.batchnorm_cl_backward(g_a1, c1, a1, g1, st1, ctx.training)
        g_x, gw1, _ = ops.conv3d_backward_cl(x, w1, g_c1, padding=1)
        g_x = ops.scale_residual_forward(g_res, g_x, torch.ones_like(g1))   # g_res + g_x on the HIP side
        return ops.ndhwc_to_ncdhw(g_x), None, None, None, None, gw1, gg1, gb1, gw2, gg2, gb2


class UnetResBlock(nn.Module):
    """dynunet_block.py:12-80 (stride 1, in_channels == out_channels, batch norm, LeakyReLU 0.01)."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name,
                 act_name=("leakyrelu", {"inplace": True, "negative_slope": 0.01}), dropout=None):
        super().__init__()
        norm = norm_name[0] if isinstance(norm_name, (tuple, list)) else norm_name
        act = act_name[0] if isinstance(act_name, (tuple, list)) else act_name
        slope = act_name[1].get("negative_slope", 0.01) if isinstance(act_name, (tuple, list)) and len(act_name) > 1 else 0.01
        if (spatial_dims != 3 or in_channels != out_channels or kernel_size != 3 or stride != 1 or str(norm).lower() != "batch"
                or str(act).lower() != "leakyrelu" or slope != 0.01 or dropout is not None):
            raise NotImplementedError("deformablelka_amd.UnetResBlock covers the D-LKA wrapper's configuration only: "
                                      "UnetResBlock(3, C, C, kernel_size=3, stride=1, norm_name='batch')")
        self.conv1 = _Conv(in_channels, out_channels, kernel_size)
        self.conv2 = _Conv(out_channels, out_channels, kernel_size)
        self.lrelu = nn.LeakyReLU(negative_slope=0.01, inplace=True)
        self.norm1 = nn.BatchNorm3d(out_channels)
        self.norm2 = nn.BatchNorm3d(out_channels)
        self.downsample = False

    def forward(self, inp):
        C = self.norm1.num_features
        # nn.BatchNorm3d keys on ITS OWN .training (the "freeze BN" pattern calls bn.eval() inside a module left in train mode);
        # TransformerBlock_3D_single_deform_LKA does the same.  One fused Function serves both norms, so they must agree.
        training = self.norm1.training
        if self.norm2.training != training:
            raise NotImplementedError("UnetResBlock: norm1 and norm2 must be in the same mode (both .train() or both .eval())")
        if training:
            st1 = torch.empty(3 * C, dtype=torch.float32, device=inp.device)
            st2 = torch.empty_like(st1)
        else:
            st1, st2 = bn_eval_stats(self.norm1), bn_eval_stats(self.norm2)
        out = _UnetResBlockFn.apply(inp, training, st1, st2, (self.norm1.eps, self.norm2.eps), self.conv1.conv.weight, self.norm1.weight,
                                    self.norm1.bias, self.conv2.conv.weight, self.norm2.weight, self.norm2.bias)
        if training:
            bn_update_running((self.norm1, st1), (self.norm2, st2))
        return out
