#!/usr/bin/env python
"""Predicted sampling offsets of every D-LKA block INSIDE the wrapper-block stack and the assembled net, as bench.py initialises them (the timing runs want
offsets of ~1.0 voxel std; the calibration table `stack._offset_std_for` was measured on the bare block fed x ~ N(0, 1)).  Prints per block: width, volume, std of the
predicted offsets, share of |offset| > 2.5 voxels (what leaves grad_input's LDS window + halo).   usage: python scripts/offset_stats_net.py [tblock|fullnet] [gain]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deformablelka_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else "fullnet"
dev = torch.device("cuda", 0)
rec = []
orig = ops.tblock3d_forward


def spy(x, x_planar, tparams, lka_params, drop_mask, training, bn_stats, dims, ln_eps=1e-5, bn_eps=1e-5, variant=0, lka_bf16=False):
    y, saved = orig(x, x_planar, tparams, lka_params, drop_mask, training, bn_stats, dims, ln_eps, bn_eps, variant, lka_bf16)
    B, C = int(x.shape[0]), int(lka_params[0].shape[0])
    H, W, D = (int(v) for v in dims)
    off = ops.tblock3d_saved_offsets(saved, B, C, (H, W, D), variant, lka_bf16)
    rec.append((C, (H, W, D), float(off.std()), float((off.abs() > 2.5).float().mean())))
    return y, saved


ops.tblock3d_forward = spy
import deformablelka_amd.transformerblock as tb
tb.ops = ops
if which == "fullnet":
    from deformablelka_amd import training
    from deformablelka_amd.stack import _offset_std_for
    torch.manual_seed(0)
    net = training.initialize_network(1, 14, (64, 128, 128), device=dev)
    with torch.no_grad():
        for blk in net.dlka_blocks():
            w = blk.epa_block.spatial_gating_unit.deform_conv.conv_offset.weight
            w.normal_(0, _offset_std_for(w.shape[1]))
    x = torch.randn(2, 1, 64, 128, 128, device=dev)
    net.train()
    with torch.no_grad():
        net(x)
else:
    import deformablelka_amd as dk
    from deformablelka_amd.stack import SYNAPSE_STAGES, CHAIN, _offset_std_for
    torch.manual_seed(0)
    for C, (H, W, D), n in SYNAPSE_STAGES:
        for c0 in range(0, n, CHAIN):
            y = torch.randn(2, H, W, D, C, device=dev).permute(0, 4, 1, 2, 3)
            for _ in range(min(CHAIN, n - c0)):
                m = dk.TransformerBlock_3D_single_deform_LKA(H * W * D, C, C, 4, dropout_rate=0.1, pos_embed=True)
                with torch.no_grad():
                    m.epa_block.spatial_gating_unit.deform_conv.conv_offset.weight.normal_(0, _offset_std_for(C))
                m.keep_channels_last = True
                m = m.to(dev)
                with torch.no_grad():
                    y = m(y)
torch.cuda.synchronize()
for i, (C, dims, sd, far) in enumerate(rec):
    print("block %2d  C=%3d %s  offset std %.3f voxels, |off| > 2.5: %.2f %%" % (i, C, "x".join(map(str, dims)), sd, far * 100))
