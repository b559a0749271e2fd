#!/usr/bin/env python
"""Wave quantisation of the depthwise row kernel: dw 7^3 dilation 3 / 5^3 at (B = 2, C = 32, D = 32, W = 32) for a sweep of H — waves = B D groups(H) 4 / 2.
usage: python scripts/time_dw_quant.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deformablelka_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for (k, p, d) in ((7, 9, 3), (5, 2, 1)):
    w = (torch.randn(32, 1, k, k, k, generator=g) * 0.1).to(dev)
    b = torch.randn(32, generator=g).to(dev)
    for H in (18, 24, 30, 32, 36, 42, 48, 54, 60):
        x = torch.randn(2, 32, H, 32, 32, generator=g).to(dev)
        for _ in range(3):
            y = ops.conv3d_forward_cl(x, w, b, p, d, 32)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            y = ops.conv3d_forward_cl(x, w, b, p, d, 32)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        blk, full = 2 * d, H // (2 * d)
        rem = H - full * blk
        groups = full * d + (rem + 1) // 2 if (full >= 1 and 1 <= rem <= d) else d * ((H + blk - 1) // blk)   # dw_row_groups (cl_dwconv.hip): tail groups since round 6
        waves = 2 * 32 * groups * 4 // 2
        print(f"k {k} H {H:3d}: {us:7.1f} us  waves {waves:5d} = {waves / 1024:.2f} per SIMD   us per 1024 waves {us / (waves / 1024):6.1f}   us per output row-plane {us / H:6.2f}")
