#!/usr/bin/env python
"""TransformerBlock_3D_single_deform_LKA fwd+bwd of one stage, repeated (for rocprofv3 --kernel-trace --stats).
Usage: python scripts/prof_tblock.py --stage 2"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import deformablelka_amd as dk
from deformablelka_amd.stack import SYNAPSE_STAGES
ap = argparse.ArgumentParser()
ap.add_argument("--stage", type=int, default=2)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
C, (H, W, D), n = SYNAPSE_STAGES[a.stage]
m = dk.TransformerBlock_3D_single_deform_LKA(H * W * D, C, C, 4, dropout_rate=0.1, pos_embed=True).to("cuda:0")
x = torch.randn(2, C, H, W, D, device="cuda:0").permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3).requires_grad_(True)
gy = torch.randn(2, H, W, D, C, device="cuda:0").permute(0, 4, 1, 2, 3)
for _ in range(a.iters):
    m(x).backward(gy)
torch.cuda.synchronize()
