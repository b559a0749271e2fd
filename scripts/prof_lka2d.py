#!/usr/bin/env python
"""One 2-D D-LKA attention block of a decoder shape (C, H = W), bf16 activations, B = 24: forward + backward, repeated — the workload of the rocprofv3 passes of
scripts/pmc_lka2d.sh (HBM bytes per kernel of the 2-D step) and of kernel-trace tables.  usage: python scripts/prof_lka2d.py --C 96 --hw 56 [--dtype bf16] [--iters 3]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deformablelka_amd as dk
from deformablelka_amd.init_utils import randomize_offset_nets
ap = argparse.ArgumentParser()
ap.add_argument("--C", type=int, default=96)
ap.add_argument("--hw", type=int, default=56)
ap.add_argument("--B", type=int, default=24)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
dev = "cuda:0"
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
torch.manual_seed(0)
m = dk.deformable_LKA_Attention(a.C).to(dev)
randomize_offset_nets(m, 0.02)
x = torch.randn(a.B, a.C, a.hw, a.hw, device=dev).to(dt).requires_grad_(True)
gy = torch.randn(a.B, a.C, a.hw, a.hw, device=dev).to(dt)
for _ in range(a.iters):
    m(x).backward(gy)
torch.cuda.synchronize()
print("done", a.C, a.hw, a.dtype)
