"""One shape of the 2-D depthwise deformable conv backward (cl_ddw2d.hip) for profiling: (C, H = W, B = 24), k = 5 (pad 2) and 7 (dil 3, pad 9),
offsets ~ N(0, std).  usage: python scripts/prof_ddw2d.py [--C 96 --hw 56 --std 1.0 --reps 5 --dtype f32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deformablelka_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--C", type=int, default=96)
ap.add_argument("--hw", type=int, default=56)
ap.add_argument("--B", type=int, default=24)
ap.add_argument("--std", type=float, default=1.0)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--dtype", default="f32")
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
dt = torch.float32 if a.dtype == "f32" else torch.bfloat16
x = torch.randn(a.B, a.hw, a.hw, a.C, device=dev).to(dt)
g = torch.randn(a.B, a.hw, a.hw, a.C, device=dev).to(dt)
for k, pad, dil in ((5, 2, 1), (7, 9, 3)):
    off = torch.randn(a.B, 2 * k * k, a.hw, a.hw, device=dev) * a.std
    w = torch.randn(a.C, 1, k, k, device=dev) * 0.1
    for _ in range(2):
        ops.deform_dwconv2d_backward_cl(x, off, w, g, pad, dil)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        ops.deform_dwconv2d_backward_cl(x, off, w, g, pad, dil)
    e1.record()
    torch.cuda.synchronize()
    print(f"k={k} C={a.C} {a.hw}^2 B={a.B} {a.dtype} std={a.std}: {e0.elapsed_time(e1) / a.reps * 1e3:.1f} us per backward call (bwd + fold + grad_input)")
