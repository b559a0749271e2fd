#!/usr/bin/env python
"""grad_input of the deformable conv: fixed-point packed window (DLKA_GX_FIXED=1) vs the fp64 window, stage-0 shape (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deformablelka_amd import ops
torch.manual_seed(0)
dev = "cuda:0"
for C, n in ((32, 32), (64, 16)):
    x = torch.randn(2, n, n, n, C, device=dev)
    off = torch.randn(2, 81, n, n, n, device=dev)
    w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
    go = torch.randn(2, n, n, n, C, device=dev)
    os.environ.pop("DLKA_GX_FIXED", None)
    gi0, _, _, _ = ops.deform_conv3d_backward_cl(x, off, w, go)
    gi0b, _, _, _ = ops.deform_conv3d_backward_cl(x, off, w, go)
    os.environ["DLKA_GX_FIXED"] = "1"
    gi1, _, _, _ = ops.deform_conv3d_backward_cl(x, off, w, go)
    os.environ.pop("DLKA_GX_FIXED", None)
    ref = gi0.double()
    print(f"C={C} {n}^3: max|gi| {ref.abs().max().item():.3f}  fp64-window run-to-run max abs diff {(gi0b.double() - ref).abs().max().item():.3e}  "
          f"fixed vs fp64: max abs {(gi1.double() - ref).abs().max().item():.3e}  rel-to-max {((gi1.double() - ref).abs().max() / ref.abs().max()).item():.3e}  "
          f"rms rel {(((gi1.double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()).item():.3e}")
