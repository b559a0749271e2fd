#!/usr/bin/env python
"""The 2-D depthwise deformable conv's forward call alone at the three decoder shapes (B = 24): device-event timing.  usage: python scripts/time_ddw2d_fwd.py [lib.so]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deformablelka_amd import ops, _lib as L
if len(sys.argv) > 1 and sys.argv[1] != "-":
    cd = ctypes.CDLL(os.path.join(ROOT, sys.argv[1]))
    for name, (rs, args) in L.SIGNATURES.items():
        if hasattr(cd, name):
            fn = getattr(cd, name); fn.restype = rs; fn.argtypes = args
    L._lib = cd
dev = "cuda:0"
torch.manual_seed(0)
for C, n in ((96, 56), (192, 28), (384, 14)):
    for k, pad, dil in ((5, 2, 1), (7, 9, 3)):
        x = torch.randn(24, n, n, C, device=dev)
        off = torch.randn(24, 2 * k * k, n, n, device=dev) * 0.3
        w = torch.randn(C, 1, k, k, device=dev) * 0.1
        for _ in range(3):
            y = ops.deform_dwconv2d_forward_cl(x, off, w, pad, dil)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = ops.deform_dwconv2d_forward_cl(x, off, w, pad, dil)
        e1.record()
        torch.cuda.synchronize()
        print(f"C {C} n {n} k {k}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per call (incl. weight preparation)  checksum {float(y.double().sum()):.6f}")
