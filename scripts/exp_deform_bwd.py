"""Experiment: how much of cl_deform_bwd is the fp32 global atomics? (grad_x = NULL skips them)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import byref
from deformablelka_amd import _lib as L
lib = L.get_lib()
dev = torch.device("cuda:0")
B, C, N = 2, 32, 32
g = torch.Generator().manual_seed(0)
mk = lambda *s: torch.randn(*s, generator=g).to(dev)
x, go = mk(B, N, N, N, C), mk(B, N, N, N, C)
w_dc = mk(C, C, 3, 3, 3) * 0.03
out = torch.empty_like(x)
geom = L.ConvGeom(B, C, N, N, N, C, 3, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 64)
wsb = lib.dlka_deform_conv3d_cl_workspace(byref(geom), 0, 1)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
st = L.stream_ptr(x)
P = L.ptr
for sigma in (0.0, 0.25, 1.0, 3.0):
    off = mk(B, 81, N, N, N) * sigma
    out_off = torch.empty_like(off)
    for name, gx, goff in (("gx+goff", out, out_off), ("goff only", None, out_off), ("gx only", out, None)):
        fn = lambda: lib.dlka_deform_conv3d_backward_cl(P(x), P(off), P(w_dc), P(go), P(gx), P(goff), P(None), P(None), P(ws), wsb, byref(geom), 0, st)
        assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"sigma {sigma:4.2f} {name:10s} {e0.elapsed_time(e1) / 10:8.4f} ms")
