#!/usr/bin/env python
"""Offset statistics + per-block fwd/bwd HIP-event times inside the full 21-block stack (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ctypes import byref
from deformablelka_amd import _lib as L
from deformablelka_amd.stack import DLKABlockStack

torch.cuda.set_device(0)
st = DLKABlockStack(2, device="cuda:0", seed=1234)
st.forward_backward()
torch.cuda.synchronize()
a256 = lambda n: (n + 255) & ~255
for i, blk in enumerate(st.blocks):
    H, W, D = blk.dims
    N = H * W * D
    E = st.B * blk.C * N
    Off = st.B * 81 * N
    o = 4 * a256(E * 4)
    off = blk.saved[o:o + Off * 4].view(torch.float32)
    x = blk.x.float()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    stream = st._stream()
    ev[0].record()
    for _ in range(5):
        rc = st.lib.dlka_lka3d_attention_tokens_forward(L.ptr(blk.x), byref(blk.pstruct), L.ptr(blk.y), L.ptr(blk.saved), blk.saved_bytes, L.ptr(st.ws), st.ws_bytes, st.B, blk.C, H, W, D, st.dt, stream)
    ev[1].record()
    for _ in range(5):
        rc = st.lib.dlka_lka3d_attention_tokens_backward(L.ptr(blk.x), byref(blk.pstruct), L.ptr(blk.gy), L.ptr(blk.saved), blk.saved_bytes, L.ptr(blk.gx), byref(blk.gstruct), L.ptr(st.ws), st.ws_bytes, st.B, blk.C, H, W, D, st.dt, stream)
    ev[2].record()
    torch.cuda.synchronize()
    print(f"block {i:2d} C={blk.C:3d} N={N:6d}: x std {x.std().item():8.3f}  offset std {off.std().item():7.3f} max {off.abs().max().item():8.2f}  frac|off|>2 {(off.abs() > 2).float().mean().item():.3f}  fwd {ev[0].elapsed_time(ev[1]) / 5:.3f} ms  bwd {ev[1].elapsed_time(ev[2]) / 5:.3f} ms  gy std {blk.gy.float().std().item():.3g}")
