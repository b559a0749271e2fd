#!/usr/bin/env python
"""Run selected kernel-level ops of one stage repeatedly (for rocprofv3 --kernel-trace --stats / --pmc).
Usage: python scripts/prof_op.py --C 32 --N 32 --ops deform_bwd_input,deform_fwd [--iters 10]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402




# (stand-alone operator calls through the channels-last entry points; bench.py's roofline no longer uses this — it traces the step's own kernels)
def time_ops(B, C, N, dtype, iters=10, only=None):
    """HIP-event timing (torch.cuda.Event on the current stream == the stream the C-ABI launches on) of every
    kernel-level op of one token-layout block, through the channels-last entry points the block itself uses."""
    from ctypes import byref
    from deformablelka_amd import _lib as L
    lib = L.get_lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    dt = L.DLKA_F32 if dtype == torch.float32 else L.DLKA_BF16
    st = L.stream_ptr(torch.empty(1, device=dev))
    g = torch.Generator().manual_seed(0)
    bf16 = dtype == torch.bfloat16
    mk = lambda *s: torch.randn(*s, generator=g).to(dev, dtype)              # activations in the run's storage type
    mkf = lambda *s: torch.randn(*s, generator=g).to(dev, torch.float32)     # offsets, parameters: always fp32
    x, go = mk(B, N, N, N, C), mk(B, N, N, N, C)                     # channels-last activations
    off, goff = mkf(B, 81, N, N, N), mkf(B, 81, N, N, N)             # planar offsets
    out, out_off = torch.empty(B, N, N, N, C, dtype=torch.float32, device=dev), torch.empty_like(off)   # (fp32-sized: also serves as fp32 grad_x)
    w_pw, w5, w7 = mkf(C, C, 1, 1, 1), mkf(C, 1, 5, 5, 5), mkf(C, 1, 7, 7, 7)
    w_off, w_dc = mkf(81, C, 3, 3, 3) * 0.02, mkf(C, C, 3, 3, 3) * 0.03
    b_c, b_81 = mkf(C), mkf(81)
    if bf16 and only is None:   # the per-op entry points carry bf16 activations for the deformable conv only (the dominant ops)
        only = ("deform_fwd", "deform_bwd_input", "deform_bwd_offset", "deform_bwd_weight")
    gw_pw, gw5, gw7, gw_off, gw_dc = (torch.empty_like(t) for t in (w_pw, w5, w7, w_off, w_dc))

    def geom(cout, k, p, d, grp):
        return L.ConvGeom(B, C, N, N, N, cout, k, k, k, 1, 1, 1, p, p, p, d, d, d, grp, 1, 64)

    G = {"pw": geom(C, 1, 0, 1, 1), "dw5": geom(C, 5, 2, 1, C), "dw7": geom(C, 7, 9, 3, C), "off": geom(81, 3, 1, 1, 1),
         "dcn": geom(C, 3, 1, 1, 1)}
    wsb = max([lib.dlka_conv3d_cl_workspace(byref(v), dt, 1) for v in G.values()] +
              [lib.dlka_deform_conv3d_cl_workspace(byref(G["dcn"]), dt, 1)])
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    P = L.ptr
    N0 = None

    def conv_fwd(key, w, b, inp, o, planar=0):
        return lambda: lib.dlka_conv3d_forward_cl(P(inp), P(w), P(b), P(o), planar, P(ws), wsb, byref(G[key]), dt, st)

    def conv_bwd(key, w, inp, gout, gx, gw, planar=0):
        return lambda: lib.dlka_conv3d_backward_cl(P(inp), P(w), P(gout), planar, P(gx), P(gw), P(N0), P(ws), wsb, byref(G[key]), dt, st)

    ops = {
        "pointwise_fwd": conv_fwd("pw", w_pw, b_c, x, out),
        "dw5_fwd": conv_fwd("dw5", w5, b_c, x, out),
        "dw7_fwd": conv_fwd("dw7", w7, b_c, x, out),
        "offset_conv_fwd": conv_fwd("off", w_off, b_81, x, out_off, 1),
        "deform_fwd": lambda: lib.dlka_deform_conv3d_forward_cl(P(x), P(off), P(w_dc), P(b_c), P(out), P(ws), wsb, byref(G["dcn"]), dt, st),
        "pointwise_bwd_data": conv_bwd("pw", w_pw, x, go, out, None),
        "pointwise_bwd_weight": conv_bwd("pw", w_pw, x, go, None, gw_pw),
        "dw5_bwd_data": conv_bwd("dw5", w5, x, go, out, None),
        "dw5_bwd_weight": conv_bwd("dw5", w5, x, go, None, gw5),
        "dw7_bwd_data": conv_bwd("dw7", w7, x, go, out, None),
        "dw7_bwd_weight": conv_bwd("dw7", w7, x, go, None, gw7),
        "offset_conv_bwd_data": conv_bwd("off", w_off, x, goff, out, None, 1),
        "offset_conv_bwd_weight": conv_bwd("off", w_off, x, goff, None, gw_off, 1),
        "deform_bwd_input": lambda: lib.dlka_deform_conv3d_backward_cl(P(x), P(off), P(w_dc), P(go), P(out), P(N0), P(N0), P(N0), P(ws), wsb, byref(G["dcn"]), dt, st),
        "deform_bwd_offset": lambda: lib.dlka_deform_conv3d_backward_cl(P(x), P(off), P(w_dc), P(go), P(N0), P(out_off), P(N0), P(N0), P(ws), wsb, byref(G["dcn"]), dt, st),
        "deform_bwd_weight": lambda: lib.dlka_deform_conv3d_backward_cl(P(x), P(off), P(w_dc), P(go), P(N0), P(N0), P(gw_dc), P(N0), P(ws), wsb, byref(G["dcn"]), dt, st),
    }
    res = {}
    for name, fn in ops.items():
        if only is not None and name not in only:
            continue
        rc = fn()
        if rc != 0:
            raise RuntimeError(f"{name}: dlka status {rc}")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / iters  # ms per launch
    return res



ap = argparse.ArgumentParser()
ap.add_argument("--C", type=int, default=32)
ap.add_argument("--N", type=int, default=32)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--ops", default="deform_bwd_input")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
torch.cuda.set_device(0)
res = time_ops(a.batch, a.C, a.N, torch.float32, iters=a.iters, only=set(a.ops.split(",")))
for k, v in res.items():
    print(f"{k}: {v:.4f} ms")
