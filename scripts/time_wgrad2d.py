#!/usr/bin/env python
"""The 2-D offset nets' backward (data + weight gradient) alone, padded against unpadded weight-gradient kernels (DLKA_WGRAD_PAD, read per call), fp32, B = 24,
ONE stream, per-kernel times from the library's launch trace.  usage: python scripts/time_wgrad2d.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import byref, c_float, create_string_buffer
from deformablelka_amd import ops, _lib as L
lib = L.get_lib()
dev = "cuda:0"
st = torch.cuda.current_stream(torch.device(dev)).cuda_stream
for C, hw in ((384, 14), (192, 28), (96, 56)):
    for k, pad, dil, cout in ((7, 9, 3, 98), (5, 2, 1, 50)):
        x = torch.randn(24, 1, hw, hw, C, device=dev)
        w = torch.randn(cout, C, 1, k, k, device=dev) * 0.05
        go = torch.randn(24, cout, 1, hw, hw, device=dev)
        row = []
        for mode in ("1", "0"):
            os.environ["DLKA_WGRAD_PAD"] = mode
            for _ in range(2):
                ops.conv3d_backward_cl(x, w, go, (0, pad, pad), (1, dil, dil), 1, grad_out_planar=True)
            torch.cuda.synchronize()
            L.check(lib.dlka_trace_start(512, st), "start")
            for _ in range(3):
                ops.conv3d_backward_cl(x, w, go, (0, pad, pad), (1, dil, dil), 1, grad_out_planar=True)
            L.check(lib.dlka_trace_stop(), "stop")
            buf, ms, acc = create_string_buffer(512), c_float(), {}
            for i in range(lib.dlka_trace_count()):
                L.check(lib.dlka_trace_get(i, buf, 512, byref(ms)), "get")
                n = buf.value.decode().replace("void dlka::", "").split("(")[0]
                if "wgrad_dense" in n or "pad_copy" in n:
                    acc[n] = acc.get(n, 0.0) + ms.value / 3
            row.append(acc)
        fl = 2 * 24 * hw * hw * k * k * 128 * C / 1e9
        print(f"C={C} {hw}x{hw} k={k}: {fl:.1f} GFLOP(padded co)  padded: " + ", ".join(f"{n[:40]} {v*1e3:.0f} us" for n, v in row[0].items()) + "  | unpadded: " + ", ".join(f"{n[:40]} {v*1e3:.0f} us" for n, v in row[1].items()))
os.environ.pop("DLKA_WGRAD_PAD", None)
