#!/usr/bin/env python
"""The 2-D depthwise deformable conv's backward call alone (grad_offset + weight-gradient partials, fold, grad_input [+ far]), ONE stream, device-event timing per call and the
library's launch trace per kernel (valid here: everything is on one stream).  usage: python scripts/time_ddw2d_gx.py [lib.so]   (bf16 and fp32, offsets from N(0, std))"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ctypes import byref, c_float, create_string_buffer
from deformablelka_amd import ops, _lib as L
if len(sys.argv) > 1 and sys.argv[1] != "-":
    cd = ctypes.CDLL(os.path.join(ROOT, sys.argv[1]))
    for name, (rs, args) in L.SIGNATURES.items():
        if hasattr(cd, name):
            fn = getattr(cd, name); fn.restype = rs; fn.argtypes = args
    L._lib = cd
lib = L.get_lib()
dev = "cuda:0"
st = torch.cuda.current_stream(torch.device(dev)).cuda_stream
torch.manual_seed(0)
SHAPES = [tuple(int(v) for v in sh.split("x")) for sh in os.environ.get("GX_SHAPES", "96x56").split(",")]   # C x n, e.g. GX_SHAPES=96x56,192x28,384x14
for dt in ((torch.bfloat16,) if os.environ.get("GX_BF16") else (torch.float32,)):
  for C, n in SHAPES:
    for std in (0.3,):
        for k, pad, dil in ((5, 2, 1), (7, 9, 3)):
            x = torch.randn(24, n, n, C, device=dev).to(dt)
            g = torch.randn(24, n, n, C, device=dev).to(dt)
            off = torch.randn(24, 2 * k * k, n, n, device=dev) * std
            w = torch.randn(C, 1, k, k, device=dev) * 0.1
            for _ in range(2):
                ops.deform_dwconv2d_backward_cl(x, off, w, g, pad, dil)
            torch.cuda.synchronize()
            L.check(lib.dlka_trace_start(512, st), "start")
            for _ in range(3):
                ops.deform_dwconv2d_backward_cl(x, off, w, g, pad, dil)
            L.check(lib.dlka_trace_stop(), "stop")
            buf, ms, acc = create_string_buffer(512), c_float(), {}
            for i in range(lib.dlka_trace_count()):
                L.check(lib.dlka_trace_get(i, buf, 512, byref(ms)), "get")
                kn = buf.value.decode().replace("void dlka::", "").split("(")[0].replace("cl_ddw2d_", "")
                acc[kn] = acc.get(kn, 0.0) + ms.value / 3
            print(("bf16" if dt == torch.bfloat16 else "f32 "), "C", C, "n", n, "std", std, "k", k, " ".join(f"{n[:22]} {v*1e3:.0f}" for n, v in acc.items()))
