"""Which kernels does torch/MIOpen pick for the fp32 3-D convs of the D_LKA_Former plumbing?  (profiles/archive/r03f: naive 'nonpacked' kernels = 61 % of a step)"""
import sys
import time
import torch
from torch.profiler import profile, ProfilerActivity

dev = "cuda:0"
cases = [("stem k(2,4,4) s(2,4,4) 1->32", torch.nn.Conv3d(1, 32, (2, 4, 4), (2, 4, 4), bias=False), (2, 1, 64, 128, 128)),
         ("down k2 s2 32->64", torch.nn.Conv3d(32, 64, 2, 2, bias=False), (2, 32, 32, 32, 32)),
         ("transp k2 s2 64->32", torch.nn.ConvTranspose3d(64, 32, 2, 2, bias=False), (2, 64, 16, 16, 16)),
         ("transp k(2,4,4) 32->16", torch.nn.ConvTranspose3d(32, 16, (2, 4, 4), (2, 4, 4), bias=False), (2, 32, 32, 32, 32)),
         ("3x3x3 1->16 full res", torch.nn.Conv3d(1, 16, 3, 1, 1, bias=False), (2, 1, 64, 128, 128)),
         ("3x3x3 16->16 full res", torch.nn.Conv3d(16, 16, 3, 1, 1, bias=False), (2, 16, 64, 128, 128)),
         ("1x1x1 16->14 full res", torch.nn.Conv3d(16, 14, 1), (2, 16, 64, 128, 128))]
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    print("== cudnn.benchmark =", bench)
    for name, m, shp in cases:
        m = m.to(dev)
        x = torch.randn(shp, device=dev, requires_grad=True)
        for _ in range(2):
            m(x).sum().backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m(x).sum().backward()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3 * 1e3
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            m(x).sum().backward()
            torch.cuda.synchronize()
        top = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:3]
        print(f"{name:28s} {dt:9.2f} ms  " + " | ".join(f"{e.key[:48]} {e.device_time_total / 1e3:.2f}ms" for e in top))
