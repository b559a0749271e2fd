#!/usr/bin/env python
"""The three GEMMs behind the net's last up-sampling transposed conv (kernel == stride (2,4,4), 32 -> 16 channels, 2 x 32^3 -> 2 x 64x128x128; network.Convolution) under several
formulations: which operand order / transposition does rocBLAS (through torch) run fastest?  us per call, median of 20."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
M, K, N = 65536, 32, 512
x = torch.randn(M, K, device=dev)
w = torch.randn(K, N, device=dev)
gy = torch.randn(M, N, device=dev)
xt = x.t().contiguous(); wt = w.t().contiguous(); gyt = gy.t().contiguous()


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[n // 2]


out = torch.empty(M, N, device=dev)
cases = {
    "fwd  y = x @ w                         (M,K)@(K,N)": lambda: torch.mm(x, w),
    "fwd  y = (w^T @ x^T)^T  via mm(wt, xt)": lambda: torch.mm(wt, xt),
    "fwd  y = x @ w, out=": lambda: torch.mm(x, w, out=out),
    "gx   = gy @ w^T  (w.t() view)           (M,N)@(N,K)": lambda: torch.mm(gy, w.t()),
    "gx   = gy @ wt (contiguous N,K)": lambda: torch.mm(gy, wt),
    "gx^T = w @ gy^T (gy.t() view)": lambda: torch.mm(w, gy.t()),
    "gx^T = w @ gyt (contiguous)": lambda: torch.mm(w, gyt),
    "gx   chunks of 8192 rows": lambda: [torch.mm(gy[i:i + 8192], wt) for i in range(0, M, 8192)],
    "gW   = x^T @ gy (x.t() view)            (K,M)@(M,N)": lambda: torch.mm(x.t(), gy),
    "gW   = xt @ gy (contiguous)": lambda: torch.mm(xt, gy),
    "gW^T = gy^T @ x (gy.t() view)": lambda: torch.mm(gy.t(), x),
    "gW   = sum of 16 chunk products (bmm)": lambda: torch.bmm(x.view(16, M // 16, K).transpose(1, 2), gy.view(16, M // 16, N)).sum(0),
    "gW   = sum of 64 chunk products (bmm)": lambda: torch.bmm(x.view(64, M // 64, K).transpose(1, 2), gy.view(64, M // 64, N)).sum(0),
    "gW   = sum of 256 chunk products (bmm)": lambda: torch.bmm(x.view(256, M // 256, K).transpose(1, 2), gy.view(256, M // 256, N)).sum(0),
    "copy 134 MB (depth-to-space stand-in: permuted copy)": lambda: gy.view(2, 32, 32, 32, 16, 2, 4, 4).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(2, 16, 64, 128, 128),
    "copy back (space-to-depth)": lambda: out.view(2, 16, 32, 2, 32, 4, 32, 4).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(M, N),
}
for name, fn in cases.items():
    print("%9.1f us  %s" % (t(fn), name))
