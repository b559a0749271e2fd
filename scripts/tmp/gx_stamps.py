#!/usr/bin/env python
"""TEMPORARY: s_memtime stamps inside cl_deform_gx_fx2_kernel (stamp build of the library, not committed)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
buf = torch.zeros(16 * 8 * 40, dtype=torch.int64, device="cuda:0")
os.environ["DLKA_GX_STAMP_PTR"] = str(buf.data_ptr())
from deformablelka_amd import ops
g = torch.Generator().manual_seed(0)
B, C, N = 2, 32, 32
x = torch.randn(B, N, N, N, C, generator=g).cuda()
off = torch.randn(B, 81, N, N, N, generator=g).cuda()
w = (torch.randn(C, C, 3, 3, 3, generator=g) * 0.03).cuda()
gy = torch.randn(B, N, N, N, C, generator=g).cuda()
for _ in range(3):
    buf.zero_()
    ops.deform_conv3d_backward_cl(x, off, w, gy, 1, 1)
torch.cuda.synchronize()
t = buf.cpu().view(16, 8, 40)
def med(v): return statistics.median(v) if v else float("nan")
ph = {"prologue: zero + weight staging + barrier": [], "norms + scale (2nd barrier)": [], "group: offsets issued + 16 MFMAs done": [], "group: offsets arrived (after MFMA)": [], "group: scatter of 4 samples/lane": [],
      "tile: grad_out row load etc (group 0 start - previous end)": [], "loop end -> drain": [], "final barrier wait": [], "flush": [], "whole kernel (entry -> end)": [], "main loop": []}
t0s = []
for s in range(16):
    for wv in range(8):
        st = [int(v) for v in t[s, wv]]
        if st[0] == 0: continue
        t0s.append(st[0])
        ph["prologue: zero + weight staging + barrier"].append(st[1] - st[0])
        ph["norms + scale (2nd barrier)"].append(st[2] - st[1])
        ph["main loop"].append(st[27] - st[2])
        for tg in range(8):
            a = 3 + 3 * tg
            ph["group: offsets issued + 16 MFMAs done"].append(st[a + 1] - st[a])
            ph["group: offsets arrived (after MFMA)"].append(st[a + 2] - st[a + 1])
            nxt = st[a + 3] if tg != 7 else st[27]
            ph["group: scatter of 4 samples/lane"].append(nxt - st[a + 2])
        ph["loop end -> drain"].append(st[28] - st[27])
        ph["final barrier wait"].append(st[29] - st[28])
        ph["flush"].append(st[30] - st[29])
        ph["whole kernel (entry -> end)"].append(st[30] - st[0])
print("ticks (s_memtime), median / p10 / p90 over %d waves" % len(t0s))
for k, v in ph.items():
    v = sorted(v)
    if v: print(f"  {k:60s} {med(v):9.0f} {v[len(v)//10]:9.0f} {v[9*len(v)//10]:9.0f}")
if t0s: print("  spread of entry stamps over the sampled workgroups:", max(t0s) - min(t0s))
# per-workgroup detail of the first sampled workgroup
for s in range(0, 16, 5):
    for wv in (0, 5):
        st = [int(v) for v in t[s, wv]]
        if st[0]: print("wg", s, "wave", wv, [st[i] - st[0] for i in range(31)])
