#!/usr/bin/env python
"""TEMPORARY: fine stamps inside one tile-group's scatter of cl_deform_gx_fx2_kernel."""
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
names = ["MFMA phase (offsets issued, 16 MFMAs issued)"]
for r in range(4):
    names += [f"s{r}: description", f"s{r}: weights + 16 atomics issued", f"s{r}: LDS queue drained (lgkmcnt 0)", f"s{r}: far ballot/queue"]
names += ["drain check"]
rows = [[] for _ in names]
for s in range(16):
    for wv in range(8):
        st = [int(v) for v in t[s, wv]]
        if st[0] == 0: continue
        # stamps: 0 start; per sample: 1+4r (top), 2+4r (described), 3+4r (atomics issued), 4+4r (lgkm 0); 17 = before drain check; 18 after
        rows[0].append(st[1] - st[0])
        for r in range(4):
            a = 1 + 4 * r
            rows[1 + 4 * r].append(st[a + 1] - st[a])
            rows[2 + 4 * r].append(st[a + 2] - st[a + 1])
            rows[3 + 4 * r].append(st[a + 3] - st[a + 2])
            nxt = st[a + 4] if r < 3 else st[17]
            rows[4 + 4 * r].append(nxt - st[a + 3])
        rows[17].append(st[18] - st[17])
print("ticks, median / p10 / p90 over", len(rows[0]), "waves (one tile-group each)")
for n, v in zip(names, rows):
    v = sorted(v)
    if v: print(f"  {n:50s} {statistics.median(v):8.0f} {v[len(v)//10]:8.0f} {v[9*len(v)//10]:8.0f}")
