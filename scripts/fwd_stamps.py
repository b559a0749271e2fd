#!/usr/bin/env python
"""s_memtime stamps inside the 16-row deformable forward kernel (DLKA_FWD_ABL=7, profiling only): where do a unit's cycles go?
Phases per unit: 0 loop top | 1 weight chunk stored to LDS | 4 finish() done (gathers waited for, interpolated, A operand read back) |
5 workgroup barrier passed | 6 next unit's weight + offsets + corner loads issued (description formed) | 7 MFMAs issued."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ["DLKA_FWD_ABL"] = "7"
buf = torch.zeros(16 * 8 * 8, dtype=torch.int64, device="cuda:0")
os.environ["DLKA_FWD_STAMP_PTR"] = str(buf.data_ptr())
from deformablelka_amd import ops
g = torch.Generator().manual_seed(0)
B, C, N = 2, 32, 32
x = torch.randn(B, N, N, N, C, generator=g).cuda()
off = torch.randn(B, 81, N, N, N, generator=g).cuda()
w = (torch.randn(C, C, 3, 3, 3, generator=g) * 0.03).cuda()
b = torch.randn(C, generator=g).cuda()
for _ in range(3):
    ops.deform_conv3d_forward_cl(x, off, w, b, 1, 1)
torch.cuda.synchronize()
t = buf.cpu().view(16, 8, 8)
names = {1: "store B chunk", 4: "finish (wait gathers + interp + A readback)", 5: "barrier", 2: "load_b + tap decode + describe + publish", 3: "table lookups", 6: "corner offsets + 16 loads issued", 7: "MFMA loop"}
import statistics
print("cycles per phase (s_memtime ticks = 100 MHz? reported raw), median over 16 waves x 7 units")
ORDER = (1, 4, 5, 2, 3, 6, 7)
rows = {k: [] for k in ORDER}
tot = []
for s in range(16):
    for u in range(7):
        st = t[s, u]
        if st[0] == 0: continue
        prev = st[0]
        for k in ORDER:
            rows[k].append(int(st[k] - prev)); prev = st[k]
        tot.append(int(t[s, u + 1, 0] - st[0]))
for k in ORDER:
    if rows[k]: print(f"  {names[k]:45s} median {statistics.median(rows[k]):8.0f}  p10 {sorted(rows[k])[len(rows[k])//10]:8.0f}  p90 {sorted(rows[k])[9*len(rows[k])//10]:8.0f}")
if tot: print(f"  {'whole unit':45s} median {statistics.median(tot):8.0f}")
