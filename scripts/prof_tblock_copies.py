#!/usr/bin/env python
"""torch ops (copies, adds, fills, foreach updates ...) of one step of the wrapper-block stack as bench.tblock_metric runs it: what surrounds the C-ABI calls of the 21 blocks.
Device time per aten op with the op chain that issued it (torch.profiler)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import deformablelka_amd as dk
from deformablelka_amd.stack import SYNAPSE_STAGES, CHAIN, _offset_std_for

dev = torch.device("cuda:0")
torch.manual_seed(0)
chains = []
for C, (H, W, D), n in SYNAPSE_STAGES:
    for c0 in range(0, n, CHAIN):
        mods = []
        for _ in range(min(CHAIN, n - c0)):
            m = dk.TransformerBlock_3D_single_deform_LKA(H * W * D, C, C, 4, dropout_rate=0.1, pos_embed=True)
            with torch.no_grad():
                m.epa_block.spatial_gating_unit.deform_conv.conv_offset.weight.normal_(0, _offset_std_for(C))
            m.keep_channels_last = True
            mods.append(m.to(dev))
        x = torch.randn(2, H, W, D, C, device=dev).permute(0, 4, 1, 2, 3).requires_grad_(True)
        gy = torch.randn(2, H, W, D, C, device=dev).permute(0, 4, 1, 2, 3)
        chains.append((mods, x, gy))
params = [p for mods, _, _ in chains for m in mods for p in m.parameters()] + [x for _, x, _ in chains]


def step():
    for p in params:
        p.grad = None
    for mods, x, gy in chains:
        y = x
        for m in mods:
            y = m(y)
        y.backward(gy)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
acc = {}
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.device_time_total > 0:   # (nested ops are listed with their parents: clone <- contiguous counts once in each line)
        chain, p = [], ev.cpu_parent
        while p is not None and len(chain) < 3:
            chain.append(p.name[:50])
            p = p.cpu_parent
        key = (ev.name, " <- ".join(chain))
        a = acc.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += ev.device_time_total
tot = sum(v[1] for v in acc.values())
print("aten leaf ops with device time: %.1f us per step" % tot)
for (n, ch), (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{n:28s} n={c:4d} dev={t:8.1f} us  {ch}")
