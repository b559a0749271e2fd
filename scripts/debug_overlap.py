import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import deformablelka_amd as dk
from deformablelka_amd import ops
from deformablelka_amd.transformerblock import WgradOverlap
from oracle import blocks
DEV = "cuda:0"
torch.manual_seed(5)
C, (H, W, D) = 64, (16, 16, 16)
mods = []
for _ in range(3):
    m = dk.TransformerBlock_3D_single_deform_LKA(H * W * D, C, C, 4, dropout_rate=0.1, pos_embed=True)
    blocks.randomize_offsets_(m, std=0.3)
    with torch.no_grad():
        m.gamma.normal_(0.5, 0.2)
    m.keep_channels_last = True
    m._draw_drop_mask = lambda B_, C_, dtype, device: torch.ones(B_, C_, dtype=dtype, device=device)
    mods.append(m.to(DEV).train())
x = torch.randn(2, H, W, D, C, device=DEV).permute(0, 4, 1, 2, 3).requires_grad_(True)
gy = torch.randn(2, H, W, D, C, device=DEV).permute(0, 4, 1, 2, 3)
names = ["x"] + [f"m{i}.{k}" for i, m in enumerate(mods) for k, _ in m.named_parameters()]
params = [p for m in mods for p in m.parameters()]

def run(overlap, side=None):
    for m in mods:
        m.wgrad_overlap = overlap
    if side is not None:
        WgradOverlap.get(torch.device(DEV)).side = side
    for p in params + [x]:
        p.grad = None
    y = x
    for m in mods:
        y = m(y)
    y.backward(gy)
    torch.cuda.synchronize()
    return [x.grad.clone()] + [p.grad.clone() for p in params]

ref = run(False)
ref2 = run(False)
def cmp(tag, got):
    bad = []
    for n, a_, b_ in zip(names, ref, got):
        e = float((a_ - b_).abs().max()) / max(float(a_.abs().max()), 1e-30)
        if e > 2e-3:
            bad.append((n, round(e, 4)))
    print(tag, "bad:", len(bad), bad[:12])
cmp("one stream again", ref2)
cur = torch.cuda.current_stream(torch.device(DEV))
cmp("overlap, side = current stream", run(True, side=cur))
cmp("overlap, real side stream", run(True, side=torch.cuda.Stream(device=DEV)))
cmp("overlap, real side stream (2nd)", run(True))
