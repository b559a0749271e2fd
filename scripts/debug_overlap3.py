import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import deformablelka_amd as dk
from deformablelka_amd.transformerblock import WgradOverlap
from oracle import blocks
variant = sys.argv[1]
DEV = "cuda:0"
torch.manual_seed(5)
C, (H, W, D) = 64, (16, 16, 16)
mods = []
for _ in range(3):
    m = dk.TransformerBlock_3D_single_deform_LKA(H * W * D, C, C, 4, dropout_rate=0.1, pos_embed=True)
    blocks.randomize_offsets_(m, std=0.3)
    m.keep_channels_last = True
    m.wgrad_overlap = True
    if "lambda" in variant:
        m._draw_drop_mask = lambda B_, C_, dtype, device: torch.ones(B_, C_, dtype=dtype, device=device)
    mods.append(m.to(DEV).train())
x = torch.randn(2, H, W, D, C, device=DEV).permute(0, 4, 1, 2, 3).requires_grad_(True)
gy = torch.randn(2, H, W, D, C, device=DEV).permute(0, 4, 1, 2, 3)
params = [p for m in mods for p in m.parameters()]
def fwd():
    y = x
    for m in mods:
        y = m(y)
    return y
if "retain" in variant:
    y = fwd()
    for _ in range(2):
        for p in params + [x]:
            p.grad = None
        y.backward(gy, retain_graph=True)
        torch.cuda.synchronize()
    if "drop" in variant:
        del y
else:
    for _ in range(2):
        for p in params + [x]:
            p.grad = None
        fwd().backward(gy)
        torch.cuda.synchronize()
for p in params + [x]:
    p.grad = None
print(variant, "capturing", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fwd().backward(gy)
print(variant, "captured", flush=True)
g.replay(); g.replay()
torch.cuda.synchronize()
print(variant, "OK", float(params[5].grad.abs().max()))
