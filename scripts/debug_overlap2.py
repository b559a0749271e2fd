import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import deformablelka_amd as dk
from deformablelka_amd import ops
from oracle import blocks
DEV = "cuda:0"
torch.manual_seed(5)
C, dims = 64, (16, 16, 16)
H, W, D = dims
B = 2
m = dk.TransformerBlock_3D_single_deform_LKA(H * W * D, C, C, 4, dropout_rate=0.1, pos_embed=True)
blocks.randomize_offsets_(m, std=0.3)
with torch.no_grad():
    m.gamma.normal_(0.5, 0.2)
m = m.to(DEV).train()
x = torch.randn(B, H * W * D, C, device=DEV)
gy = torch.randn(B, H * W * D, C, device=DEV)
tparams = [None if p is None else p.detach() for p in m.wrapper_params()]
lparams = [p.detach() for p in m.epa_block.block_params()]
stats = torch.empty(6 * C, dtype=torch.float32, device=DEV)
mask = torch.ones(B, C, device=DEV)
y, saved = ops.tblock3d_forward(x, False, tparams, lparams, mask, True, stats, dims, 1e-5, 1e-5, 0, False)
def flat(r):
    return [r[0]] + [t for t in r[1] if t is not None] + list(r[2])
whole = flat(ops.tblock3d_backward(tparams, lparams, mask, True, stats, gy, saved, dims, 0, False))
torch.cuda.synchronize()
def cmp(tag, got):
    torch.cuda.synchronize()
    bad = [(k, round(float((a - b).abs().max()) / max(float(a.abs().max()), 1e-30), 4)) for k, (a, b) in enumerate(zip(whole, flat(got))) if float((a - b).abs().max()) > 2e-3 * max(float(a.abs().max()), 1e-30)]
    print(tag, "bad:", len(bad), bad[:10])
cmp("inline", ops.tblock3d_backward(tparams, lparams, mask, True, stats, gy, saved, dims, 0, False, side_stream="inline"))
side = torch.cuda.Stream(device=DEV)
for rep in range(3):
    r = ops.tblock3d_backward(tparams, lparams, mask, True, stats, gy, saved, dims, 0, False, side_stream=side)
    cmp(f"side stream rep {rep}", r[:3])
cur = torch.cuda.current_stream(torch.device(DEV))
cmp("side = current", ops.tblock3d_backward(tparams, lparams, mask, True, stats, gy, saved, dims, 0, False, side_stream=cur)[:3])
os.environ["DLKA_GX_FORK_MIN_ROWS"] = "1000000000"
from deformablelka_amd import _lib as L
L.get_lib().dlka_env_refresh()
r = ops.tblock3d_backward(tparams, lparams, mask, True, stats, gy, saved, dims, 0, False, side_stream=side)
cmp("side stream, no internal gx fork", r[:3])
