import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deformablelka_amd.stack import DLKABlockStack, SYNAPSE_STAGES
from deformablelka_amd import _lib as L
torch.cuda.set_device(0)
st = DLKABlockStack(2, device="cuda:0", seed=1234)
names = L.LKA3D_FIELDS
a256 = lambda n: (n + 255) & ~255
for it in range(int(os.environ.get("ITERS", "30"))):
    st.forward_backward()
    torch.cuda.synchronize()
    bad = []
    for i, blk in enumerate(st.blocks):
        H, W, D = blk.dims
        N = H * W * D
        E, Off = st.B * blk.C * N, st.B * 81 * N
        segs = [("h", E), ("a", E), ("t1", E), ("t", E), ("off", Off), ("f", E), ("g1", E)]
        o = 0
        for nm, n in segs:
            t = blk.saved[o:o + n * 4].view(torch.float32)
            if not torch.isfinite(t).all():
                bad.append((i, "saved." + nm, int((~torch.isfinite(t)).sum())))
            o += a256(n * 4)
        for nm, t in (("x", blk.x), ("y", blk.y), ("gy", blk.gy), ("gx", blk.gx)):
            if not torch.isfinite(t).all():
                bad.append((i, nm, int((~torch.isfinite(t)).sum())))
        for k, g in enumerate(blk.grads):
            if not torch.isfinite(g).all():
                bad.append((i, "grad." + names[k], int((~torch.isfinite(g)).sum())))
    if bad or it % 5 == 0: print("iter", it, "bad:", bad[:12], "max|gx| blocks:", [round(float(b.gx.abs().max()), 1) for b in st.blocks[::3]])
    st.reduce_and_update(1e-12, 1, None)
