import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deformablelka_amd.stack import DLKABlockStack
from deformablelka_amd import _lib as L
torch.cuda.set_device(0)
for C, dims in ((32, (32, 32, 32)), (64, (16, 16, 16)), (128, (8, 8, 8)), (256, (4, 4, 4))):
    st = DLKABlockStack(2, stages=((C, dims, 1),), device="cuda:0", seed=1)
    res = []
    for fill in (0xFF, 0x00, 0x7F):
        st.ws.fill_(fill)
        st.forward()
        st.ws.fill_(fill)
        st.backward()
        torch.cuda.synchronize()
        res.append([g.clone() for g in st.blocks[0].grads] + [st.blocks[0].gx.clone(), st.blocks[0].y.clone()])
    names = list(L.LKA3D_FIELDS) + ["gx", "y"]
    for k, nm in enumerate(names):
        a, b, c = res[0][k], res[1][k], res[2][k]
        nonfin = int((~torch.isfinite(a)).sum())
        d = float((b - c).abs().max())
        if nonfin or d > 1e-4 * max(1.0, float(b.abs().max())):
            print(f"C={C} {nm}: nonfinite with NaN-poisoned workspace: {nonfin}; max |diff| between 0x00 and 0x7F poison: {d:.3e} (max |val| {float(b.abs().max()):.3e})")
    print(f"C={C} done")
