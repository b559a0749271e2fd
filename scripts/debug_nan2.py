import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deformablelka_amd.stack import DLKABlockStack
torch.cuda.set_device(0)
st = DLKABlockStack(2, device="cuda:0", seed=1234)
n = int(os.environ.get("ITERS", "25"))
upd = os.environ.get("UPD", "1") == "1"
for it in range(n):
    st.forward_backward()
    if upd:
        st.reduce_and_update(1e-12, 1, None)
torch.cuda.synchronize()
print("after", n, "unsynced iterations, update", upd, ":", st.health())
bad = [(i, k) for i, blk in enumerate(st.blocks) for k, g in enumerate(blk.grads) if not torch.isfinite(g).all()]
print("non-finite grads (block, param):", bad[:20])
badp = [(i, k) for i, blk in enumerate(st.blocks) for k, g in enumerate(blk.params) if not torch.isfinite(g).all()]
print("non-finite params (block, param):", badp[:20])
