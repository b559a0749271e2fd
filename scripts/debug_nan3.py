import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deformablelka_amd.stack import DLKABlockStack
from deformablelka_amd import _lib as L
torch.cuda.set_device(0)
st = DLKABlockStack(2, device="cuda:0", seed=1234)
st.forward_backward()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    st.forward_backward()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    st.forward_backward()
names = L.LKA3D_FIELDS
sync_each = os.environ.get("SYNC", "1") == "1"
for it in range(40):
    g.replay()
    st.reduce_and_update(1e-12, 1, None)
    if sync_each or it == 39:
        torch.cuda.synchronize()
        bad = [(i, names[k], int((~torch.isfinite(gr)).sum())) for i, blk in enumerate(st.blocks) for k, gr in enumerate(blk.grads) if not torch.isfinite(gr).all()]
        badx = [(i, nm) for i, blk in enumerate(st.blocks) for nm, t in (("y", blk.y), ("gx", blk.gx)) if not torch.isfinite(t).all()]
        if bad or badx or it % 10 == 0:
            print("replay", it, "bad grads:", bad[:10], "bad acts:", badx[:10])
        if bad or badx:
            break
