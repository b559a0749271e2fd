import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deformablelka_amd.stack import DLKABlockStack
from deformablelka_amd import _lib as L
torch.cuda.set_device(0)
stages = ((32, (32, 32, 32), 2), (64, (16, 16, 16), 1)) if os.environ.get("SMALL") else None
st = DLKABlockStack(2, device="cuda:0", seed=1234) if stages is None else DLKABlockStack(2, stages=stages, device="cuda:0", seed=1234)
a256 = lambda n: (n + 255) & ~255
def snapshot():
    torch.cuda.synchronize()
    out = {}
    for i, blk in enumerate(st.blocks):
        H, W, D = blk.dims
        N = H * W * D
        E, Off = st.B * blk.C * N, st.B * 81 * N
        o = 0
        for nm, n in (("h", E), ("a", E), ("t1", E), ("t", E), ("off", Off), ("f", E), ("g1", E)):
            out[(i, "0fwd." + nm)] = blk.saved[o:o + n * 4].view(torch.float32).clone()
            o += a256(n * 4)
        out[(i, "0fwd.y")] = blk.y.clone()
        out[(i, "1bwd.gx")] = blk.gx.clone()
        for k, g in enumerate(blk.grads):
            out[(i, "1bwd.grad." + L.LKA3D_FIELDS[k])] = g.clone()
    return out
st.forward_backward()
ref = snapshot()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    st.forward_backward()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    st.forward_backward()
for it in range(3):
    g.replay()
    cur = snapshot()
    diffs = []
    for k in sorted(ref.keys(), key=lambda kk: (kk[1][0] == "1", kk[0] if kk[1][0] == "0" else -kk[0], kk[1])):
        a, b = ref[k], cur[k]
        bad = ~torch.isfinite(b)
        d = float((a - b).abs().max()) if not bad.any() else float("inf")
        if d > 1e-3 * max(1.0, float(a.abs().max())):
            diffs.append((k, d, int(bad.sum())))
    print("replay", it, "first differing tensors (forward order, then backward order):", diffs[:6])
