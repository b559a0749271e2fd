#!/usr/bin/env python
"""Where does the launch-to-launch spread of cl_deform_gx_fx2_kernel come from (124 us min / 152 avg / 217 max on one stream)?
Per-LAUNCH durations of the stage-0 grad_input kernel from the library's launch trace (one stream, eager, back to back), keyed by block and by repetition:
a spread that repeats per block is DATA (each block has its own offsets); a spread across repetitions of the same block is PLACEMENT / machine state.
Printed beside each block: the fraction of samples with |offset| > 1 voxel and the far-sample share.  usage: python scripts/gx_spread.py [reps]
(A/B: run again with DLKA_NO_XCD_SWIZZLE=1 — read once per process.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ctypes import byref, c_float, create_string_buffer
from deformablelka_amd import _lib as L
from deformablelka_amd.stack import DLKABlockStack, SYNAPSE_STAGES

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
os.environ["DLKA_GX_FORK_MIN_ROWS"] = "1000000000"   # one stream
lib = L.get_lib()
lib.dlka_env_refresh()
dev = torch.device("cuda", 0)
st = DLKABlockStack(2, stages=(SYNAPSE_STAGES[0],), device=dev, seed=1234, data_seed=4321)
st._overlap = False
for _ in range(2):
    st.forward_backward()
torch.cuda.synchronize()
stream = torch.cuda.current_stream(dev).cuda_stream
order = []


def hook(i):
    order.append(i)
    L.check(lib.dlka_trace_mark(stream), "mark")


torch.cuda._sleep(int(60e6))
L.check(lib.dlka_trace_start(16384, stream), "start")
for r in range(reps):
    st.forward_backward(on_block=hook)
L.check(lib.dlka_trace_stop(), "stop")
buf, ms = create_string_buffer(512), c_float()
mark_i, cur = -1, -1
per = {}     # block -> [durations of gx_fx2 in issue order]
other = {}
for i in range(lib.dlka_trace_count()):
    L.check(lib.dlka_trace_get(i, buf, 512, byref(ms)), "get")
    name = buf.value.decode()
    if name == "(mark)":
        mark_i += 1
        cur = order[mark_i]
        continue
    if "cl_deform_gx_fx2" in name:
        per.setdefault(cur, []).append(ms.value * 1e3)
    elif "cl_deform_goff16" in name:
        other.setdefault(cur, []).append(ms.value * 1e3)
h = st.health() if hasattr(st, "health") else {}
print("offset std per stage:", h.get("offset_std"))
from deformablelka_amd import ops
import statistics as S
allv = []
print("block | gx_fx2 us per repetition ... | mean  sd | goff16 mean sd | frac |off|>1  frac |off|>2.5")
for bi, blk in enumerate(st.blocks):
    off = ops.lka3d_tokens_saved_offsets(blk.saved, st.B, blk.C, blk.dims)
    f1 = float((off.abs() > 1).float().mean()); f2 = float((off.abs() > 2.5).float().mean())
    v = per.get(bi, []); g = other.get(bi, [])
    allv += v
    print("%5d | %s | %6.1f %5.1f | %6.1f %5.1f | %.3f %.4f" % (bi, " ".join("%6.1f" % x for x in v), S.mean(v), S.pstdev(v), S.mean(g), S.pstdev(g), f1, f2))
print("all launches: n=%d min %.1f mean %.1f max %.1f sd %.1f" % (len(allv), min(allv), S.mean(allv), max(allv), S.pstdev(allv)))
# per-repetition means: does a whole repetition run slow (machine state) ?
for r in range(reps):
    vs = [per[b][r] for b in sorted(per) if len(per[b]) > r]
    print("rep %d: mean %.1f" % (r, S.mean(vs)))
