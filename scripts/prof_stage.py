#!/usr/bin/env python
"""One block of one Synapse stage, fwd+bwd as a hipGraph, replayed (for rocprofv3 --kernel-trace --stats).
Usage: python scripts/prof_stage.py --stage 3 [--iters 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deformablelka_amd.stack import SYNAPSE_STAGES, DLKABlockStack  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--stage", type=int, default=3)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
a = ap.parse_args()
torch.cuda.set_device(0)
C, dims, n = SYNAPSE_STAGES[a.stage]
st = DLKABlockStack(a.batch, stages=((C, dims, 1),), device="cuda:0", dtype=torch.float32 if a.dtype == "f32" else torch.bfloat16)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    st.forward_backward()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    st.forward_backward()
g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    g.replay()
e1.record()
torch.cuda.synchronize()
print(f"stage {a.stage} C={C} {dims}: graph fwd+bwd {e0.elapsed_time(e1) / a.iters:.4f} ms")
