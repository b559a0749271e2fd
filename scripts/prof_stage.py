#!/usr/bin/env python
"""One block of one Synapse stage, fwd+bwd as a hipGraph, replayed (for rocprofv3 --kernel-trace --stats).
Usage: python scripts/prof_stage.py --stage 3 [--iters 20] [--trace]
--trace: also print the per-kernel durations of the block from the library's own launch trace (include/dlka.h dlka_trace_*: a HIP event behind
every launch, eager run) — a few seconds instead of a rocprofv3 pass; each record includes ~3 us of event cost."""
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
ap.add_argument("--trace", action="store_true")
ap.add_argument("--lib", default=None, help="another build of libdlka_hip.so (scripts/build_variant.sh), path relative to the repo root")
a = ap.parse_args()
if a.lib:
    import ctypes
    from deformablelka_amd import _lib as _L
    _L._lib = _L.bind(ctypes.CDLL(os.path.join(ROOT, a.lib)))
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

if a.trace:
    from ctypes import byref, c_float, create_string_buffer
    from deformablelka_amd import _lib as L
    lib = L.get_lib()
    stream = torch.cuda.current_stream().cuda_stream
    reps = 5
    st.forward_backward()
    torch.cuda.synchronize()
    try:
        torch.cuda._sleep(int(40e6))   # keep the host ahead of the device
    except Exception:
        pass
    L.check(lib.dlka_trace_start(4096, stream), "trace_start")
    try:
        for _ in range(reps):
            st.forward_backward()
    finally:
        rc = lib.dlka_trace_stop()
    L.check(rc, "trace_stop")
    buf, ms, acc = create_string_buffer(512), c_float(), {}
    for i in range(lib.dlka_trace_count()):
        L.check(lib.dlka_trace_get(i, buf, 512, byref(ms)), "trace_get")
        k = buf.value.decode().replace("void dlka::", "").replace("dlka::", "").split("(")[0]
        e = acc.setdefault(k, [0, 0.0])
        e[0] += 1
        e[1] += ms.value
    tot = sum(v[1] for v in acc.values()) / reps
    print(f"launch trace: {sum(v[0] for v in acc.values()) // reps} launches, {tot * 1e3:.1f} us per fwd+bwd (records include the event cost)")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k[:70]:70s} x{v[0] // reps:<3d} {v[1] / v[0] * 1e3:8.1f} us")
