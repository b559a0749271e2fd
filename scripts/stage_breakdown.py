#!/usr/bin/env python
"""Per-stage / per-op HIP-event timing of the D-LKA block stack (GPU box).  Writes JSON to stdout.
Usage: python scripts/stage_breakdown.py [--batch 2]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from deformablelka_amd.stack import SYNAPSE_STAGES, DLKABlockStack  # noqa: E402


def ev_time(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    out = {"batch": a.batch, "stages": []}
    for C, dims, n in SYNAPSE_STAGES:
        st = DLKABlockStack(a.batch, stages=((C, dims, 1),), device="cuda:0")
        fwd = ev_time(st.forward)
        bwd = ev_time(st.backward)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            st.forward_backward()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            st.forward_backward()
        gr = ev_time(g.replay)
        ops = bench.time_ops(a.batch, C, dims[0], torch.float32)
        tab = bench.op_table(a.batch, C, dims[0], 4)
        rows = {k: {"ms": round(v, 4), "tflops": round(tab[k][0] / v / 1e9, 2), "gbs": round(tab[k][1] / v / 1e6, 1)} for k, v in ops.items()}
        out["stages"].append({"C": C, "dims": dims, "blocks": n, "block_fwd_ms": round(fwd, 4), "block_bwd_ms": round(bwd, 4),
                              "block_graph_ms": round(gr, 4), "ops": rows})
        print(f"C={C} {dims}: fwd {fwd:.3f} bwd {bwd:.3f} graph fwd+bwd {gr:.3f} ms  x{n} = {gr * n:.2f} ms", file=sys.stderr)
        for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
            print(f"    {k:26s} {r['ms']:8.4f} ms {r['tflops']:8.2f} TF/s {r['gbs']:8.1f} GB/s", file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
