#!/usr/bin/env python
"""Run selected kernel-level ops of one stage repeatedly (for rocprofv3 --kernel-trace --stats / --pmc).
Usage: python scripts/prof_op.py --C 32 --N 32 --ops deform_bwd_input,deform_fwd [--iters 10]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--C", type=int, default=32)
ap.add_argument("--N", type=int, default=32)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--ops", default="deform_bwd_input")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
torch.cuda.set_device(0)
res = bench.time_ops(a.batch, a.C, a.N, torch.float32, iters=a.iters, only=set(a.ops.split(",")))
for k, v in res.items():
    print(f"{k}: {v:.4f} ms")
