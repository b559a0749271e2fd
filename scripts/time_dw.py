#!/usr/bin/env python
"""A/B of the depthwise 5^3 / 7^3 dilation-3 forward kernels at the stage shapes: DLKA_DW_2P=0 (row kernel) against 1 / 2 (software-pipelined
row-pair kernel, deep / one-row ring), read per launch.  HIP events on the current stream, results compared bit for bit.
Usage: python scripts/time_dw.py [--iters 50] [--bf16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--modes", default="0,1,2")
    ap.add_argument("--shapes", default="0,1,2")
    args = ap.parse_args()
    from deformablelka_amd import ops
    dev = torch.device("cuda", 0)
    dt = torch.bfloat16 if args.bf16 else torch.float32
    g = torch.Generator().manual_seed(0)
    allshapes = ((2, 32, 32), (2, 64, 16), (2, 128, 8))
    for (B, C, N) in [allshapes[int(i)] for i in args.shapes.split(",")]:
        x = torch.randn(B, N, N, N, C, generator=g).to(dev, dt)
        for (k, p, d) in ((5, 2, 1), (7, 9, 3)):
            w = (torch.randn(C, 1, k, k, k, generator=g) * 0.1).to(dev)
            b = torch.randn(C, generator=g).to(dev)
            ref = None
            for mode in args.modes.split(","):
                os.environ["DLKA_DW_2P"] = mode
                y = ops.conv3d_forward_cl(x, w, b, p, d, C)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    y = ops.conv3d_forward_cl(x, w, b, p, d, C)
                e1.record()
                torch.cuda.synchronize()
                same = True if ref is None else bool(torch.equal(ref, y))
                if ref is None:
                    ref = y
                print(f"C={C} N={N} k={k} mode={mode}: {e0.elapsed_time(e1) / args.iters * 1e3:8.1f} us per call (incl. weight preparation launch)  equal={same}", flush=True)
    os.environ.pop("DLKA_DW_2P", None)


if __name__ == "__main__":
    main()
