#!/usr/bin/env python
"""HIP-event timing of TransformerBlock_3D_single_deform_LKA (wrapper + D-LKA block) per Synapse stage (GPU box).
Usage: python scripts/tblock_breakdown.py [--batch 2]   -> JSON on stdout, table on stderr"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import deformablelka_amd as dk  # noqa: E402
from deformablelka_amd.stack import SYNAPSE_STAGES  # noqa: E402


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
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = "cuda:0"
    out = {"batch": a.batch, "stages": []}
    for C, (H, W, D), n in SYNAPSE_STAGES:
        torch.manual_seed(0)
        m = dk.TransformerBlock_3D_single_deform_LKA(H * W * D, C, C, 4, dropout_rate=0.1, pos_embed=True).to(dev)
        m.keep_channels_last = True
        inner = m.epa_block
        x = torch.randn(a.batch, C, H, W, D, device=dev).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3).requires_grad_(True)
        gy = torch.randn(a.batch, H, W, D, C, device=dev).permute(0, 4, 1, 2, 3)
        tok = torch.randn(a.batch, H * W * D, C, device=dev, requires_grad=True)
        gtok = torch.randn_like(tok)

        def full():   # (grads dropped each time: autograd's `.grad +=` accumulation is not part of the block)
            m.zero_grad(set_to_none=True)
            x.grad = None
            m(x).backward(gy)

        def lka():
            m.zero_grad(set_to_none=True)
            tok.grad = None
            inner(tok, a.batch, C, H, W, D).backward(gtok)

        def fwd():
            with torch.no_grad():
                m(x)
        t_full, t_lka, t_fwd = ev_time(full, a.iters), ev_time(lka, a.iters), ev_time(fwd, a.iters)
        row = {"C": C, "dims": [H, W, D], "blocks": n, "wrapper_fwd_bwd_ms": round(t_full, 4), "lka_only_fwd_bwd_ms": round(t_lka, 4),
               "wrapper_fwd_ms": round(t_fwd, 4)}
        out["stages"].append(row)
        print(f"C={C:3d} {H}x{W}x{D}: wrapper fwd+bwd {t_full:.3f} ms (fwd {t_fwd:.3f}), D-LKA alone {t_lka:.3f} ms -> surroundings {t_full - t_lka:.3f} ms",
              file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
