#!/usr/bin/env python
"""A/B of two builds of the library on the 2-D block metric (bench.lka2d_metric: config 2, bf16, B = 24) — or, with AB_METRIC=tblock, on the wrapper-block stack
(bench.tblock_metric) — in ONE process on ONE box, interleaved rounds (box-to-box spread on the pool exceeds most single-kernel changes).
usage: [AB_METRIC=tblock] python scripts/ab_lka2d.py OUT.json [alt_lib/libdlka_hip_prev.so]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from deformablelka_amd import _lib as L
out_path = sys.argv[1]
alt = sys.argv[2] if len(sys.argv) > 2 else "alt_lib/libdlka_hip_prev.so"
libs = {"cur": None}
if os.path.exists(os.path.join(ROOT, alt)):
    cd = ctypes.CDLL(os.path.join(ROOT, alt))
    for name, (rs, args) in L.SIGNATURES.items():   # (an older build may lack this round's new exports: bind what it has)
        if hasattr(cd, name):
            fn = getattr(cd, name); fn.restype = rs; fn.argtypes = args
    libs["prev"] = cd
res = {k: [] for k in libs}
kern = {}
for rnd in range(3):
    for name, lib in libs.items():
        L._lib = lib
        if os.environ.get("AB_METRIC") == "tblock":
            r = bench.tblock_metric(2, 10, 3, torch.device("cuda", 0))
            res[name].append((r["value"], r.get("hipgraph", {}).get("value")))
            continue
        r = bench.lka2d_metric(10, torch.device("cuda", 0), torch.bfloat16)
        res[name].append(r["value"])
        if "roofline" in r:
            kern[name] = {f'{k["kernel"]} {k["shape"]}': k["avg_us"] for k in r["roofline"]["kernels"]}
L._lib = None
json.dump({"images_per_s": res, "kernels": kern}, open(out_path, "w"), indent=1)
for k, v in res.items():
    print(k, v)
for k in sorted(set().union(*[set(d) for d in kern.values()]), key=lambda k: -max(d.get(k, 0) for d in kern.values())):
    print("%-70s %s" % (k[:70], " ".join("%9.1f" % kern[n].get(k, float("nan")) for n in kern)))
