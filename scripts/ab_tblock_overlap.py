#!/usr/bin/env python
"""The wrapper-block stack (bench.tblock_metric) and the full-net trainer iteration (bench.fullnet_metric) with the weight gradients on the side stream
(transformerblock.WgradOverlap) against one stream, interleaved in ONE process.  usage: python scripts/ab_tblock_overlap.py OUT.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from deformablelka_amd import training
dev = torch.device("cuda", 0)
res = {"tblock_overlap": [], "tblock_one_stream": [], "fullnet_overlap": [], "fullnet_one_stream": []}
for rnd in range(2):
    for ov in (True, False):
        r = bench.tblock_metric(2, 10, 3, dev, overlap=ov)
        res["tblock_overlap" if ov else "tblock_one_stream"].append((r["value"], r.get("hipgraph", {}).get("value")))
        torch.cuda.empty_cache()
orig = training.initialize_network
for rnd in range(2):
    for ov in (True, False):
        training.initialize_network = lambda *a, _ov=ov, **k: orig(*a, **dict(k, wgrad_overlap=_ov))
        r = bench.fullnet_metric(2, 5, dev)
        res["fullnet_overlap" if ov else "fullnet_one_stream"].append(r["value"])
        torch.cuda.empty_cache()
training.initialize_network = orig
json.dump(res, open(sys.argv[1], "w"), indent=1)
for k, v in res.items():
    print(k, v)
