#!/usr/bin/env python
"""BASELINE.md §3 protocol, offline: the oracle D-LKA block (ATen CPU convs + the C oracle, autograd backward) at B = 2, fp32, offsets ~1 voxel, on the host
cores of the machine this runs on — 3 warm-up + 10 timed iterations per stage shape (median), all threads, then ONE thread for every stage (per-core
figure).  bench.py runs a ~40 s bounded sample of the same code in-run; this script gives it the time the protocol asks for (several minutes).
usage: python scripts/cpu_baseline_protocol.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

res = bench.cpu_baseline("stage", 2, budget_s=float(os.environ.get("DLKA_CPU_BUDGET_S", "1500")))
res["protocol"] = "BASELINE.md §3: 3 warm-up + 10 timed iterations per stage, median; single-thread run of every stage (1 warm + 1 timed)"
out = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("value", "cores", "per_stage_block_s", "one_thread_block_s", "wall_s")}))
