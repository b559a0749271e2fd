#!/usr/bin/env python
"""Launches of a rocprofv3 --kernel-trace CSV whose grid leaves CUs idle: per kernel name and grid, workgroups per launch, waves per launch, average duration, launches — sorted by
time spent in launches of fewer than 256 workgroups (MI355X: 256 CUs).   usage: python scripts/grid_audit.py kernel_trace.csv [min_us]"""
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        key = (r["Kernel_Name"].split("(")[0][-70:], grid // wg, wg // 64)
        a = acc[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
rows = [(v[1], k, v[0]) for k, v in acc.items() if k[1] < 256 and v[1] / v[0] >= min_us]
tot = sum(v[1] for v in acc.values())
print("launch time in the trace: %.1f ms; in launches of < 256 workgroups lasting >= %.0f us: %.1f ms" % (tot / 1e3, min_us, sum(r[0] for r in rows) / 1e3))
for t, (name, nwg, wpw), n in sorted(rows, reverse=True)[:40]:
    print("%9.1f us total  %6d launches  avg %7.1f us  %5d workgroups x %d waves  %s" % (t, n, t / n, nwg, wpw, name))
