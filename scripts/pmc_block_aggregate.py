#!/usr/bin/env python
"""Aggregate the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_block.sh into HBM bytes per launch of every kernel of a stage block.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes for wide streaming reads (MI355X_MICROARCH.md §HBM),
so the read side is doubled:  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Keys are the short kernel names bench.py prints."""
import csv, glob, json, os, sys

root, rnd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "?")
out = {"_meta": {"round": rnd, "made_by": "scripts/pmc_block.sh + scripts/pmc_block_aggregate.py", "correction": "read side x2 (gfx950 FETCH_SIZE, guide §HBM)",
                 "workload": "scripts/prof_stage.py: one block of the stage, forward + backward, hipGraph replays"}}


def short(name):
    return name.replace("void dlka::", "").replace("dlka::", "").split("(")[0]


for sdir in sorted(glob.glob(os.path.join(root, "stage*_*"))):
    if not os.path.isdir(sdir):
        continue
    acc = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(sdir, ctr, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") != ctr or "dlka::" not in r["Kernel_Name"]:
                    continue
                e = acc.setdefault(short(r["Kernel_Name"]), {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
                e[ctr][0] += float(r["Counter_Value"])
                e[ctr][1] += 1
    res = {}
    for name, e in acc.items():
        nf, nw = max(e["FETCH_SIZE"][1], 1), max(e["WRITE_SIZE"][1], 1)
        fetch, write = e["FETCH_SIZE"][0] / nf, e["WRITE_SIZE"][0] / nw
        res[name] = {"launches": max(e["FETCH_SIZE"][1], e["WRITE_SIZE"][1]), "fetch_kib": round(fetch, 1), "write_kib": round(write, 1),
                     "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
    out[os.path.basename(sdir)] = res
print(json.dumps(out, indent=1))
