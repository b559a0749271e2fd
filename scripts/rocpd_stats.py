#!/usr/bin/env python
"""Export the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) to CSV."""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
w = csv.writer(open(sys.argv[2], "w", newline=""))
w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
for r in rows:
    w.writerow([r[0], r[1], round(r[2], 1), round(r[3], 1), round(r[4], 4)])
print(f"wrote {len(rows)} kernels to {sys.argv[2]}")
