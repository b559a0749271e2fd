#!/usr/bin/env python
"""GPU idle time inside the timed step, from a rocprofv3 --kernel-trace CSV of the bench command: per step (delimited by cl_prep_table_kernel launches) the wall time, the union of the
kernels' busy intervals, the idle remainder and how it splits into gaps (count, total, largest; which kernels sit on either side of the largest ones), and the busy time per queue.
usage: python scripts/timeline_idle.py kernel_trace.csv"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if "cl_prep_table_kernel" in r[2]]
# the job table is split in two launches per step (first stage / the rest): a step starts at a prep launch that follows a non-prep kernel
step_starts = [i for k, i in enumerate(starts) if k == 0 or starts[k - 1] != i - 1 and "prep_table" not in rows[i - 1][2]]
print("kernels", len(rows), "steps", len(step_starts))
res = []
for a, b in zip(step_starts[-8:-1], step_starts[-7:]):
    seg = rows[a:b]
    t0, t1 = seg[0][0], rows[b][0]
    busy, gaps, cur_end, prev = 0, [], seg[0][0], None
    for s, e, n, q in seg:
        if s > cur_end:
            gaps.append((s - cur_end, prev, n))
            busy += 0
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
            prev = n
    if t1 > cur_end:
        gaps.append((t1 - cur_end, prev, "(next step)"))
    perq = collections.Counter()
    for s, e, n, q in seg:
        perq[q] += e - s
    res.append((t1 - t0, busy, gaps, perq, len(seg)))
for wall, busy, gaps, perq, n in res:
    g = sorted(gaps, reverse=True)
    print("step: wall %.3f ms, busy (union) %.3f ms, idle %.3f ms in %d gaps (>=2 us: %d, sum %.3f ms; >=5 us: %d, sum %.3f ms); launches %d; kernel time per queue (ms): %s" % (
        wall / 1e6, busy / 1e6, (wall - busy) / 1e6, len(gaps), sum(1 for x in gaps if x[0] >= 2000), sum(x[0] for x in gaps if x[0] >= 2000) / 1e6,
        sum(1 for x in gaps if x[0] >= 5000), sum(x[0] for x in gaps if x[0] >= 5000) / 1e6, n, {k: round(v / 1e6, 3) for k, v in perq.items()}))
wall, busy, gaps, perq, n = res[-1]
print("largest gaps of the last step (us, after -> before):")
for d, a, b in sorted(gaps, reverse=True)[:12]:
    print("  %7.1f  %s  ->  %s" % (d / 1e3, (a or "")[:60], (b or "")[:60]))
hist = collections.Counter(min(int(d / 1000), 10) for d, _, _ in gaps)
print("gap histogram (us bucket: count):", dict(sorted(hist.items())))

# per queue of the last full step: kernel time by name, the queue's own gaps (a queue that waits for another one shows it here)
a, b = step_starts[-2], step_starts[-1]
seg = rows[a:b]
byq = collections.defaultdict(list)
for r in seg:
    byq[r[3]].append(r)
for q, rs in sorted(byq.items()):
    rs.sort()
    names = collections.Counter()
    for s_, e_, n_, _ in rs:
        names[n_.split("(")[0][-70:]] += e_ - s_
    gaps = [(rs[i + 1][0] - rs[i][1], rs[i][2].split("(")[0][-48:], rs[i + 1][2].split("(")[0][-48:]) for i in range(len(rs) - 1)]
    span = rs[-1][1] - rs[0][0]
    print("queue %s: %d launches, span %.3f ms, kernel time %.3f ms, gaps %.3f ms (>= 5 us: %d, %.3f ms)" % (q, len(rs), span / 1e6, sum(e_ - s_ for s_, e_, _, _ in rs) / 1e6,
          sum(max(g[0], 0) for g in gaps) / 1e6, sum(1 for g in gaps if g[0] >= 5000), sum(g[0] for g in gaps if g[0] >= 5000) / 1e6))
    for n_, t_ in names.most_common(8):
        print("      %8.1f us  %s" % (t_ / 1e3, n_))
    agg = collections.Counter()
    for d, x, y in gaps:
        if d >= 3000:
            agg[(x, y)] += d
    for (x, y), d in agg.most_common(8):
        print("   gap %8.1f us total  %s -> %s" % (d / 1e3, x, y))
