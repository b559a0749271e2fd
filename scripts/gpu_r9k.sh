#!/bin/bash
# round 5, late: the gaps BETWEEN kernels of a replayed hipGraph — rocprofv3 kernel trace (start / end per dispatch) of one block per stage, reduced to a gap histogram
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r9k; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for s in 0 2 3; do
  DLKA_STACK_WGRAD_OVERLAP=0 DLKA_GX_FORK_MIN_ROWS=1000000000 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/p_$s -o t -- python $R/scripts/prof_stage.py --stage $s --iters 20 > $R/$OUT/p_$s.log 2>&1
  F=$(find $R/$OUT/p_$s -name "*kernel_trace.csv" | head -1)
  python - "$F" $s <<'PY' | tee $R/$OUT/gaps_stage$s.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "dlka::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]   # the replays (second half of the run)
gaps, durs = [], []
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g < 50000:
        gaps.append(g)
    durs.append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
gaps.sort()
n = len(gaps)
print("stage", sys.argv[2], "one stream, graph replays: kernels", len(rows), "gap ns: median", gaps[n // 2], "p10", gaps[n // 10], "p90", gaps[9 * n // 10], "mean", sum(gaps) // n,
      "| kernel ns: mean", sum(durs) // len(durs), "| share of gaps", round(sum(gaps) / (sum(gaps) + sum(durs)), 3))
PY
done
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
