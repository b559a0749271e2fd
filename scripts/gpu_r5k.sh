#!/bin/bash
# round 4, call k: per-kernel tables of the wrapper block (TransformerBlock_3D_single_deform_LKA fwd+bwd) at the four stages
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5k; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for s in 0 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_s$s -o t -- python $R/scripts/prof_tblock.py --stage $s --iters 12 > $R/$OUT/p_s$s.log 2>&1
  F=$(find $R/$OUT/p_s$s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/tblock_stage${s}_kernel_stats.csv
done
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
python - <<PY
import csv
for s in range(4):
    rows=list(csv.DictReader(open("$OUT/tblock_stage%d_kernel_stats.csv"%s)))
    tot=sum(int(r['TotalDurationNs']) for r in rows)/12/1e3
    print("stage",s,"sum of kernel time per fwd+bwd: %.0f us"%tot)
    for r in rows[:16]:
        print("   %-88s x%-4.1f %7.1f us"%(r['Name'].replace('void dlka::','').split('(')[0][:88], int(r['Calls'])/12, float(r['AverageNs'])/1e3))
PY
