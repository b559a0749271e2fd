#!/bin/bash
# r4g: tblock tests at the contract's tolerances; stage-0 profile with dw7 at 3 waves/SIMD; packed grad_offset hand-over (DLKA_GOFF_PACKED=1) re-measured
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r4g}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== tblock tests"; timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -s -k "tblock" > $OUT/pytest_tblock.log 2>&1; echo "exit $?"; grep -E "passed|failed|AssertionError|tblock C=" $OUT/pytest_tblock.log | cut -c1-300 | tail
cd /tmp
for v in default packed; do
  if [ $v = packed ]; then export DLKA_GOFF_PACKED=1; else unset DLKA_GOFF_PACKED; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_$v -o t -- python $R/scripts/prof_stage.py --stage 0 --dtype f32 > $R/$OUT/p_$v.log 2>&1
  F=$(find $R/$OUT/p_$v -name "*kernel_stats.csv" | head -1); cp $F $R/$OUT/${v}_stage0_f32_kernel_stats.csv
  echo "$v: $(grep ' ms' $R/$OUT/p_$v.log)"
done
unset DLKA_GOFF_PACKED
cd $R
python - <<PY
import csv
for v in ("default","packed"):
    rows=[r for r in csv.DictReader(open("$OUT/%s_stage0_f32_kernel_stats.csv"%v)) if "dlka::" in r["Name"]]
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    print("--", v)
    for r in rows[:14]:
        n=r["Name"].replace("void dlka::","").replace("dlka::","").split("(")[0]
        print("   %-62s x%-2d %8.1f us"%(n[:62], int(r["Calls"])//22, float(r["AverageNs"])/1e3))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; du -sh $OUT
