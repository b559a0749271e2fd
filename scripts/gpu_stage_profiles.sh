#!/bin/bash
# parity subset + rocprofv3 per-kernel stats of one block per stage (fp32 stages 0-3, bf16 stage 0/1) + the bench line.   usage: gpu_stage_profiles.sh TAG [pytest -k expr]
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-prof}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
KEXPR=${2:-"tokens or deform3d_cl or bf16"}
echo "== gpu tests (subset: $KEXPR)"; timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "$KEXPR" > $OUT/pytest_sub.log 2>&1; echo "exit $?"; tail -3 $OUT/pytest_sub.log
cd /tmp
for cfg in ${STAGES:-0:f32 1:f32 2:f32 3:f32 0:bf16 1:bf16}; do
  s=${cfg%%:*}; dt=${cfg##*:}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${s}_$dt -o t -- python $R/scripts/prof_stage.py --stage $s --dtype $dt > $R/$OUT/p_${s}_$dt.log 2>&1
  F=$(find $R/$OUT/p_${s}_$dt -name "*kernel_stats.csv" | head -1)
  echo "stage $s $dt: $(grep ' ms' $R/$OUT/p_${s}_$dt.log)"
  [ -n "$F" ] && cp $F $R/$OUT/${dt}_stage${s}_block_kernel_stats.csv
done
cd $R
python - <<PY
import csv,glob,os
for f in sorted(glob.glob("$OUT/*_block_kernel_stats.csv")):
    rows=[r for r in csv.DictReader(open(f)) if "dlka::" in r["Name"]]
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    print("--", os.path.basename(f))
    for r in rows[:10]:
        n=r["Name"].replace("void dlka::","").replace("dlka::","").split("(")[0]
        print("   %-62s x%-2d %8.1f us"%(n[:62], int(r["Calls"])//22, float(r["AverageNs"])/1e3))
PY
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline --no-tblock > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "exit $?"; head -8 $OUT/bench_f32.err
python -c "
import json; d=json.load(open('$OUT/bench_f32.json')); print(d['value'], d['ms_per_step'], d['other_dtype'])"
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; du -sh $OUT
