#!/bin/bash
# round 4, call i: whole GPU suite, fp32 stage profiles with / without the split-bf16 backward contractions, the default bench line
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${1:-r5i}; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest -m gpu (driver order)"
timeout 1800 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "exit $?"; tail -4 $OUT/pytest_gpu.log
echo "== bench (default)"
timeout 900 python bench.py > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "exit $?"
python - <<PY
import json
d=json.load(open("$OUT/bench_f32.json"))
print(d["value"], d["ms_per_step"], "bf16:", (d.get("other_dtype") or {}).get("value"), "tblock:", (d.get("tblock") or {}).get("value"), "lka2d:", (d.get("lka2d") or {}).get("value"), "roof:", d["roofline"]["kernel"], d["roofline"]["frac"])
PY
export DLKA_STACK_WGRAD_OVERLAP=0
cd /tmp
for b16 in 1 0; do for s in 0 1 2 3; do
  DLKA_DEFORM_B16=$b16 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${b16}_s$s -o t -- python $R/scripts/prof_stage.py --stage $s --dtype f32 > $R/$OUT/p_${b16}_s$s.log 2>&1
  F=$(find $R/$OUT/p_${b16}_s$s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/f32_b16_${b16}_stage${s}_block_kernel_stats.csv
  echo "b16=$b16 stage $s: $(grep ' ms' $R/$OUT/p_${b16}_s$s.log | tail -1)"
done; done
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
python - <<PY
import csv
for s in range(4):
    for b in (0,1):
        for r in csv.DictReader(open("$OUT/f32_b16_%d_stage%d_block_kernel_stats.csv"%(b,s))):
            n=r['Name']
            if 'goff' in n or 'gx_fx2' in n or 'gx_kernel' in n: print(s, b, n.split('(')[0].replace('void dlka::','')[:64], round(float(r['AverageNs'])/1e3,1))
PY
