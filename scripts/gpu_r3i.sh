#!/bin/bash
# sample hand-over (grad_offset -> weight gradient): tests, A/B bench, per-stage kernel stats
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r3i}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "tokens or tblock or stored_samples" > $OUT/pytest_sel.log 2>&1; echo "exit $?"; grep -E "passed|failed|^FAILED|^E  " $OUT/pytest_sel.log | cut -c1-300 | head
for mode in samp gather; do
  if [ $mode = gather ]; then export DLKA_WGRAD_GATHER=1; else unset DLKA_WGRAD_GATHER; fi
  for dt in f32 bf16; do
    echo "== bench $mode $dt"; timeout 900 python bench.py --steps 20 --warmup 5 --dtype $dt --no-cpu-baseline --no-tblock > $OUT/bench_${mode}_$dt.json 2> $OUT/bench_${mode}_$dt.err; python - <<PY
import json
d=json.load(open("$OUT/bench_${mode}_$dt.json")); r=d["roofline"]
print("$mode $dt", d["value"], d["ms_per_step"], sorted(r["per_op_ms"].items(), key=lambda kv:-kv[1])[:6])
PY
  done
done
unset DLKA_WGRAD_GATHER
cd /tmp
for s in 0 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_s$s -o t -- python $R/scripts/prof_stage.py --stage $s > $R/$OUT/prof_s$s.log 2>&1
  F=$(find $R/$OUT/prof_s$s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/stage${s}_block_kernel_stats.csv
  grep " ms" $R/$OUT/prof_s$s.log
done
head -14 $R/$OUT/stage0_block_kernel_stats.csv | cut -c1-150
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete; find $OUT -name "*.db" -size +2M -delete; du -sh $OUT
