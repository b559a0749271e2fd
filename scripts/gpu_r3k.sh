#!/bin/bash
# bf16 polish: paired dword accesses in the pointwise kernel, dword G / bf16 S in the stored-sample weight gradient, wave-granular offset conv forward
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r3k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "tokens or tblock or stored_samples or bf16 or pointwise or conv3d_cl" > $OUT/pytest_sel.log 2>&1; echo "exit $?"; grep -E "passed|failed|^FAILED|^E  " $OUT/pytest_sel.log | cut -c1-300 | head
for dt in f32 bf16; do
  echo "== bench $dt"; timeout 900 python bench.py --steps 20 --warmup 5 --dtype $dt --no-cpu-baseline --no-tblock > $OUT/bench_$dt.json 2> $OUT/bench_$dt.err; python - <<PY
import json
d=json.load(open("$OUT/bench_$dt.json")); r=d["roofline"]
print("$dt", d["value"], d["ms_per_step"], sorted(r["per_op_ms"].items(), key=lambda kv:-kv[1])[:6])
PY
done
cd /tmp
for s in 0 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_bf16_s$s -o t -- python $R/scripts/prof_stage.py --stage $s --dtype bf16 > $R/$OUT/prof_bf16_s$s.log 2>&1
  F=$(find $R/$OUT/prof_bf16_s$s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/bf16_stage${s}_block_kernel_stats.csv
  grep " ms" $R/$OUT/prof_bf16_s$s.log
done
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -size +2M -delete; du -sh $OUT
