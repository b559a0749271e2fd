#!/bin/bash
# round-2 GPU pass b: full -m gpu suite (incl. reference-native parity + fixed-point window tests), bench (new line), per-stage rocprofv3 stats.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r3b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_gpu.log; grep -E "passed|failed|^FAILED|^E  |tokens C=|gx fixed" $OUT/pytest_gpu.log | head -40
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d.get("tblock"))
r=d["roofline"]; print({k:r[k] for k in ("op","kernel_ms","frac","achieved")}, r.get("step"))
print(sorted(r["per_op_ms"].items(), key=lambda kv:-kv[1])[:8])
print(d.get("cpu_baseline"))
PY
tail -5 $OUT/bench.err
echo "== rocprof per stage"
cd /tmp
for s in 0 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_s$s -o t -- python $R/scripts/prof_stage.py --stage $s > $R/$OUT/prof_s$s.log 2>&1
  F=$(find $R/$OUT/prof_s$s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/stage${s}_block_kernel_stats.csv
  grep " ms" $R/$OUT/prof_s$s.log
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_bench -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-tblock --no-roofline > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_bench.err
F=$(find $R/$OUT/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/bench_kernel_stats.csv && head -12 "$F" | cut -c1-150
cd $R
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -size +2M -delete
du -sh $OUT
