#!/bin/bash
# round-2 GPU pass c: bf16 token path (parity + bench line), full suite, fp32 bench regression check.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r3c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_gpu.log; grep -E "passed|failed|^FAILED|^E  |bf16 tokens C=|gx fixed" $OUT/pytest_gpu.log | cut -c1-400 | head -40
for dt in f32 bf16; do
echo "== bench $dt"; timeout 900 python bench.py --steps 20 --warmup 5 --dtype $dt $([ $dt = bf16 ] && echo --no-cpu-baseline) > $OUT/bench_$dt.json 2> $OUT/bench_$dt.err; echo "bench exit $?"; python - <<PY
import json
d=json.load(open("$OUT/bench_$dt.json"))
print({k:d[k] for k in ("value","ms_per_step","dtype")}, d.get("tblock",{}) and d["tblock"].get("value"))
r=d["roofline"]; print({k:r[k] for k in ("op","kernel_ms","frac","achieved")}, r.get("step"))
print(sorted(r["per_op_ms"].items(), key=lambda kv:-kv[1])[:6]); print(d["config"].get("offset_std_voxels_by_stage"))
PY
tail -3 $OUT/bench_$dt.err
done
echo "== rocprof per stage bf16"
cd /tmp
for s in 0 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_bf16_s$s -o t -- python $R/scripts/prof_stage.py --stage $s --dtype bf16 > $R/$OUT/prof_bf16_s$s.log 2>&1
  F=$(find $R/$OUT/prof_bf16_s$s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/bf16_stage${s}_block_kernel_stats.csv
  grep " ms" $R/$OUT/prof_bf16_s$s.log
done
head -14 $R/$OUT/bf16_stage0_block_kernel_stats.csv | cut -c1-140
cd $R
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -size +2M -delete
du -sh $OUT
