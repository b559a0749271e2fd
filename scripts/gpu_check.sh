#!/bin/bash
# Runs on the GPU box via gpurun: parity tests, smoke, bench, rocprofv3 kernel trace.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r01}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== rocm-smi" > $OUT/env.log; (rocm-smi --showproductname 2>&1 | head -20; nproc; free -g | head -2) >> $OUT/env.log 2>&1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
cat $OUT/bench.json; tail -25 $OUT/bench.err
echo "== rocprofv3 kernel trace"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o trace -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-graph --no-cpu-baseline > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof_bench.err"); echo "rocprof exit $?"
find $OUT/prof -name "*stats*" | head; 
F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -40 "$F"
# keep the merge-back small: drop the big raw trace, keep the stats
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
