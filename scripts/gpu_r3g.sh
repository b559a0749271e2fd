#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r3g}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== gpu suite"; timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_gpu.log; grep -E "passed|failed|^FAILED|^E  " $OUT/pytest_gpu.log | cut -c1-300 | head -30
echo "== bench extras"; timeout 1200 python bench.py --steps 10 --warmup 3 --extras --no-cpu-baseline > $OUT/bench_extras.json 2> $OUT/bench_extras.err; echo "bench exit $?"; python - <<PY
import json
d=json.load(open("$OUT/bench_extras.json"))
for k in ("value","ms_per_step"): print(k, d[k])
for k in ("tblock","fullnet","lka2d","inference"): print(k, d.get(k))
PY
tail -3 $OUT/bench_extras.err
echo "== full net fp32 profile"
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_net -o t -- python -c "
import sys; sys.path.insert(0,'$R')
import torch, bench
print(bench.fullnet_metric(2, 3, torch.device('cuda:0')))" > $R/$OUT/prof_net.log 2>&1
grep metric $R/$OUT/prof_net.log | cut -c1-300
F=$(find $R/$OUT/prof_net -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/fullnet_f32_kernel_stats.csv && head -16 "$F" | cut -c1-150
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -size +2M -delete; du -sh $OUT
