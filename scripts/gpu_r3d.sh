#!/bin/bash
# round-2 GPU pass d: net-level tests (full D_LKA_Former, 2-D decoder, sliding window), bench extras, bf16 parity thresholds.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r3d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== new gpu tests"; timeout 1500 python -m pytest tests/test_nets_gpu.py tests/test_parity_gpu.py -m gpu -q -k "nets or decoder2d or plumbing or full_net or sliding or bf16" > $OUT/pytest_new.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_new.log; grep -E "passed|failed|^FAILED|^E  " $OUT/pytest_new.log | cut -c1-300 | head -30
echo "== bench extras"; timeout 1200 python bench.py --steps 10 --warmup 3 --extras --no-cpu-baseline > $OUT/bench_extras.json 2> $OUT/bench_extras.err; echo "bench exit $?"; python - <<PY
import json
d=json.load(open("$OUT/bench_extras.json"))
for k in ("value","ms_per_step"): print(k, d[k])
for k in ("tblock","fullnet","lka2d","inference"): print(k, d.get(k))
PY
tail -5 $OUT/bench_extras.err
echo "== bench bf16 extras"; timeout 600 python bench.py --steps 5 --warmup 2 --extras --no-cpu-baseline --dtype bf16 > $OUT/bench_bf16_extras.json 2> $OUT/bench_bf16_extras.err; python -c "
import json; d=json.load(open('$OUT/bench_bf16_extras.json')); print(d['value'], d.get('fullnet'))"
echo "== rocprof 2d"
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_2d -o t -- python -c "
import sys; sys.path.insert(0,'$R')
import torch, bench
print(bench.lka2d_metric(3, torch.device('cuda:0')))" > $R/$OUT/prof_2d.log 2>&1
F=$(find $R/$OUT/prof_2d -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/lka2d_kernel_stats.csv && head -12 "$F" | cut -c1-150
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -size +2M -delete; du -sh $OUT
