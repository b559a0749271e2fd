#!/bin/bash
# bf16 per-stage kernel stats (where does the bf16 path lose against fp32?)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r3j}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for s in 0 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_bf16_s$s -o t -- python $R/scripts/prof_stage.py --stage $s --dtype bf16 > $R/$OUT/prof_bf16_s$s.log 2>&1
  F=$(find $R/$OUT/prof_bf16_s$s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/bf16_stage${s}_block_kernel_stats.csv
  grep " ms" $R/$OUT/prof_bf16_s$s.log
done
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -size +2M -delete; du -sh $OUT
