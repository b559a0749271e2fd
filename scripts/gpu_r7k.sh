#!/bin/bash
# one-call backward (nn.Module path): the block's weight gradients on the library's side stream, forked as their inputs appear, joined at the end of the call (DLKA_SIDE_STREAM=1)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r7k}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for v in 0 1; do
  if [ $v = 1 ]; then export DLKA_SIDE_STREAM=1; else unset DLKA_SIDE_STREAM; fi
  python - <<'PY' 2>/dev/null | tee -a $OUT/tblock.txt
import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
r = bench.tblock_metric(2, 10, 3, torch.device("cuda", 0))
print("side_stream", os.environ.get("DLKA_SIDE_STREAM"), r["value"], r["ms_per_step"], (r.get("hipgraph") or {}).get("value"))
PY
done
