#!/bin/bash
# 2-D block: the offset nets' weight gradients on the internal stream beside the data chain (default) against one stream (DLKA_LKA2D_FORK=0)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r7h}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -k "lka2d or 2d or canary" > $OUT/pytest_2d.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/pytest_2d.log
python - <<'PY'
import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = torch.device("cuda", 0)
for rnd in range(2):
    for v in ("0", "1", "2"):   # 0: one stream, 1: the offset nets' weight gradients on the internal stream, 2 (= unset): + grad_input beside grad_offset
        os.environ["DLKA_LKA2D_FORK"] = v
        r = bench.lka2d_metric(8, dev, torch.bfloat16)
        print("fork", v, json.dumps({k: r[k] for k in ("value", "ms_per_step", "ms_per_block_fwd_bwd")} if r else None))
PY
