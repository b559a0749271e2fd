#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for dt in f32 bf16; do
  echo "== $dt registers"; DLKA_DW_NO_WL=1 python scripts/prof_stage.py --stage 0 --dtype $dt --trace 2>&1 | grep -E "graph fwd|rowsN_kernel<.*7, 3"
  echo "== $dt LDS weights"; python scripts/prof_stage.py --stage 0 --dtype $dt --trace 2>&1 | grep -E "graph fwd|rowsN_kernel<.*7, 3"
done
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "conv3d_cl or tokens" 2>&1 | tail -2
