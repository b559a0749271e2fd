#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for n in 0 3 5 7 9 0; do
  echo "== DLKA_GX_STAGGER=$n"; DLKA_GX_STAGGER=$n python scripts/prof_stage.py --stage 0 --trace 2>&1 | grep -E "graph fwd|gx_fx2|gx_gather"
done
