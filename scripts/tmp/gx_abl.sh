#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for n in 0 5 6 0 5 6; do
  echo "== DLKA_GX_ABLX=$n"; DLKA_GX_ABLX=$n python scripts/prof_stage.py --stage 0 --trace 2>&1 | grep -E "gx_fx2"
done
