#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for s in 0 1; do python scripts/prof_stage.py --stage $s --trace 2>&1 | grep -E "graph fwd|launch trace|gx_|goff|fwd16|deform_fwd"; done
python scripts/prof_stage.py --stage 0 --dtype bf16 --trace 2>&1 | grep -E "graph fwd|gx_"
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "gx or deform3d_cl or tokens" 2>&1 | tail -2
