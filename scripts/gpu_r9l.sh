#!/bin/bash
# round 5: the net-level gradient test failed once on the final tree (2.7e-2 on one conv51 weight of the 8^3 stage): kink or kernel?  + the whole GPU suite without -x
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r9l; mkdir -p $OUT
timeout 600 python scripts/debug_net_kink.py 2>&1 | grep -v Warning | tail -5 | tee $OUT/kink_default.txt
timeout 600 python scripts/debug_net_kink.py DLKA_DWPAIR=0 2>&1 | grep -v Warning | tail -5 | tee $OUT/kink_nopair.txt
timeout 600 python scripts/debug_net_kink.py DLKA_PREP_TILED=0 2>&1 | grep -v Warning | tail -5 | tee $OUT/kink_noprep.txt
( time timeout 2400 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1; tail -8 $OUT/pytest_gpu.log
