#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8m; mkdir -p $OUT; export TMPDIR=/tmp
DLKA_WGRAD_COT=1 timeout 600 python scripts/time_wgrad2d.py 2>&1 | grep -v Warning | tail -8 | tee $OUT/time_wgrad2d_cot1.txt
