#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_lka2d.py $OUT/ab_lka2d.json - alt_lib/libdlka_hip_head.so alt_lib/libdlka_hip_prev.so 2>&1 | grep -v Warning | tail -4
