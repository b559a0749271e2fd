#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8p; mkdir -p $OUT; export TMPDIR=/tmp
for l in - alt_lib/libdlka_hip_oldgx3.so; do echo "== $l"; timeout 300 python scripts/time_ddw2d_gx.py $l 2>&1 | grep -v Warning | tail -5; done | tee $OUT/time_ddw2d_gx.txt
