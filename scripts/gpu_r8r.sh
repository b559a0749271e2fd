#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8r; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/fuzz_lka2d.py 30 7 2>&1 | grep -v Warning | tail -34 | tee $OUT/fuzz_lka2d.txt
