#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5g; mkdir -p $OUT; export TMPDIR=/tmp
DLKA_PARITY_VERBOSE=1 timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "mixed_bf16" -s > $OUT/mixed.log 2>&1
grep -E "passed|failed|^FAILED|AssertionError|worst" $OUT/mixed.log | cut -c1-330 | head -30
