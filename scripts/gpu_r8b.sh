#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8b; mkdir -p $OUT; export TMPDIR=/tmp
for cfg in "cur" "cur poison" "prev" "prev poison"; do
  timeout 300 python scripts/debug_mixed.py $cfg 2>&1 | grep -v Warning | tail -8
done > $OUT/mixed.log 2>&1
DLKA_MIXED_XN32=0 timeout 300 python scripts/debug_mixed.py cur 2>&1 | tail -6 >> $OUT/mixed.log
cat $OUT/mixed.log
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "fork" -s > $OUT/pytest_fork.log 2>&1; tail -5 $OUT/pytest_fork.log
