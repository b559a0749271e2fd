#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8t; mkdir -p $OUT; export TMPDIR=/tmp
FILES=$(ls tests/test_*gpu*.py tests/test_parity_gpu.py 2>/dev/null | sort -u)
( time timeout 1500 python -m pytest $(echo $FILES | tr ' ' '\n' | sort -r | tr '\n' ' ') -q -m gpu -p no:cacheprovider ) > $OUT/suite_reverse.log 2>&1; tail -4 $OUT/suite_reverse.log
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_parity_gpu.py -q -m gpu -p no:cacheprovider -k "fork or wgrad_overlap or hipgraph_replay or stack_prepare" 2>&1 | tail -1; done | tee $OUT/loops.log
