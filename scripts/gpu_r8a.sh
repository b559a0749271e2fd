#!/bin/bash
# round 5, first call: the fork-context tests, the mixed-mode parity change, then the driver line as a baseline on this box
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "fork or mixed_bf16_real or canary" -s > $OUT/pytest_fork.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_fork.log
tail -5 $OUT/pytest_fork.log
timeout 600 python -m pytest tests/test_ws_canary_gpu.py -x -q > $OUT/pytest_canary.log 2>&1; tail -2 $OUT/pytest_canary.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
