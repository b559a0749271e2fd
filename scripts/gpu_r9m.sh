#!/bin/bash
# the kink-exposed gradient tests, five times on one box (their outcome used to depend on which element sat within rounding of LeakyReLU's kink in that run)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r9m; mkdir -p $OUT
for i in 1 2 3 4 5; do timeout 900 python -m pytest tests/test_nets_gpu.py tests/test_parity_gpu.py -q -m gpu -k "assembled_net or tblock" 2>&1 | tail -2; done | tee $OUT/loops.log
