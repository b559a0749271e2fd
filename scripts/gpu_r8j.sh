#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8j; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/pmc_lka2d.sh r8j_pmc2d 2>&1 | tail -26
ROUND="r08 (round 5)" bash scripts/pmc_block.sh r8j_pmc3d "0 1" "f32" 2>&1 | tail -26
ROUND="r08 (round 5)" bash scripts/pmc_block.sh r8j_pmc3d_bf16 "0" "bf16" 2>&1 | tail -14
cd $R
timeout 900 python -m pytest tests/test_nets_gpu.py -x -q > $OUT/pytest_nets.log 2>&1; tail -3 $OUT/pytest_nets.log
