#!/bin/bash
# round 4, call j: PMC traffic per kernel of the stage blocks at HEAD (the file bench.py quotes), then the isolation protocol (scripts/gpu_isolation.sh)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5j; mkdir -p $OUT; export TMPDIR=/tmp
export DLKA_STACK_WGRAD_OVERLAP=0
ROUND=r05 bash scripts/pmc_block.sh r5j/pmcb "0 1" "f32" > $OUT/pmc_block.log 2>&1; tail -14 $OUT/pmc_block.log
ROUND=r05 bash scripts/pmc_block.sh r5j/pmcb_bf16 "0" "bf16" > $OUT/pmc_block_bf16.log 2>&1; tail -6 $OUT/pmc_block_bf16.log
unset DLKA_STACK_WGRAD_OVERLAP
LOOPS=6 bash scripts/gpu_isolation.sh r5j/iso 2>&1 | tail -16
