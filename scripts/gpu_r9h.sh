#!/bin/bash
# round 5, late: fold grouping of the weight-gradient partial sums on the side stream, 1 vs 2 blocks per launch (two entries each: the spread between equal configurations is the noise)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r9h; mkdir -p $OUT
timeout 1200 python scripts/ab_stack_knobs.py $OUT/ab_fin_f32.json --rounds 5 --steps 30 -- base: fin1:DLKA_STACK_FINALIZE_GROUP=1 base_again: fin1_again:DLKA_STACK_FINALIZE_GROUP=1 2>&1 | grep -v Warning | tail -5 | tee $OUT/ab_f32.log
timeout 1200 python scripts/ab_stack_knobs.py $OUT/ab_fin_bf16.json --dtype bf16 --rounds 5 --steps 30 -- base: fin1:DLKA_STACK_FINALIZE_GROUP=1 base_again: fin1_again:DLKA_STACK_FINALIZE_GROUP=1 2>&1 | grep -v Warning | tail -5 | tee $OUT/ab_bf16.log
