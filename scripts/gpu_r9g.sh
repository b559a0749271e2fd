#!/bin/bash
# round 5, late: re-check the schedule knobs on the final tree (in-process A/B on the timed step; all of them are read per stack / per call)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r9g; mkdir -p $OUT
timeout 1200 python scripts/ab_stack_knobs.py $OUT/ab_knobs_f32.json --rounds 3 --steps 30 -- base: fork16k:DLKA_GX_FORK_MIN_ROWS=16384 fork4k:DLKA_GX_FORK_MIN_ROWS=4096 nofork:DLKA_GX_FORK_MIN_ROWS=1000000000 fin1:DLKA_STACK_FINALIZE_GROUP=1 fin3:DLKA_STACK_FINALIZE_GROUP=3 minc64:DLKA_STACK_WGRAD_OVERLAP_MIN_C=64 stages:DLKA_STACK_ORDER=stages 2>&1 | grep -v Warning | tail -10 | tee $OUT/ab_f32.log
