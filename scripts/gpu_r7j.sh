#!/bin/bash
# re-check of the fold grouping on the side stream with the round's final kernels (DLKA_STACK_FINALIZE_GROUP)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r7j}; OUT=gpurun_out/$TAG; mkdir -p $OUT
K=DLKA_STACK_FINALIZE_GROUP
timeout 600 python scripts/ab_stack_knobs.py $OUT/ab_f32.json --dtype f32 --rounds 3 --steps 30 -- fg2: fg1:$K=1 fg3:$K=3 fg0:$K=0 2> $OUT/ab.err | tee $OUT/ab_f32.txt
tail -2 $OUT/ab.err
