#!/bin/bash
# grad_input beside grad_offset on the library's internal stream: default (calls of >= 32768 rows fork) against never (DLKA_GX_FORK_MIN_ROWS=1000000000)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r7d}; OUT=gpurun_out/$TAG; mkdir -p $OUT
K=DLKA_GX_FORK_MIN_ROWS
for dt in f32 bf16; do
timeout 900 python scripts/ab_stack_knobs.py $OUT/ab_$dt.json --dtype $dt --rounds 3 --steps 30 -- s0_never:_stages=0,$K=1000000000 s0_fork:_stages=0 full_never:$K=1000000000 full_fork: full_all:$K=1 2> $OUT/ab_$dt.err | tee $OUT/ab_$dt.txt
tail -2 $OUT/ab_$dt.err
done
