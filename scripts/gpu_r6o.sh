#!/bin/bash
# dense weight gradient: one 18-row window for the three w-taps of a wave (default) against per-tap rows (DLKA_WGRAD_WIN3=0); stage stacks + full step, both dtypes
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r6o}; OUT=gpurun_out/$TAG; mkdir -p $OUT
K=DLKA_WGRAD_WIN3
for dt in f32 bf16; do
AB_TRACE_ROWS=30 timeout 900 python scripts/ab_stack_knobs.py $OUT/ab_$dt.json --dtype $dt --rounds 3 --steps 30 -- s0_tap:_stages=0,$K=0 s0_win:_stages=0 s1_tap:_stages=1,$K=0 s1_win:_stages=1 full_tap:$K=0 full_win: 2> $OUT/ab_$dt.err | tee $OUT/ab_$dt.txt
tail -2 $OUT/ab_$dt.err
done
