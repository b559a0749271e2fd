#!/bin/bash
# cl_conv_brick tile candidates (DLKA_CONV_BRICK_WAVES=4 | 8 | 42) on the stage-0 stack, one process
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r6j}; OUT=gpurun_out/$TAG; mkdir -p $OUT
W=DLKA_CONV_BRICK_WAVES
AB_TRACE_ROWS=10 timeout 900 python scripts/ab_stack_knobs.py $OUT/ab_f32.json --dtype f32 --rounds 3 --steps 30 -- s0_wave:_stages=0,DLKA_CONV_BRICK=0 s0_b8:_stages=0,$W=8 s0_b4:_stages=0,$W=4 s0_b8td2:_stages=0,$W=8,DLKA_CONV_BRICK_TD=2 s0_b8td4:_stages=0,$W=8,DLKA_CONV_BRICK_TD=4 full_b8:$W=8 full_b8td2:$W=8,DLKA_CONV_BRICK_TD=2 2> $OUT/ab_f32.err | tee $OUT/ab_f32.txt
tail -2 $OUT/ab_f32.err
