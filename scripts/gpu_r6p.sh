#!/bin/bash
# forward offset conv from an LDS brick (cl_conv_brick3_kernel, default) against cl_igemm_kernel<0,1,3,3> (DLKA_CONV_BRICK=2: the data gradient's brick only)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r6p}; OUT=gpurun_out/$TAG; mkdir -p $OUT
K=DLKA_CONV_BRICK
for dt in f32 bf16; do
AB_TRACE_ROWS=12 timeout 900 python scripts/ab_stack_knobs.py $OUT/ab_$dt.json --dtype $dt --rounds 3 --steps 30 --trace -- s0_igemm:_stages=0,$K=2 s0_brick3:_stages=0 full_igemm:$K=2 full_brick3: 2> $OUT/ab_$dt.err | tee $OUT/ab_$dt.txt
tail -2 $OUT/ab_$dt.err
done
