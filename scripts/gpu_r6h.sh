#!/bin/bash
# LDS-brick data gradient of the offset conv (cl_conv_brick.hip) against the kernel it replaces (DLKA_CONV_BRICK=0), one process, both dtypes
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r6h}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for dt in f32 bf16; do
AB_TRACE_ROWS=14 timeout 900 python scripts/ab_stack_knobs.py $OUT/ab_$dt.json --dtype $dt --rounds 3 --steps 30 --trace -- s0_wave:_stages=0,DLKA_CONV_BRICK=0 s0_brick:_stages=0 full_wave:DLKA_CONV_BRICK=0 full_brick: 2> $OUT/ab_$dt.err | tee $OUT/ab_$dt.txt
tail -2 $OUT/ab_$dt.err
done
