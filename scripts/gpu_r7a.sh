#!/bin/bash
# depthwise convs with two output planes per work-item (cl_dwconv_rows2d_kernel, default at C = 32) against one plane (DLKA_DW_TD2=0)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r7a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
K=DLKA_DW_TD2
for dt in f32 bf16; do
AB_TRACE_ROWS=16 timeout 900 python scripts/ab_stack_knobs.py $OUT/ab_$dt.json --dtype $dt --rounds 3 --steps 30 --trace -- s0_one:_stages=0,$K=0 s0_two:_stages=0 full_one:$K=0 full_two: 2> $OUT/ab_$dt.err | tee $OUT/ab_$dt.txt
tail -2 $OUT/ab_$dt.err
done
