#!/bin/bash
# round 4, second session: where the step's time is per stage WITH the side-stream overlap, and the stack-level knobs (weight gradients of the widest
# stage in line, packed grad_offset hand-over) — all in one process on one box (scripts/ab_stack_knobs.py)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r6a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
M=DLKA_STACK_WGRAD_OVERLAP_MIN_C; O=DLKA_STACK_WGRAD_OVERLAP
timeout 900 python scripts/ab_stack_knobs.py $OUT/ab_f32.json --rounds 3 --steps 30 -- base: in64:$M=64 in128:$M=128 packed:DLKA_GOFF_PACKED=1 pk_in64:DLKA_GOFF_PACKED=1,$M=64 noov:$O=0 \
   s0:_stages=0 s0in:_stages=0,$M=64 s0pk:_stages=0,DLKA_GOFF_PACKED=1 s0no:_stages=0,$O=0 s1:_stages=1 s1no:_stages=1,$O=0 s2:_stages=2 s2no:_stages=2,$O=0 s3:_stages=3 s3no:_stages=3,$O=0 \
   2> $OUT/ab_f32.err | tee $OUT/ab_f32.txt
tail -3 $OUT/ab_f32.err
timeout 600 python scripts/ab_stack_knobs.py $OUT/ab_bf16.json --dtype bf16 --rounds 3 --steps 30 -- base: in64:$M=64 noov:$O=0 s0:_stages=0 s0in:_stages=0,$M=64 s1:_stages=1 s2:_stages=2 s3:_stages=3 \
   2> $OUT/ab_bf16.err | tee $OUT/ab_bf16.txt
tail -3 $OUT/ab_bf16.err
