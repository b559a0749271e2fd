#!/bin/bash
# round 5, late: kernel-choice thresholds re-checked on the final tree (in-process A/B, knobs read per call)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r9j; mkdir -p $OUT
timeout 1400 python scripts/ab_stack_knobs.py $OUT/ab_thresh_f32.json --rounds 3 --steps 30 -- base: fwd16s1:DLKA_FWD16_MIN_ROWS=8192 goff16s1:DLKA_GOFF16_MIN_ROWS=8192 win3off:DLKA_WGRAD_WIN3=0 pwkw1:DLKA_PW_KW=1 packed:DLKA_GOFF_PACKED=1 brickoff:DLKA_CONV_BRICK=0 dwlds:DLKA_DW_LDS=1 base2: 2>&1 | grep -v Warning | tail -10 | tee $OUT/ab_f32.log
