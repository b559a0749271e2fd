#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/ab_stack_knobs.py $OUT/ab_trail_f32.json --rounds 3 --steps 20 -- stages:DLKA_STACK_ORDER=stages unet0: unet1:DLKA_STACK_TRAIL=1 unet3:DLKA_STACK_TRAIL=3 unet6:DLKA_STACK_TRAIL=6 2>&1 | grep -v Warning | tail -6
