#!/bin/bash
# round 5, fused small-volume depthwise pair: GPU parity tests + in-process A/B of DLKA_DWPAIR (read per call) on the timed stack step, fp32 and bf16
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r9a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "dwpair or tblock or tokens" 2>&1 | tail -5 | tee $OUT/tests.log
timeout 600 python scripts/ab_stack_knobs.py $OUT/ab_dwpair_f32.json --rounds 4 --steps 30 -- pair: unfused:DLKA_DWPAIR=0 2>&1 | grep -v Warning | tail -6 | tee $OUT/ab_f32.log
timeout 600 python scripts/ab_stack_knobs.py $OUT/ab_dwpair_bf16.json --dtype bf16 --rounds 4 --steps 30 -- pair: unfused:DLKA_DWPAIR=0 2>&1 | grep -v Warning | tail -6 | tee $OUT/ab_bf16.log
