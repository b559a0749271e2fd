#!/bin/bash
# round 5: the fused depthwise pair at 16^3 (W = 16, two channels per workgroup): parity tests, kernel time (rocprofv3 of one stage-1 block), in-process A/B on the timed step
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r9d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "dwpair" 2>&1 | tail -3 | tee $OUT/tests.log
bash scripts/gpu_r9c.sh "dwpair|dwconv_rows" "1" f32 | tee $OUT/stage1.log
timeout 600 python scripts/ab_stack_knobs.py $OUT/ab_dwpair_f32.json --rounds 4 --steps 30 -- pair: unfused:DLKA_DWPAIR=0 2>&1 | grep -v Warning | tail -3 | tee $OUT/ab_f32.log
timeout 600 python scripts/ab_stack_knobs.py $OUT/ab_dwpair_bf16.json --dtype bf16 --rounds 4 --steps 30 -- pair: unfused:DLKA_DWPAIR=0 2>&1 | grep -v Warning | tail -3 | tee $OUT/ab_bf16.log
