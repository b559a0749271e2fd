#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8e; mkdir -p $OUT; export TMPDIR=/tmp
# A/B of the block order and the workspace pool, one process (f32 then bf16)
timeout 900 python scripts/ab_stack_knobs.py $OUT/ab_order_f32.json --rounds 3 --steps 20 -- stages2:DLKA_STACK_ORDER=stages,DLKA_STACK_WS_POOL=2 stages8:DLKA_STACK_ORDER=stages,DLKA_STACK_WS_POOL=8 unet2:DLKA_STACK_WS_POOL=2 unet4:DLKA_STACK_WS_POOL=4 unet8: unet12:DLKA_STACK_WS_POOL=12 2>&1 | grep -v Warning | tail -8
timeout 900 python scripts/ab_stack_knobs.py $OUT/ab_order_bf16.json --dtype bf16 --rounds 3 --steps 20 -- stages2:DLKA_STACK_ORDER=stages,DLKA_STACK_WS_POOL=2 unet8: 2>&1 | grep -v Warning | tail -4
timeout 300 python scripts/gx_spread.py 8 > $OUT/gx_spread.txt 2>&1; tail -22 $OUT/gx_spread.txt
DLKA_NO_XCD_SWIZZLE=1 timeout 300 python scripts/gx_spread.py 8 > $OUT/gx_spread_noxcd.txt 2>&1; tail -12 $OUT/gx_spread_noxcd.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "stack" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
