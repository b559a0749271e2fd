#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8l; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_ref_d3d_2d_gpu.py -x -q -k "lka2d or tokens or lka3d or tblock3d_vs or stack" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 600 python scripts/ab_lka2d.py $OUT/ab_lka2d.json 2>&1 | grep -v Warning | tail -14
timeout 900 python scripts/ab_stack_knobs.py $OUT/ab_pad_f32.json --rounds 3 --steps 20 --trace -- pad: nopad:DLKA_WGRAD_PAD=0 2>&1 | grep -v Warning | grep "wgrad_dense\|pad_copy\|median\|sum ms" | head -20
timeout 900 python scripts/ab_stack_knobs.py $OUT/ab_pad_bf16.json --dtype bf16 --rounds 3 --steps 20 -- pad: nopad:DLKA_WGRAD_PAD=0 2>&1 | grep -v Warning | tail -2
