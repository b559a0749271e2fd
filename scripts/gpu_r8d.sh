#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q -k "tblock or layernorm or batchnorm or scale_residual or mixed_bf16_real" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 600 python scripts/ab_lka2d.py $OUT/ab_lka2d.json 2>&1 | grep -v Warning | tail -22
AB_METRIC=tblock timeout 600 python scripts/ab_lka2d.py $OUT/ab_tblock.json 2>&1 | grep -v Warning | tail -4
