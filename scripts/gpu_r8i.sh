#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8i; mkdir -p $OUT; export TMPDIR=/tmp
DLKA_TBLOCK_WGRAD_OVERLAP=0 timeout 300 python scripts/debug_overlap3.py retain 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3 | tee $OUT/debug3_no_overlap.txt
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_nets_gpu.py -x -q -k "wgrad_overlap or phased or nets or net" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
