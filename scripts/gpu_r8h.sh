#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_nets_gpu.py tests/test_ws_canary_gpu.py -x -q -k "tblock or nets or net or canary or wrapper" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 900 python scripts/ab_tblock_overlap.py $OUT/ab_tblock_overlap.json 2>&1 | grep -v Warning | tail -5
