#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8q; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python scripts/ab_lka2d.py $OUT/ab_lka2d_tapsplit.json - alt_lib/libdlka_hip_head.so 2>&1 | grep -v Warning | tail -2 | cut -c1-330
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_ref_d3d_2d_gpu.py -x -q -k "lka2d or ddw2d or deform2d or dwconv2d or 2d" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
