#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8k; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python scripts/ab_lka2d.py $OUT/ab_lka2d.json 2>&1 | grep -v Warning | tail -16
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_ref_d3d_2d_gpu.py tests/test_nets_gpu.py -x -q -k "lka2d or ddw2d or deform2d or dwconv2d or 2d or nets or net" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
