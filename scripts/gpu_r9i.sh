#!/bin/bash
# round 5, late: split-K width of the small stages' dense / deformable kernels (cl_igemm_pick_splits: workgroups aimed for / cap) — builds of the library, every measurement in its own process
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r9i; mkdir -p $OUT
AB_METRIC=stack timeout 1400 python scripts/ab_lka2d.py $OUT/ab_splits_f32.json - alt_lib/libdlka_w256.so alt_lib/libdlka_w384.so alt_lib/libdlka_c16.so alt_lib/libdlka_w1024c64.so 2>&1 | grep -v Warning | tail -6 | tee $OUT/ab_f32.log
