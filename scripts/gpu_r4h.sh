#!/bin/bash
# r4h: after removing the runtime divisions from the hot loops: stamps of the forward kernel, per-stage block profiles, bench
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r4h}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== stamps"; python scripts/fwd_stamps.py 2>/dev/null | tail -9
STAGES="0:f32 1:f32 2:f32 3:f32 0:bf16" bash scripts/gpu_stage_profiles.sh $TAG "tokens or deform3d_cl or conv3d_cl or lka2d"
