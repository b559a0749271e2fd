#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r4k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "conv3d_vs_aten or groupnorm or instancenorm or batchnorm" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 200 python scripts/time_conv3.py 2>&1 | tee $OUT/time_conv3.log | tail -12
timeout 300 python -c "
import torch, bench
print(bench.fullnet_metric(2, 6, torch.device('cuda:0')))" 2>&1 | tail -1 | cut -c1-330 | tee $OUT/fullnet.log
