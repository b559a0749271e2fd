#!/bin/bash
# rocprofv3 kernel table of the full-net training iteration (bench.fullnet_metric, batch 2, fp32): where the plumbing around the D-LKA blocks goes.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-netprof}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_net -o t -- python -c "
import sys; sys.path.insert(0,'$R')
import torch, bench
print(bench.fullnet_metric(2, 4, torch.device('cuda:0')))" > $R/$OUT/prof_net.log 2>&1
grep -i "metric\|Error" $R/$OUT/prof_net.log | cut -c1-400
F=$(find $R/$OUT/prof_net -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/fullnet_f32_kernel_stats.csv && head -40 "$F" | cut -c1-170
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -size +2M -delete; du -sh $OUT
