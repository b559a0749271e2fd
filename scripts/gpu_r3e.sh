#!/bin/bash
# round-2 GPU pass e: 2-D fast path (parity at the real shapes, images/s, rocprof), full-net fp32 profile (what makes the plumbing slow).
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r3f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== 2d tests"; timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_nets_gpu.py -m gpu -q -s -k "lka2d or decoder2d or golden" > $OUT/pytest_2d.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_2d.log; grep -E "passed|failed|^FAILED|^E  |lka2d C=" $OUT/pytest_2d.log | cut -c1-420 | head -30
echo "== 2d metric + rocprof"
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_2d -o t -- python -c "
import sys; sys.path.insert(0,'$R')
import torch, bench
print(bench.lka2d_metric(5, torch.device('cuda:0')))" > $R/$OUT/prof_2d.log 2>&1
grep metric $R/$OUT/prof_2d.log | cut -c1-400
F=$(find $R/$OUT/prof_2d -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/lka2d_kernel_stats.csv && head -14 "$F" | cut -c1-170
echo "== full net fp32 profile"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_net -o t -- python -c "
import sys; sys.path.insert(0,'$R')
import torch, bench
print(bench.fullnet_metric(2, 2, torch.device('cuda:0')))" > $R/$OUT/prof_net.log 2>&1
grep metric $R/$OUT/prof_net.log | cut -c1-300
F=$(find $R/$OUT/prof_net -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/fullnet_f32_kernel_stats.csv && head -14 "$F" | cut -c1-200
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -size +2M -delete; du -sh $OUT
