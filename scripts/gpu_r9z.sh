#!/bin/bash
# final tree: rocprofv3 --kernel-trace --stats table of the bench step (every kernel on ONE stream: the durations are the kernels' own), and of the default two-stream step
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r9z; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
DLKA_STACK_WGRAD_OVERLAP=0 DLKA_GX_FORK_MIN_ROWS=1000000000 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/bench1s -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tblock --no-lka2d --no-fullnet --no-companion --no-roofline > $R/$OUT/bench_one_stream.json 2> $R/$OUT/bench_one_stream.err
F=$(find $R/$OUT/bench1s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/$OUT/bench_one_stream_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/bench2s -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tblock --no-lka2d --no-fullnet --no-companion > $R/$OUT/bench_default.json 2> $R/$OUT/bench_default.err
F=$(find $R/$OUT/bench2s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/$OUT/bench_default_kernel_stats.csv
cd $R
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete; du -sh $OUT
head -5 $OUT/bench_one_stream_kernel_stats.csv | cut -c1-160
