#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r3l}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for mode in reg nkc0; do
  if [ $mode = nkc0 ]; then export DLKA_GOFF_NKC0=1; else unset DLKA_GOFF_NKC0; fi
  for dt in f32 bf16; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_${mode}_$dt -o t -- python $R/scripts/prof_stage.py --stage 0 --dtype $dt > $R/$OUT/prof_${mode}_$dt.log 2>&1
    F=$(find $R/$OUT/prof_${mode}_$dt -name "*kernel_stats.csv" | head -1); cp "$F" $R/$OUT/${mode}_${dt}_stage0_kernel_stats.csv
    echo "$mode $dt $(grep ' ms' $R/$OUT/prof_${mode}_$dt.log) $(grep goff2 $R/$OUT/${mode}_${dt}_stage0_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120)"
  done
done
unset DLKA_GOFF_NKC0
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
for dt in f32 bf16; do timeout 600 python bench.py --steps 20 --warmup 5 --dtype $dt --no-cpu-baseline --no-tblock > $OUT/bench_$dt.json 2> $OUT/bench_$dt.err; python -c "
import json; d=json.load(open('$OUT/bench_$dt.json')); print('$dt', d['value'], d['ms_per_step'])"; done
