#!/bin/bash
# round 5: LDS-tiled weight preparation — GPU bitwise test + rocprofv3 tables of the wrapper block at stages 2 / 3 (the prep launches) + the tblock metric
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r9b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "prepar or dwpair or stack" 2>&1 | tail -3 | tee $OUT/tests.log
cd /tmp
for s in 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/tb_$s -o t -- python $R/scripts/prof_tblock.py --stage $s --iters 10 > $R/$OUT/tb_$s.log 2>&1
  F=$(find $R/$OUT/tb_$s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/$OUT/tblock_stage${s}_kernel_stats.csv
  grep -h "prep_batch\|prep_table" $R/$OUT/tblock_stage${s}_kernel_stats.csv | cut -c1-120
done
cd $R
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-lka2d --no-fullnet --no-companion --no-roofline > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
