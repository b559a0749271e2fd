#!/bin/bash
# round-5 profile set: one-stream rocprofv3 table of the bench step, per-stage block tables (fp32 0-3, bf16 0-1), wrapper-block tables per stage, 2-D block tables per shape,
# per-process tblock A/B against round 4's library
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8z; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
DLKA_STACK_WGRAD_OVERLAP=0 DLKA_GX_FORK_MIN_ROWS=1000000000 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/bench1s -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tblock --no-lka2d --no-fullnet --no-companion --no-roofline > $R/$OUT/bench_one_stream.json 2> $R/$OUT/bench_one_stream.err
F=$(find $R/$OUT/bench1s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/$OUT/bench_one_stream_kernel_stats.csv
for cfg in 0:f32 1:f32 2:f32 3:f32 0:bf16 1:bf16; do
  s=${cfg%%:*}; dt=${cfg##*:}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${s}_$dt -o t -- python $R/scripts/prof_stage.py --stage $s --dtype $dt > $R/$OUT/p_${s}_$dt.log 2>&1
  F=$(find $R/$OUT/p_${s}_$dt -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/$OUT/${dt}_stage${s}_block_kernel_stats.csv
  echo "stage $s $dt: $(grep ' ms' $R/$OUT/p_${s}_$dt.log | tail -1)"
done
for s in 0 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/tb_$s -o t -- python $R/scripts/prof_tblock.py --stage $s --iters 10 > $R/$OUT/tb_$s.log 2>&1
  F=$(find $R/$OUT/tb_$s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/$OUT/tblock_stage${s}_kernel_stats.csv
done
for shape in "384 14" "192 28" "96 56"; do set -- $shape
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/l2_$1 -o t -- python $R/scripts/prof_lka2d.py --C $1 --hw $2 --iters 10 > $R/$OUT/l2_$1.log 2>&1
  F=$(find $R/$OUT/l2_$1 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/$OUT/lka2d_C$1_kernel_stats.csv
done
cd $R
AB_METRIC=tblock timeout 900 python scripts/ab_lka2d.py $OUT/ab_tblock_per_process.json - alt_lib/libdlka_hip_prev.so 2>&1 | grep -v Warning | tail -3
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete; du -sh $OUT
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/bench_one_stream_kernel_stats.csv"))]
tot=sum(float(r["TotalDurationNs"]) for r in rows if "dlka::" in r["Name"])
print("one-stream kernel time total (all launches) ms:", tot/1e6)
PY
