#!/bin/bash
# round 5 ablation: the split-K kernels of the small stages WITHOUT their global atomics (plain stores: wrong sums, timing only) — how much of their time is the atomics?
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r9e; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for s in 1 2 3; do
 for v in base noatomic; do
  L=""; [ $v = noatomic ] && L="--lib alt_lib/libdlka_noatomic.so"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${s}_$v -o t -- python $R/scripts/prof_stage.py --stage $s $L > $R/$OUT/p_${s}_$v.log 2>&1
  F=$(find $R/$OUT/p_${s}_$v -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/$OUT/stage${s}_${v}_kernel_stats.csv
  echo "stage $s $v: $(grep ' ms' $R/$OUT/p_${s}_$v.log | tail -1)"
  grep -h -E "cl_deform_fwd_kernel|cl_igemm_kernel|cl_conv_wave_kernel" $R/$OUT/stage${s}_${v}_kernel_stats.csv | cut -d, -f1,4 | cut -c1-120
 done
done
cd $R
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
