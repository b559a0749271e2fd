#!/bin/bash
# round 5: per-kernel tables of one D-LKA block at the small stages (rocprofv3), to time single kernels after a change: prints the lines matching $1
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r9c; mkdir -p $OUT; export TMPDIR=/tmp
PAT="${1:-dwpair}"; STAGES="${2:-2 3}"; DT="${3:-f32}"
cd /tmp
for s in $STAGES; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${s}_$DT -o t -- python $R/scripts/prof_stage.py --stage $s --dtype $DT > $R/$OUT/p_${s}_$DT.log 2>&1
  F=$(find $R/$OUT/p_${s}_$DT -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/$OUT/${DT}_stage${s}_block_kernel_stats.csv
  echo "stage $s $DT: $(grep ' ms' $R/$OUT/p_${s}_$DT.log | tail -1)"
  grep -h -E "$PAT" $R/$OUT/${DT}_stage${s}_block_kernel_stats.csv | cut -c1-150
done
cd $R
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
