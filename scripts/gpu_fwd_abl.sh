#!/bin/bash
# in-situ ablations of the 16-row deformable forward kernel (DLKA_FWD_ABL, profiling only)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${1:-abl}; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for v in ${ABLS:-0 5 6}; do
  DLKA_FWD_ABL=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_$v -o t -- python $R/scripts/prof_stage.py --stage 0 --dtype f32 > $R/$OUT/p_$v.log 2>&1
  F=$(find $R/$OUT/p_$v -name "*kernel_stats.csv" | head -1)
  echo "ABL=$v: $(grep deform_fwd16 $F | awk -F'",' '{print $1}' | cut -c1-60) $(grep deform_fwd16 $F | awk -F, '{print $(NF-4)}')"
  grep deform_fwd16 $F | cut -c1-200
done
find $R/$OUT -name "*kernel_trace.csv" -delete; find $R/$OUT -name "*.db" -delete
