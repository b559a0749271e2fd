#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${1:-gxabl}; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for mode in ${MODES:-fx2 fx2_nofar fx1} ${EXTRA:-}; do
  unset DLKA_GX_ABL DLKA_GX_FIXED DLKA_GX_TAPFAR
  case $mode in fx2_nofar) export DLKA_GX_ABL=3;; fx2_abl1) export DLKA_GX_ABL=1;; fx2_abl2) export DLKA_GX_ABL=2;; fx1) export DLKA_GX_FIXED=2;; fx1_abl1) export DLKA_GX_FIXED=2 DLKA_GX_ABL=1;; fx1_abl2) export DLKA_GX_FIXED=2 DLKA_GX_ABL=2;; esac
  for st in ${STAGES:-0 1}; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${mode}_$st -o t -- python $R/scripts/prof_stage.py --stage $st > $R/$OUT/p_${mode}_$st.log 2>&1
    F=$(find $R/$OUT/p_${mode}_$st -name "*kernel_stats.csv" | head -1)
    echo "$mode stage $st: $(grep 'gx_' $F | grep -v gather | awk -F'",' '{print $2}' | cut -d, -f3)"
  done
done
find $R/$OUT -name "*kernel_trace.csv" -delete; find $R/$OUT -name "*.db" -delete
