#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${1:-gxc128}; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for mode in default fixed1; do
  unset DLKA_GX_FIXED
  [ $mode = fixed1 ] && export DLKA_GX_FIXED=1
  for st in 2 3; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${mode}_$st -o t -- python $R/scripts/prof_stage.py --stage $st > $R/$OUT/p_${mode}_$st.log 2>&1
    F=$(find $R/$OUT/p_${mode}_$st -name "*kernel_stats.csv" | head -1)
    echo "$mode stage $st: $(grep ' ms' $R/$OUT/p_${mode}_$st.log | sed 's/.*bwd//') $(grep 'gx_' $F | grep -v gather | cut -c1-60 | tr '\n' ' ') $(grep 'gx_' $F | grep -v gather | awk -F'",' '{print $2}' | cut -d, -f3)"
  done
done
find $R/$OUT -name "*kernel_trace.csv" -delete; find $R/$OUT -name "*.db" -delete
