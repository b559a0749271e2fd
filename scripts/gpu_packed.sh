#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${1:-packed}; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for mode in default packed; do
  unset DLKA_GOFF_PACKED
  [ $mode = packed ] && export DLKA_GOFF_PACKED=1
  for st in 0 1; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${mode}_$st -o t -- python $R/scripts/prof_stage.py --stage $st > $R/$OUT/p_${mode}_$st.log 2>&1
    F=$(find $R/$OUT/p_${mode}_$st -name "*kernel_stats.csv" | head -1); cp $F $R/$OUT/${mode}_stage$st.csv
    echo "$mode stage $st: $(grep ' ms' $R/$OUT/p_${mode}_$st.log | sed 's/.*bwd//')"
    grep "goff2\|conv_wave\|wgrad_dense\|igemm_kernel<2" $F | awk -F'",' '{n=split($2,a,","); printf "    %-70s %8.1f us\n", substr($1,2,70), a[3]/1000}'
  done
  cd $R; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tblock --no-companion --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode bench', d['value'], d['ms_per_step'])"; cd /tmp
done
find $R/$OUT -name "*kernel_trace.csv" -delete; find $R/$OUT -name "*.db" -delete
