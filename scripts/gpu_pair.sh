#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${1:-pair}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "tokens or tblock or hipgraph" 2>&1 | tail -3
for mode in fused unfused; do
  unset DLKA_PW_UNFUSED
  [ $mode = unfused ] && export DLKA_PW_UNFUSED=1
  for dt in f32 bf16; do
    python bench.py --steps 20 --warmup 5 --dtype $dt --no-cpu-baseline --no-tblock --no-companion --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode $dt', d['value'], d['ms_per_step'])"
  done
done
unset DLKA_PW_UNFUSED
cd /tmp
for st in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_$st -o t -- python $R/scripts/prof_stage.py --stage $st > $R/$OUT/p_$st.log 2>&1
  F=$(find $R/$OUT/p_$st -name "*kernel_stats.csv" | head -1)
  echo "stage $st: $(grep ' ms' $R/$OUT/p_$st.log | sed 's/.*bwd//')"; grep "pointwise" $F | awk -F'",' '{n=split($2,a,","); printf "    %-70s x%d %8.1f us\n", substr($1,2,70), a[1]/22, a[3]/1000}'
done
find $R/$OUT -name "*kernel_trace.csv" -delete; find $R/$OUT -name "*.db" -delete
