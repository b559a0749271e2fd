#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${1:-lat}; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for dt in f32 bf16; do for st in ${STAGES:-0 1}; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${dt}_$st -o t -- python $R/scripts/prof_stage.py --stage $st --dtype $dt > $R/$OUT/p_${dt}_$st.log 2>&1
    F=$(find $R/$OUT/p_${dt}_$st -name "*kernel_stats.csv" | head -1); cp $F $R/$OUT/${dt}_stage$st.csv
    echo "$dt stage $st: $(grep ' ms' $R/$OUT/p_${dt}_$st.log | sed 's/.*bwd//')"
    grep "${KERNELS:-gx_fx2\|deform_fwd\|goff2}" $F | awk -F'",' '{n=split($2,a,","); printf "    %-75s %8.1f us\n", substr($1,2,75), a[3]/1000}'
done; done
find $R/$OUT -name "*kernel_trace.csv" -delete; find $R/$OUT -name "*.db" -delete
