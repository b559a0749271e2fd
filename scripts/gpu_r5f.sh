#!/bin/bash
# round 4, call f: bf16 parity on the GPU with the deformable contractions on the bf16 matrix cores, per-stage kernel tables with / without (DLKA_DEFORM_B16=0)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${1:-r5h}; mkdir -p $OUT; export TMPDIR=/tmp
echo "== bf16 parity"; timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "bf16" 2>&1 | tail -3
export DLKA_STACK_WGRAD_OVERLAP=0
cd /tmp
for b16 in 1; do for s in 0 1 2 3; do
  DLKA_DEFORM_B16=$b16 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${b16}_s$s -o t -- python $R/scripts/prof_stage.py --stage $s --dtype bf16 > $R/$OUT/p_${b16}_s$s.log 2>&1
  F=$(find $R/$OUT/p_${b16}_s$s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/bf16_b16_${b16}_stage${s}_block_kernel_stats.csv
  echo "b16=$b16 stage $s: $(grep ' ms' $R/$OUT/p_${b16}_s$s.log | tail -1)"
  grep -E "deform_fwd|goff|deform_gx" $R/$OUT/bf16_b16_${b16}_stage${s}_block_kernel_stats.csv | awk -F, '{printf "    %-90s %8.1f us\n", substr($1,1,90), $4/1000}'
done; done
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
