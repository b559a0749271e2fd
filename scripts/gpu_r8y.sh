#!/bin/bash
# per-stage block tables with the grad_input fork OFF (one stream: durations are the kernels' own), 3-D PMC refresh with the round's kernels
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8y; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for cfg in 0:f32 1:f32 2:f32 3:f32 0:bf16 1:bf16; do
  s=${cfg%%:*}; dt=${cfg##*:}
  DLKA_GX_FORK_MIN_ROWS=1000000000 DLKA_STACK_WGRAD_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${s}_$dt -o t -- python $R/scripts/prof_stage.py --stage $s --dtype $dt > $R/$OUT/p_${s}_$dt.log 2>&1
  F=$(find $R/$OUT/p_${s}_$dt -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/$OUT/${dt}_stage${s}_block_kernel_stats.csv
  echo "stage $s $dt: $(grep ' ms' $R/$OUT/p_${s}_$dt.log | tail -1)"
done
cd $R
for cfg in 0:f32 1:f32 2:f32 3:f32; do s=${cfg%%:*}; dt=${cfg##*:}; timeout 120 python scripts/prof_stage.py --stage $s --dtype $dt 2>&1 | grep " ms" | tail -1; done
ROUND="r08y (round 5, final tree)" bash scripts/pmc_block.sh r8y_pmc3d "0 1" "f32" 2>&1 | tail -3
ROUND="r08y (round 5, final tree)" bash scripts/pmc_block.sh r8y_pmc3d_bf16 "0" "bf16" 2>&1 | tail -2
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
