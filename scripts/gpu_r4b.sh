#!/bin/bash
# r4b: bf16 contract at the four stage shapes; A/B of the 16-row-wave deformable forward (DLKA_FWD16); bench
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r4b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== bf16 gpu tests"; timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "bf16 or deform3d_cl" -s > $OUT/pytest_bf16.log 2>&1; echo "exit $?"; grep -E "bf16 tokens|passed|failed|Error" $OUT/pytest_bf16.log | tail -20
cd /tmp
for v in 0 1; do for cfg in "0 f32" "0 bf16" "1 f32"; do
  set -- $cfg
  DLKA_FWD16=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${v}_$1_$2 -o t -- python $R/scripts/prof_stage.py --stage $1 --dtype $2 > $R/$OUT/p_${v}_$1_$2.log 2>&1
  F=$(find $R/$OUT/p_${v}_$1_$2 -name "*kernel_stats.csv" | head -1)
  echo "FWD16=$v stage $1 $2: $(grep ' ms' $R/$OUT/p_${v}_$1_$2.log)   $(grep deform_fwd $F | cut -d, -f1-4 | tr '\n' ' ')"
  [ -n "$F" ] && cp $F $R/$OUT/fwd16_${v}_stage$1_$2_kernel_stats.csv
done; done
cd $R
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline --no-tblock > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "exit $?"; head -12 $OUT/bench_f32.err
python -c "
import json; d=json.load(open('$OUT/bench_f32.json')); print(d['value'], d['ms_per_step'], d['other_dtype'])"
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; du -sh $OUT
