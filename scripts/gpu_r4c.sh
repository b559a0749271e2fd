#!/bin/bash
# r4c: after the VALU diet of the gather kernels (cheap corner offsets, published weights, separable derivatives): parity subset, per-stage block
# profiles (FWD16 A/B at stage 0), bench
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r4c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== gpu tests (subset)"; timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_ref_d3d_gpu.py -m gpu -q -x -k "bf16 or deform3d_cl or tokens or reference_native" > $OUT/pytest_sub.log 2>&1; echo "exit $?"; tail -3 $OUT/pytest_sub.log
cd /tmp
for cfg in "0 0 f32" "1 0 f32" "0 0 bf16" "1 0 bf16" "x 1 f32" "x 2 f32" "x 3 f32"; do
  set -- $cfg
  if [ "$1" = x ]; then unset DLKA_FWD16; else export DLKA_FWD16=$1; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_$1_$2_$3 -o t -- python $R/scripts/prof_stage.py --stage $2 --dtype $3 > $R/$OUT/p_$1_$2_$3.log 2>&1
  F=$(find $R/$OUT/p_$1_$2_$3 -name "*kernel_stats.csv" | head -1)
  echo "FWD16=$1 stage $2 $3: $(grep ' ms' $R/$OUT/p_$1_$2_$3.log)"
  [ -n "$F" ] && cp $F $R/$OUT/fwd16_$1_stage$2_$3_kernel_stats.csv && grep -E "deform_fwd|goff2|wgrad_samp|gx_fx2" $F | cut -d, -f1-4 | cut -c1-140
done
unset DLKA_FWD16
cd $R
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline --no-tblock > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "exit $?"; head -8 $OUT/bench_f32.err
python -c "
import json; d=json.load(open('$OUT/bench_f32.json')); print(d['value'], d['ms_per_step'], d['other_dtype'])"
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; du -sh $OUT
