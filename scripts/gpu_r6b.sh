#!/bin/bash
# in-process A/B of library VARIANTS (scripts/build_variant.sh -> alt_lib/libdlka_NAME.so) on the stack step and on each stage alone
#   bash scripts/gpu_r6b.sh TAG variant...      env: DT=f32|bf16, STAGES="0 1 2 3" (which single-stage stacks), FULL=0 (skip the 21-block step), TRACE=1 (per-kernel table)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r6b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; shift
CFG=""
for v in base "$@"; do
  L=""; [ $v != base ] && L="_lib=alt_lib/libdlka_$v.so"
  [ "${FULL:-1}" = 1 ] && CFG="$CFG $v:$L"
  for s in ${STAGES-0 1 2 3}; do CFG="$CFG ${v}_s$s:${L:+$L,}_stages=$s"; done
done
T=""; [ "${TRACE:-0}" = 1 ] && T="--trace"
timeout 900 python scripts/ab_stack_knobs.py $OUT/ab_${DT:-f32}.json --dtype ${DT:-f32} --rounds 3 --steps 30 $T -- $CFG 2> $OUT/ab.err | tee $OUT/ab_${DT:-f32}.txt
tail -3 $OUT/ab.err
