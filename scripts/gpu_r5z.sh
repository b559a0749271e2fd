#!/bin/bash
# round 4, end-of-round validation: gpu_final.sh (suite, smoke, bench lines, rocprof tables, PMC), the random-shape fuzz (with the opt-in LDS-brick
# depthwise kernels too) and the offline CPU-baseline protocol
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r5z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ROUND=r05 bash scripts/gpu_final.sh $TAG
echo "== fuzz"; timeout 600 python scripts/fuzz_tokens.py 18 41 > $OUT/fuzz.log 2>&1; echo "exit $?"; tail -2 $OUT/fuzz.log
echo "== fuzz (DLKA_DW_LDS=2)"; DLKA_DW_LDS=2 timeout 600 python scripts/fuzz_tokens.py 12 43 > $OUT/fuzz_dwlds.log 2>&1; echo "exit $?"; tail -2 $OUT/fuzz_dwlds.log
echo "== cpu baseline protocol"; timeout 900 python scripts/cpu_baseline_protocol.py $OUT/cpu_baseline_protocol.json 2> $OUT/cpu_protocol.err | tail -2
