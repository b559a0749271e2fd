#!/bin/bash
# round 4, call o: grouped weight-gradient folds on the side stream (DLKA_STACK_FINALIZE_GROUP) — test + A/B of the bench step
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${1:-r5o}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "stack or hipgraph" > $OUT/pytest_part.log 2>&1; echo "exit $?"; tail -3 $OUT/pytest_part.log
for g in 0 3 1 2 5 0 3; do
  DLKA_STACK_FINALIZE_GROUP=$g timeout 600 python bench.py --no-cpu-baseline --no-tblock --no-companion --no-lka2d --no-roofline > $OUT/bench_g$g.json 2> $OUT/bench_g$g.err
  python -c "
import json; d=json.load(open('$OUT/bench_g$g.json')); print('group $g:', d['value'], d['ms_per_step'], d.get('repetitions'))"
done
