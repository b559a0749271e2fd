#!/bin/bash
# round 3 (r04 tag), first GPU call: reference-op fixtures (3-D + the new 2-D embedding), the GPU suite, the bench with the launch-trace roofline,
# PMC traffic per kernel of the stage-0 block
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r4a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== goldens"; timeout 600 python tests/golden/make_ref_golden.py $OUT/d3d_reference_vectors.pt > $OUT/golden.log 2>&1; echo "exit $?"; tail -3 $OUT/golden.log
echo "== new gpu tests"; timeout 900 python -m pytest tests/test_ref_d3d_2d_gpu.py -m gpu -q > $OUT/pytest_2d.log 2>&1; echo "exit $?"; tail -5 $OUT/pytest_2d.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "exit $?"; tail -5 $OUT/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "exit $?"; tail -40 $OUT/bench_f32.err
echo "== pmc"; bash scripts/pmc_block.sh $TAG/pmc "0" "f32" 2>&1 | tail -30
du -sh $OUT
