#!/bin/bash
# round-2 first GPU pass: reference-native parity (oracle/_ref), golden vectors from the reference op, full -m gpu suite, bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3a; mkdir -p $OUT; export TMPDIR=/tmp
(rocm-smi --showproductname 2>&1 | head -12; nproc) > $OUT/env.log 2>&1
echo "== ref tests"; timeout 900 python -m pytest tests/test_ref_d3d_gpu.py -m gpu -q -s > $OUT/pytest_ref.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_ref.log; tail -25 $OUT/pytest_ref.log
echo "== golden from reference op"; timeout 300 python tests/golden/make_ref_golden.py gpurun_out/r3a/d3d_reference_vectors.pt > $OUT/make_ref_golden.log 2>&1; echo "exit $?"; tail -3 $OUT/make_ref_golden.log
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -q -s --deselect tests/test_ref_d3d_gpu.py > $OUT/pytest_gpu.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_gpu.log; grep -E "passed|failed|^FAILED|^E  |tokens C=" $OUT/pytest_gpu.log | head -60
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json | cut -c1-600
du -sh $OUT
