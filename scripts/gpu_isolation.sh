#!/bin/bash
# Rules a self-inflicted corruption in or out behind a box that fails GPU tests (profiles/r04t_pytest_gpu_faulty_box.log):
#  1. the reference-vs-oracle file ALONE in a fresh process (none of this repo's kernels have run in it),
#  2. the canary test (guard bytes around every saved / workspace buffer, tests/test_ws_canary_gpu.py),
#  3. the whole suite twice with the test files in two different orders (a stray write by an earlier kernel would move with the order),
#  4. the two-stream stack tests in a loop with the weight-gradient overlap on and off (a stream race would be intermittent).
# Usage: bash scripts/gpu_isolation.sh [tag]      -> gpurun_out/<tag>/isolation.log
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-isolation}; OUT=gpurun_out/$TAG; mkdir -p $OUT; LOG=$OUT/isolation.log; : > $LOG
run() { echo "== $1" | tee -a $LOG; shift; "$@" >> $LOG 2>&1; echo "exit $?" | tee -a $LOG; tail -2 $LOG | head -1; }
run "1. tests/test_ref_d3d_gpu.py alone, fresh process" timeout 900 python -m pytest tests/test_ref_d3d_gpu.py -q -m gpu -p no:cacheprovider
run "2. canary: guard bytes around every scratch buffer" timeout 900 python -m pytest tests/test_ws_canary_gpu.py -q -m gpu -p no:cacheprovider
FILES=$(ls tests/test_*gpu*.py tests/test_parity_gpu.py 2>/dev/null | sort -u)
run "3a. suite, files in reverse alphabetical order" timeout 1500 python -m pytest $(echo $FILES | tr ' ' '\n' | sort -r | tr '\n' ' ') -q -m gpu -p no:cacheprovider
run "3b. suite, parity file last, nets first" timeout 1500 python -m pytest tests/test_nets_gpu.py tests/test_ws_canary_gpu.py tests/test_ref_d3d_2d_gpu.py tests/test_ref_d3d_gpu.py tests/test_parity_gpu.py -q -m gpu -p no:cacheprovider
for ov in 1 0; do
  run "4. stack graph-replay / prepare tests x ${LOOPS:-25}, DLKA_STACK_WGRAD_OVERLAP=$ov" env DLKA_STACK_WGRAD_OVERLAP=$ov timeout 1500 python - <<PY
import subprocess, sys
n = int("${LOOPS:-25}")
bad = 0
for i in range(n):
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_parity_gpu.py", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                        "hipgraph_replay or stack_prepare"], capture_output=True, text=True)
    if r.returncode != 0:
        bad += 1
        print(r.stdout[-1500:])
print(f"{n - bad} of {n} loops green")
sys.exit(1 if bad else 0)
PY
done
grep -E "^(==|exit|[0-9]+ (passed|failed)|.* loops green)" $LOG
