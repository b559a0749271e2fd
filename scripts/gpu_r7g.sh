#!/bin/bash
# grad_input fork default-on: the stack / graph / net tests and the bench line (tblock and full net see it through the one-call backward)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r7g}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -k "stack or graph or net or tblock or tokens or canary" > $OUT/pytest_subset.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_subset.log
for v in 0 1000000000; do
DLKA_GX_FORK_MIN_ROWS=$v timeout 600 python bench.py --no-cpu-baseline --extras --no-lka2d > $OUT/bench_$v.json 2> $OUT/bench_$v.err; echo "bench($v) exit $?"
python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("min_rows=$v f32", d["value"], d["ms_per_step"], "tblock", d["tblock"]["value"], d["tblock"]["hipgraph"].get("value"), "fullnet", (d.get("fullnet") or {}).get("value"), (d.get("fullnet") or {}).get("hipgraph"), "inf", (d.get("inference") or {}).get("value"))
PY
done
