#!/bin/bash
# parity subset around the LDS-brick data gradient + the bench line (both dtypes ride in the default line)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r6m}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_ref_d3d_gpu.py tests/test_ws_canary_gpu.py -x -q -m gpu > $OUT/pytest_subset.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_subset.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "bench exit $?"
python - <<PY
import json
d=json.loads(open("$OUT/bench_f32.json").read().strip().splitlines()[-1])
print("f32", d["value"], d["ms_per_step"], "bf16", d["other_dtype"]["value"], d["other_dtype"]["ms_per_step"], "tblock", d["tblock"]["value"], "lka2d", d["lka2d"]["value"])
print("roof", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("traffic"))
PY
