#!/bin/bash
# last check of the round's final tree: the whole GPU suite, smoke, the default bench line, bf16 line
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r7z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "exit $?"; tail -3 $OUT/pytest_gpu.log
echo "== smoke"; timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "exit $?"; tail -1 $OUT/smoke.log
echo "== bench (default)"; timeout 900 python bench.py > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "exit $?"
python - <<PY
import json
d=json.loads(open("$OUT/bench_f32.json").read().strip().splitlines()[-1])
r=d.get("roofline") or {}
print("f32", d["value"], d["ms_per_step"], "bf16", d["other_dtype"]["value"], d["other_dtype"]["ms_per_step"], "tblock", d["tblock"]["value"], "lka2d", d["lka2d"]["value"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
print("roof", r.get("kernel"), r.get("frac"), r.get("traffic"), r.get("traffic_source","")[:60])
PY
