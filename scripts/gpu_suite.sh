#!/bin/bash
# the driver's round-end checks on one box: the whole -m gpu suite, smoke(), the default bench line
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-suite}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][0])
print("f32", d['value'], d['ms_per_step'], "bf16", d['other_dtype']['value'], "tblock", d['tblock']['value'], "lka2d", d['lka2d']['value'], "fullnet", (d.get('fullnet') or {}).get('value'))
PY
