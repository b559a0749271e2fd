#!/bin/bash
# round 5, last change: the wrapper block's forward prepares its own and the attention's weights in ONE launch — the tests that run through it + the wrapper-block / full-net lines
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r9p; mkdir -p $OUT
timeout 600 python -m pytest tests/test_nets_gpu.py tests/test_parity_gpu.py -q -m gpu -k "assembled_net or tblock or prepar or hipgraph or DLKAFormer or former" 2>&1 | tail -3 | tee $OUT/tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lka2d --no-companion --no-roofline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][0])
print("f32", d['value'], "tblock", d['tblock']['value'], d['tblock'].get('hipgraph',{}).get('value'), "fullnet", (d.get('fullnet') or {}).get('value'), ((d.get('fullnet') or {}).get('hipgraph') or {}).get('value'))
PY
