#!/bin/bash
# r4e: full GPU suite (new: 2-D bf16 at the real shapes, tblock at the headline stage, same-cells comparison), bench --extras (2-D bf16 + fp32, full net, inference)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r4f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "exit $?"; grep -E "passed|failed|Error|^FAILED|tblock C=|bf16 lka2d|tokens C=32 dims=\(32" $OUT/pytest_gpu.log | cut -c1-400 | tail -30
echo "== bench --extras"; timeout 1200 python bench.py --extras --no-tblock --no-companion > $OUT/bench_extras.json 2> $OUT/bench_extras.err; echo "exit $?"; tail -5 $OUT/bench_extras.err
python - <<PY
import json
d=json.load(open("$OUT/bench_extras.json"))
print("headline", d["value"], d["ms_per_step"])
for k in ("lka2d","lka2d_f32","fullnet","inference","inference_config5"):
    v=d.get(k) or {}
    print(k, v.get("value"), v.get("ms_per_step"), v.get("ms_per_block_fwd_bwd"), (v.get("roofline") or {}).get("kernel"), (v.get("roofline") or {}).get("frac"), (v.get("cpu_baseline") or {}).get("value"))
r=(d.get("lka2d") or {}).get("roofline") or {}
for k in r.get("kernels", []): print("   ", k)
PY
du -sh $OUT
