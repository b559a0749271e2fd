#!/bin/bash
# round 4, call a: the new parity tests verbosely, then the whole GPU suite in the driver's order, then the default bench line
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5a; mkdir -p $OUT; export TMPDIR=/tmp
echo "== new tests (verbose)"
timeout 1500 python -m pytest tests/test_nets_gpu.py tests/test_ws_canary_gpu.py tests/test_parity_gpu.py -m gpu -q -s -p no:cacheprovider \
  -k "assembled or canary or buffers or lka3d_block_vs_oracle or lka2d_attention_real_shapes or config2_batch24 or stack_prepare or experimental" > $OUT/new_tests.log 2>&1
echo "exit $?"; grep -E "passed|failed|Error|assert" $OUT/new_tests.log | tail -15
echo "== pytest -m gpu (driver order)"
timeout 1800 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "exit $?"; tail -4 $OUT/pytest_gpu.log
echo "== bench (default)"
timeout 900 python bench.py > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "exit $?"
python - <<PY
import json
d=json.load(open("$OUT/bench_f32.json"))
print(d["value"], d["ms_per_step"], "bf16:", (d.get("other_dtype") or {}).get("value"), "tblock:", (d.get("tblock") or {}).get("value"), "roof:", d["roofline"]["kernel"], d["roofline"]["frac"])
PY
