#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r8c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_ref_d3d_2d_gpu.py -x -q -k "lka2d or ddw2d or mixed_bf16_real or deform2d or dwconv2d or 2d" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for dt in bf16 f32; do timeout 120 python scripts/prof_ddw2d.py --dtype $dt 2>&1 | grep "us per" ; done | tee $OUT/prof_ddw2d.txt
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d['value'],d['ms_per_step'],d['other_dtype']['value'],d['tblock']['value'],d['lka2d']['value'])
for k in d['lka2d']['roofline']['kernels'][:12]: print(k)
PY
