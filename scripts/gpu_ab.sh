#!/bin/bash
# A/B on ONE box: the current library against alt_lib/libdlka_hip_prev.so (bench only).  Build the other library first, e.g. from a
# `git worktree` of the commit to compare with: make -C <worktree>/deformablelka_amd/csrc && cp <worktree>/deformablelka_amd/_lib/libdlka_hip.so alt_lib/libdlka_hip_prev.so
[ -f "${GRAFT_REPO_ROOT:-/root/repo}/alt_lib/libdlka_hip_prev.so" ] || echo "(no alt_lib/libdlka_hip_prev.so: the 'prev' rows repeat the current library)"
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
L=deformablelka_amd/_lib/libdlka_hip.so
cp $L /tmp/cur.so
for round in 1 2; do
for which in cur prev; do
  if [ $which = prev ] && [ -f alt_lib/libdlka_hip_prev.so ]; then cp alt_lib/libdlka_hip_prev.so $L; else cp /tmp/cur.so $L; fi
  for dt in ${DTYPES:-f32 bf16}; do
    timeout 600 python bench.py --steps 20 --warmup 5 --dtype $dt --no-cpu-baseline --no-tblock > $OUT/bench_${which}_${dt}_$round.json 2> $OUT/bench_${which}_${dt}_$round.err
    python - <<PY
import json
d=json.load(open("$OUT/bench_${which}_${dt}_$round.json"))
print("$round $which $dt", d["value"], d["ms_per_step"])
PY
  done
done
done
cp /tmp/cur.so $L
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
