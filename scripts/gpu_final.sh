#!/bin/bash
# end-of-round validation: full GPU suite, smoke, default bench (+ rocprof stats of the same command), bf16 bench, extras, per-stage block profiles
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ "${PROFILES_ONLY:-0}" != 1 ]; then
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "exit $?"; tail -3 $OUT/pytest_gpu.log
echo "== smoke"; timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "exit $?"; tail -2 $OUT/smoke.log
echo "== bench (default)"; timeout 900 python bench.py > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "exit $?"
echo "== bench --dtype bf16"; timeout 900 python bench.py --dtype bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; echo "exit $?"
echo "== bench under torch.distributed.run (1 rank, nccl), overlapped all-reduce schedule forced"
DLKA_BENCH_FORCE_SPLIT=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-tblock --no-companion --no-lka2d --no-roofline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; echo "exit $?"; python -c "
import json; d=json.load(open('$OUT/bench_dist1.json')); print('dist1', d['value'], d['ms_per_step'], d['config']['allreduce_overlap'], d['config']['allreduce_split_block'])"
if [ "${EXTRAS:-1}" = 1 ]; then echo "== bench --extras"; timeout 1200 python bench.py --extras --no-cpu-baseline --no-companion > $OUT/bench_extras.json 2> $OUT/bench_extras.err; echo "exit $?"; fi
python - <<PY
import json
for f in ("bench_f32","bench_bf16","bench_extras"):
    try:
        d=json.load(open("$OUT/%s.json"%f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d.get("roofline") or {}
    print(f, d["value"], d["ms_per_step"], "other:", (d.get("other_dtype") or {}).get("value"), "tblock:", (d.get("tblock") or {}).get("value"),
          "roof:", r.get("kernel"), r.get("frac"), "cpu:", (d.get("cpu_baseline") or {}).get("value"),
          "fullnet:", (d.get("fullnet") or {}).get("value"), "lka2d:", (d.get("lka2d") or {}).get("value"), "inf:", (d.get("inference") or {}).get("value"))
PY
fi
if [ "${SKIP_PROFILES:-0}" = 1 ]; then   # (block-stack kernels unchanged since the last profile set: only the full net's table)
  bash $R/scripts/gpu_netprof.sh $TAG/netprof | tail -3
  exit 0
fi
cd /tmp
echo "== rocprof of the bench command (as timed: the weight gradients of a block overlap the next block's data chain on a second stream — concurrent kernels stretch each other)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_bench -o t -- python $R/bench.py --no-cpu-baseline --no-tblock --no-companion --no-lka2d --no-fullnet > $R/$OUT/prof_bench.log 2>&1
F=$(find $R/$OUT/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/bench_kernel_stats.csv && head -6 $R/$OUT/bench_kernel_stats.csv | cut -c1-150
echo "== rocprof of the bench command on ONE stream (DLKA_STACK_WGRAD_OVERLAP=0): the per-kernel durations the roofline block's launch trace must agree with"
# (no library-internal forks either, and none of the companion metrics: the nn.Module / full-net loops run the same kernels beside their own side-stream work, which would stretch the averages)
DLKA_STACK_WGRAD_OVERLAP=0 DLKA_GX_FORK_MIN_ROWS=1000000000 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_bench1 -o t -- python $R/bench.py --no-cpu-baseline --no-tblock --no-companion --no-lka2d --no-fullnet --no-roofline > $R/$OUT/prof_bench1.log 2>&1
F=$(find $R/$OUT/prof_bench1 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/bench_one_stream_kernel_stats.csv && head -6 $R/$OUT/bench_one_stream_kernel_stats.csv | cut -c1-150
export DLKA_STACK_WGRAD_OVERLAP=0   # (one block per stage below: nothing to overlap with)
for dt in f32 bf16; do for s in 0 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_${dt}_s$s -o t -- python $R/scripts/prof_stage.py --stage $s --dtype $dt > $R/$OUT/prof_${dt}_s$s.log 2>&1
  F=$(find $R/$OUT/prof_${dt}_s$s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/${dt}_stage${s}_block_kernel_stats.csv
  echo "$dt $(grep ' ms' $R/$OUT/prof_${dt}_s$s.log)"
done; done
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; du -sh $OUT
echo "== PMC traffic per kernel of the stage-0 / stage-1 blocks (fp32), stage 0 bf16"
ROUND=${ROUND:-r10} bash scripts/pmc_block.sh $TAG/pmcb "0 1" "f32" > $OUT/pmc_block.log 2>&1; tail -14 $OUT/pmc_block.log
ROUND=${ROUND:-r10} bash scripts/pmc_block.sh $TAG/pmcb_bf16 "0" "bf16" > $OUT/pmc_block_bf16.log 2>&1; tail -5 $OUT/pmc_block_bf16.log
du -sh $OUT
