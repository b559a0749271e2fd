#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
TAG=${1:-r3h}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== bf16 + gx tests"; timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "bf16 or gx_fixed or headline or batchnorm" > $OUT/pytest_sel.log 2>&1; echo "exit $?"; grep -E "passed|failed|^FAILED|^E  " $OUT/pytest_sel.log | cut -c1-300 | head
for dt in bf16 f32; do
echo "== bench $dt"; timeout 900 python bench.py --steps 20 --warmup 5 --dtype $dt --no-cpu-baseline --no-tblock > $OUT/bench_$dt.json 2> $OUT/bench_$dt.err; python - <<PY
import json
d=json.load(open("$OUT/bench_$dt.json")); r=d["roofline"]
print("$dt", d["value"], d["ms_per_step"], sorted(r["per_op_ms"].items(), key=lambda kv:-kv[1])[:5])
PY
done
cd /tmp
for s in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_bf16_s$s -o t -- python $R/scripts/prof_stage.py --stage $s --dtype bf16 > $R/$OUT/prof_bf16_s$s.log 2>&1
  F=$(find $R/$OUT/prof_bf16_s$s -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/bf16_stage${s}_block_kernel_stats.csv
  grep " ms" $R/$OUT/prof_bf16_s$s.log
done
head -12 $R/$OUT/bf16_stage0_block_kernel_stats.csv | cut -c1-130
echo "== pmc traffic (fp32 stage-0 ops)"
OPS=deform_bwd_input,deform_bwd_offset,deform_bwd_weight,deform_fwd,offset_conv_fwd bash $R/scripts/pmc_traffic.sh $TAG/pmc > $R/$OUT/pmc.log 2>&1
python - <<PY
import json
d=json.load(open("$R/$OUT/pmc/pmc_traffic.json"))
for k,v in d.items(): print(k, v["traffic_bytes_per_launch"]/1e6, "MB")
PY
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete; find $OUT -name "*.db" -size +2M -delete; du -sh $OUT
