#!/bin/bash
# round 4, call l: the LDS-brick depthwise kernel — A/B of the stage-0 / stage-1 block tables (DLKA_DW_LDS=0: register-row kernels; 1: LDS brick, single rows;
# + DLKA_DW_LDS_TH=2: row pairs), fp32 and bf16; the GPU parity tests that touch the block; the default bench line
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${1:-r5l}; mkdir -p $OUT; export TMPDIR=/tmp
echo "== parity (token path, stack, nets)"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_nets_gpu.py -x -q -m gpu -k "tokens or stack or assembled or lka3d" > $OUT/pytest_part.log 2>&1; echo "exit $?"; tail -3 $OUT/pytest_part.log
export DLKA_STACK_WGRAD_OVERLAP=0
cd /tmp
for v in 0 1; do for s in 0 1; do for dt in f32 bf16; do
  if [ $v = 2 ] && [ $s = 1 ]; then continue; fi
  if [ $v = 0 ]; then export DLKA_DW_LDS=0; unset DLKA_DW_LDS_TH; elif [ $v = 1 ]; then export DLKA_DW_LDS=1; unset DLKA_DW_LDS_TH; else export DLKA_DW_LDS=1; export DLKA_DW_LDS_TH=2; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${v}_s${s}_$dt -o t -- python $R/scripts/prof_stage.py --stage $s --dtype $dt > $R/$OUT/p_${v}_s${s}_$dt.log 2>&1
  F=$(find $R/$OUT/p_${v}_s${s}_$dt -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/${dt}_dwlds${v}_stage${s}_block_kernel_stats.csv
  echo "variant $v stage $s $dt: $(grep ' ms' $R/$OUT/p_${v}_s${s}_$dt.log | tail -1)"
done; done; done
unset DLKA_DW_LDS DLKA_DW_LDS_TH DLKA_STACK_WGRAD_OVERLAP
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
python - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/*_block_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        n=r['Name']
        if 'dwconv' in n and 'wgrad' not in n: print(f.split('/')[-1][:24], n.split('(')[0].replace('void dlka::','')[:70], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
echo "== bench (default)"
timeout 900 python bench.py > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "exit $?"
python - <<PY
import json
d=json.load(open("$OUT/bench_f32.json"))
print(d["value"], d["ms_per_step"], "bf16:", (d.get("other_dtype") or {}).get("value"), "tblock:", (d.get("tblock") or {}).get("value"), "lka2d:", (d.get("lka2d") or {}).get("value"), "roof:", d["roofline"]["kernel"], d["roofline"]["frac"])
PY
