#!/bin/bash
# round 4, call n: A/B of the whole library built with / without the SLP vectoriser's packed fp32 (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32):
# stage tables and the bench line for both builds (deformablelka_amd/_lib/libdlka_hip.so vs libdlka_hip_noslp.so, built by hand with
# -fno-slp-vectorize -DDLKA_NO_PK), each with the register-row (DLKA_DW_LDS=0) and the LDS-brick depthwise kernels
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${1:-r5n}; mkdir -p $OUT; export TMPDIR=/tmp
L=deformablelka_amd/_lib
cp $L/libdlka_hip.so /tmp/lib_slp.so
for lib in slp noslp; do
  if [ $lib = noslp ]; then cp $L/libdlka_hip_noslp.so $L/libdlka_hip.so; else cp /tmp/lib_slp.so $L/libdlka_hip.so; fi
  export DLKA_STACK_WGRAD_OVERLAP=0
  cd /tmp
  for v in 0 1; do for s in 0 1 2; do for dt in f32 bf16; do
    if [ $dt = bf16 ] && [ $s != 0 ]; then continue; fi
    if [ $v = 1 ] && [ $s = 2 ]; then continue; fi
    export DLKA_DW_LDS=$v
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/p_${lib}_${v}_s${s}_$dt -o t -- python $R/scripts/prof_stage.py --stage $s --dtype $dt > $R/$OUT/p_${lib}_${v}_s${s}_$dt.log 2>&1
    F=$(find $R/$OUT/p_${lib}_${v}_s${s}_$dt -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/${lib}_${dt}_dwlds${v}_stage${s}_block_kernel_stats.csv
    echo "$lib dwlds $v stage $s $dt: $(grep ' ms' $R/$OUT/p_${lib}_${v}_s${s}_$dt.log | tail -1)"
  done; done; done
  unset DLKA_DW_LDS DLKA_STACK_WGRAD_OVERLAP
  cd $R
  echo "== bench ($lib, DLKA_DW_LDS=0)"
  DLKA_DW_LDS=0 timeout 900 python bench.py > $OUT/bench_${lib}.json 2> $OUT/bench_${lib}.err; echo "exit $?"
  python - <<PY
import json
d=json.load(open("$OUT/bench_${lib}.json"))
print("$lib", d["value"], d["ms_per_step"], "bf16:", (d.get("other_dtype") or {}).get("value"), "tblock:", (d.get("tblock") or {}).get("value"), "lka2d:", (d.get("lka2d") or {}).get("value"), "roof:", d["roofline"]["kernel"], d["roofline"]["frac"])
PY
done
cp /tmp/lib_slp.so $L/libdlka_hip.so
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
python - <<PY
import csv, glob, collections
rows=collections.defaultdict(dict)
for f in sorted(glob.glob("$OUT/*_block_kernel_stats.csv")):
    tag=f.split('/')[-1].replace('_block_kernel_stats.csv','')
    lib=tag.split('_')[0]; rest=tag[len(lib)+1:]
    for r in csv.DictReader(open(f)):
        n=r['Name'].split('(')[0].replace('void dlka::','')[:72]
        rows[(rest,n)][lib]=float(r['AverageNs'])/1e3
for (rest,n),d in sorted(rows.items()):
    if 'slp' in d and 'noslp' in d and max(d.values())>8 and abs(d['slp']-d['noslp'])>0.03*d['slp']:
        print(f"{rest:24s} {n:72s} slp {d['slp']:7.1f}  noslp {d['noslp']:7.1f}")
PY
