#!/bin/bash
# round 4, call e: PMC counters of the 2-D grad_input kernel (what is the tile kernel waiting for?)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5e; mkdir -p $OUT; export TMPDIR=/tmp
python scripts/prof_ddw2d.py
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/p$i -o t -- python $R/scripts/prof_ddw2d.py --reps 2 > $R/$OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
vals=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$R/$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "ddw2d" not in k: continue
        k=k.replace("void dlka::","").replace("dlka::","")[:44]
        vals[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
for k in vals:
    print(k)
    for c,v in sorted(vals[k].items()):
        print("   %-34s %14.4g per launch (%d launches)"%(c, v/max(1,cnt[k][c]), cnt[k][c]))
PY
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +2M -delete
