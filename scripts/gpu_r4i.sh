#!/bin/bash
# r4i: what bounds cl_deform_gx_fx2_kernel — LDS micro-benchmark of its scatter pattern + LDS PMC counters of the stage-0 block
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${1:-r4i}; mkdir -p $OUT; export TMPDIR=/tmp
echo "== ubench"; timeout 300 scripts/ubench/lds_u64_patterns | tee $OUT/lds_u64_patterns.txt
cd /tmp
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/$tag -o t -- python $R/scripts/prof_stage.py --stage 0 --dtype f32 --iters 3 > $R/$OUT/$tag.log 2>&1
  echo "$tag: $(ls $R/$OUT/$tag 2>/dev/null | tr '\n' ' ')"
done
python - <<PY
import csv, glob, collections
vals=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$R/$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "dlka::" not in k: continue
        k=k.replace("void dlka::","").replace("dlka::","").split("(")[0][:56]
        vals[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
cols=["GRBM_GUI_ACTIVE","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_INSTS_LDS","SQ_ACTIVE_INST_LDS","SQ_LDS_IDX_ACTIVE","SQ_LDS_BANK_CONFLICT","SQ_LDS_ADDR_CONFLICT","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_VALU","SQ_VALU_MFMA_BUSY_CYCLES"]
names=sorted(vals, key=lambda k:-vals[k].get("GRBM_GUI_ACTIVE",0)/max(1,n[k].get("GRBM_GUI_ACTIVE",1)))
print("per launch:  %-56s "%"kernel"+" ".join("%12s"%c[-12:] for c in cols))
for k in names[:14]:
    print("             %-56s "%k+" ".join("%12.4g"%(vals[k].get(c,0)/max(1,n[k].get(c,1))) for c in cols))
PY
find $R/$OUT -name "*kernel_trace.csv" -delete; find $R/$OUT -name "*.db" -delete; find $R/$OUT -name "*counter_collection.csv" -size +2M -delete
