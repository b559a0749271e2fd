#!/bin/bash
# issue / wait / busy counters of the deformable forward kernel, with and without the in-situ ablations (DLKA_FWD_ABL): what is a unit's time made of?
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${1:-pmcf}; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for abl in ${ABLS:-0 6}; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"; do
  tag=abl${abl}_$(echo $set | tr ' ' '_' | cut -c1-60)
  DLKA_FWD_ABL=$abl timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/$tag -o t -- python $R/scripts/prof_stage.py --stage 0 --dtype f32 --iters 3 > $R/$OUT/$tag.log 2>&1
  echo "$tag: $(ls $R/$OUT/$tag 2>/dev/null | tr '\n' ' ') $(tail -1 $R/$OUT/$tag.log | cut -c1-100)"
done; done
python - <<PY
import csv, glob, collections
vals=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$R/$OUT/*/*counter_collection.csv"):
    abl=f.split("/abl")[1].split("_")[0]
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "deform_fwd16" not in k and "goff2" not in k: continue
        key=(abl, k.replace("void dlka::","").split("(")[0][:40])
        vals[key][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[key][r["Counter_Name"]]+=1
for key in sorted(vals):
    print("==", key)
    for c in sorted(vals[key]):
        print("   %-28s %14.4g per launch" % (c, vals[key][c]/max(cnt[key][c],1)))
PY
find $R/$OUT -name "*kernel_trace.csv" -delete; find $R/$OUT -name "*.db" -delete; find $R/$OUT -name "*counter_collection.csv" -size +2M -delete
