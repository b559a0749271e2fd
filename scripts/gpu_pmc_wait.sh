#!/bin/bash
# per-kernel wait / issue counters of one stage-0 block (which kernels wait rather than issue?)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${1:-pmcw}; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM" "SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/$tag -o t -- python $R/scripts/prof_stage.py --stage ${STAGE:-0} --dtype ${DT:-f32} > $R/$OUT/$tag.log 2>&1
  echo "$tag: $(ls $R/$OUT/$tag 2>/dev/null | tr '\n' ' ')"
done
python - <<PY
import csv, glob, collections
vals=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(int)
for f in glob.glob("$R/$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "dlka::" not in k: continue
        k=k.replace("void dlka::","").replace("dlka::","")[:60]
        vals[k][r["Counter_Name"]]+=float(r["Counter_Value"])
names=sorted(vals, key=lambda k:-vals[k].get("SQ_WAVE_CYCLES",0))
print("%-60s %12s %8s %8s %8s %10s %10s"%("kernel","wave_cycles","wait_i%","valu%","any%","valu_insts","vmem_insts"))
for k in names[:22]:
    v=vals[k]; wc=v.get("SQ_WAVE_CYCLES",1) or 1
    print("%-60s %12.3g %8.1f %8.1f %8.1f %10.3g %10.3g"%(k, wc, 100*v.get("SQ_WAIT_INST_ANY",0)/wc, 100*v.get("SQ_ACTIVE_INST_VALU",0)/wc, 100*v.get("SQ_ACTIVE_INST_ANY",0)/wc, v.get("SQ_INSTS_VALU",0), v.get("SQ_INSTS_VMEM",0)))
PY
find $R/$OUT -name "*kernel_trace.csv" -delete; find $R/$OUT -name "*.db" -delete; find $R/$OUT -name "*counter_collection.csv" -size +2M -delete
