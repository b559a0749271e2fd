#!/bin/bash
# SQ counters per kernel of one 2-D block (or any command): two rocprofv3 --pmc passes of 8 SQ counters each (MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel-trace only.
# usage: pmc_sq.sh TAG [command ...]   -> gpurun_out/TAG/sq_pass{1,2}_counter_collection.csv + a per-kernel summary on stdout
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-pmcsq}; shift
CMD=${@:-python $R/scripts/prof_lka2d.py --C 96 --hw 56 --iters 2}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"
i=1
for P in "$P1" "$P2"; do
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pass$i -o sq -- $CMD > $O/pass$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob("$O/pass*/**/*counter_collection.csv", recursive=True) + glob.glob("$O/pass*/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "").replace("dlka::", ""))[:70]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen and r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_LDS"):
            seen.add(key)
for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:14]:
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    print(name)
    print("   wave_cycles %.3g  wait_any %.2f  wait_inst_any %.2f (lds %.2f)  active_inst %.2f | mfma_busy_cycles %.3g  busy_cycles %.3g  mfma/busy %.3f | valu insts %.3g" % (
        wc, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_WAIT_INST_LDS", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), c.get("SQ_BUSY_CYCLES", 0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("SQ_BUSY_CYCLES", 1), 1), c.get("SQ_INSTS_VALU", 0)))
    print("   lds insts %.3g  bank_conflict/idx_active %.3f  vmem_rd insts %.3g  salu %.3g | active valu %.3g lds %.3g vmem %.3g" % (
        c.get("SQ_INSTS_LDS", 0), c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1), c.get("SQ_INSTS_VMEM_RD", 0), c.get("SQ_INSTS_SALU", 0),
        c.get("SQ_ACTIVE_INST_VALU", 0), c.get("SQ_ACTIVE_INST_LDS", 0), c.get("SQ_ACTIVE_INST_VMEM", 0)))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
