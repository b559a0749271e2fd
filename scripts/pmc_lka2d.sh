#!/bin/bash
# HBM traffic per KERNEL of the 2-D block at the three decoder shapes of config 2 (bf16, B = 24): FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (TCC slot limit, MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel-trace only; aggregated like scripts/pmc_block.sh (read side x2 on gfx950).
# usage: pmc_lka2d.sh TAG      -> gpurun_out/TAG/pmc_traffic_lka2d.json   (copy to profiles/: bench.py's lka2d.roofline.traffic reads it)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-pmc2d}; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for shape in "384 14" "192 28" "96 56"; do set -- $shape; C=$1; HW=$2
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/stage2d_C${C}_bf16/$ctr -o t -- python $R/scripts/prof_lka2d.py --C $C --hw $HW --iters 3 > $O/lka2d_C${C}.$ctr.log 2>&1
  done
done
python $R/scripts/pmc_block_aggregate.py $O ${ROUND:-r08} | sed 's#scripts/prof_stage.py: one block of the stage, forward + backward, hipGraph replays#scripts/prof_lka2d.py: one 2-D block per decoder shape (bf16, B = 24), forward + backward, eager#; s#scripts/pmc_block.sh#scripts/pmc_lka2d.sh#' > $O/pmc_traffic_lka2d.json
python - <<PY
import json
d=json.load(open("$O/pmc_traffic_lka2d.json"))
for k,v in d.items():
    if k=="_meta": continue
    for n,e in sorted(v.items(), key=lambda kv:-kv[1]["hbm_bytes_per_launch"])[:8]:
        print(k, n[:70], e["launches"], round(e["hbm_bytes_per_launch"]/1e6,1), "MB")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -size +4M -delete
