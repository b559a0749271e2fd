#!/bin/bash
# HBM traffic per KERNEL of one stage block as the timed step runs it (scripts/prof_stage.py = forward + backward of one block in a hipGraph):
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC slot limit, MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel-trace only.
# usage: pmc_block.sh TAG [stages="0"] [dtypes="f32"]      -> gpurun_out/TAG/pmc_traffic_block.json
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-pmcb}; STAGES=${2:-0}; DTS=${3:-f32}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for dt in $DTS; do for s in $STAGES; do for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/stage${s}_${dt}/$ctr -o t -- python $R/scripts/prof_stage.py --stage $s --dtype $dt --iters 3 > $O/stage${s}_${dt}.$ctr.log 2>&1
done; done; done
python $R/scripts/pmc_block_aggregate.py $O ${ROUND:-r04} > $O/pmc_traffic_block.json
python - <<PY
import json
d=json.load(open("$O/pmc_traffic_block.json"))
for k,v in d.items():
    if k=="_meta": continue
    for n,e in sorted(v.items(), key=lambda kv:-kv[1]["hbm_bytes_per_launch"])[:12]:
        print(k, n[:70], e["launches"], round(e["hbm_bytes_per_launch"]/1e6,1), "MB")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -size +4M -delete
