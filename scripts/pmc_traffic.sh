#!/bin/bash
# HBM traffic of the kernel-level ops at stage 0 from the L2 fabric counters (MI355X_MICROARCH.md §HBM): FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC slot limit), kernel-trace only.  GPU box.  usage: pmc_traffic.sh TAG
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-pmc}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
OPS=${OPS:-deform_bwd_input,deform_bwd_offset,deform_fwd,offset_conv_fwd,offset_conv_bwd_data,offset_conv_bwd_weight,deform_bwd_weight,dw7_fwd,dw7_bwd_weight,pointwise_fwd}
for op in ${OPS//,/ }; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$op/$ctr -o t -- python $R/scripts/prof_op.py --C 32 --N 32 --iters 3 --ops $op > $O/$op.$ctr.log 2>&1
  done
done
python $R/scripts/pmc_aggregate.py $O > $O/pmc_traffic.json
cat $O/pmc_traffic.json
find $O -name "*kernel_trace.csv" -delete
