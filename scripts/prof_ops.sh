#!/bin/bash
# rocprofv3 kernel-trace stats of selected ops at given stages (GPU box).  usage: prof_ops.sh TAG "C N" ["C N" ...]
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
OPS=${OPS:-deform_bwd_input,deform_bwd_offset,deform_bwd_weight,offset_conv_bwd_weight,deform_fwd,offset_conv_fwd}
cd /tmp
for st in "$@"; do
  set -- $st
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c$1 -o t -- python $R/scripts/prof_op.py --C $1 --N $2 --ops $OPS > $O/prof_c$1.log 2>&1
  F=$(find $O/prof_c$1 -name "*kernel_stats.csv" | head -1)
  echo "== C=$1 N=$2 $F"
  [ -n "$F" ] && cut -c1-160 "$F" | head -${HEAD:-16}
  grep " ms" $O/prof_c$1.log
done
find $O -name "*kernel_trace.csv" -size +5M -delete
find $O -name "*.db" -size +5M -delete
