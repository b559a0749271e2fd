#!/bin/bash
# round 4, call c: 2-D block with the lane = channel-pair grad_input kernel (A/B against the window kernel), rocprof table
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5d; mkdir -p $OUT; export TMPDIR=/tmp
cat > /tmp/lka2d_ab.py <<'PY'
import json, sys, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
for dt in (torch.bfloat16, torch.float32):
    r = bench.lka2d_metric(5, dev, dt)
    k = [(x["kernel"], x["shape"], x["avg_us"]) for x in r.get("roofline", {}).get("kernels", [])][:6]
    print(json.dumps({"dtype": r["dtype"], "images_s": r["value"], "ms": r["ms_per_step"], "per_block": r["ms_per_block_fwd_bwd"], "top": k}))
PY
echo "== parity of the 2-D block"; timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_ref_d3d_2d_gpu.py -q -m gpu -k "lka2d or 2d or ddw" 2>&1 | tail -2
echo "== lka2d default (per-shape selection)"; python /tmp/lka2d_ab.py 2>&1 | tail -2
echo "== lka2d tiles forced"; DLKA_DDW2D_GX=tiles python /tmp/lka2d_ab.py 2>&1 | tail -2
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_lka2d -o t -- python /tmp/lka2d_ab.py > $R/$OUT/prof_lka2d.log 2>&1
F=$(find $R/$OUT/prof_lka2d -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/lka2d_kernel_stats.csv && grep -E "gx3|gx_far|gx_kernel" $R/$OUT/lka2d_kernel_stats.csv | cut -c1-200
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
