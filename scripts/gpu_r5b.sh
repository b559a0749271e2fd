#!/bin/bash
# round 4, call b: full GPU suite (driver order), 2-D block A/B (new lane=channel grad_input vs the window kernel), rocprof table of the 2-D step
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5b; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest -m gpu (driver order)"
timeout 1800 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "exit $?"; tail -4 $OUT/pytest_gpu.log
cat > /tmp/lka2d_ab.py <<'PY'
import json, sys, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
for dt in (torch.bfloat16, torch.float32):
    r = bench.lka2d_metric(5, dev, dt)
    k = [(x["kernel"], x["shape"], x["avg_us"]) for x in r.get("roofline", {}).get("kernels", [])][:8]
    print(json.dumps({"dtype": r["dtype"], "images_s": r["value"], "ms": r["ms_per_step"], "per_block": r["ms_per_block_fwd_bwd"], "top": k}))
PY
echo "== lka2d new kernel"; python /tmp/lka2d_ab.py 2>&1 | tail -2
echo "== lka2d window kernel (DLKA_DDW2D_GX_WINDOW=1)"; DLKA_DDW2D_GX_WINDOW=1 python /tmp/lka2d_ab.py 2>&1 | tail -2
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_lka2d -o t -- python /tmp/lka2d_ab.py > $R/$OUT/prof_lka2d.log 2>&1
F=$(find $R/$OUT/prof_lka2d -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $R/$OUT/lka2d_kernel_stats.csv && head -12 $R/$OUT/lka2d_kernel_stats.csv | cut -c1-160
cd $R; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
