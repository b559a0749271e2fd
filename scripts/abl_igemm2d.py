#!/usr/bin/env python
"""Per-kernel times of the 2-D block's implicit-GEMM launches under other builds of the library (e.g. the TIMING-ONLY ablations -DDLKA_ABL=bits of cl_igemm.hip: wrong
results), one process per build.  usage: python scripts/abl_igemm2d.py lib.so [lib.so ...]   ("-" = the tree's library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import ctypes
    import torch
    import bench
    from deformablelka_amd import _lib as L
    path = sys.argv[2]
    if path != "-":
        cd = ctypes.CDLL(os.path.join(ROOT, path))
        for name, (rs, args) in L.SIGNATURES.items():
            if hasattr(cd, name):
                fn = getattr(cd, name); fn.restype = rs; fn.argtypes = args
        L._lib = cd
    os.environ["DLKA_BENCH_2D_ROWS"] = "60"
    d = bench.lka2d_metric(5, torch.device("cuda", 0), torch.bfloat16)
    print(path, d["value"], d["ms_per_block_fwd_bwd"])
    for k in d["roofline"]["kernels"]:
        if "igemm" in k["kernel"]:
            print("   %-50s %-14s x%d  %8.1f us" % (k["kernel"], k["shape"], k["launches_per_step"], k["avg_us"]))
    sys.exit(0)
for lib in sys.argv[1:]:
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], check=False)
