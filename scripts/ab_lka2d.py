#!/usr/bin/env python
"""A/B of builds of the library on the 2-D block metric (bench.lka2d_metric: config 2, bf16, B = 24) — or, with AB_METRIC=tblock, on the wrapper-block stack — on ONE box:
every measurement in its OWN process (two libraries bound in one process are not measured alike: the second one bound ran 6 - 9 % slower whichever it was — round 5's first
version of this script did that and overstated a gain), alternating A B A B.  usage: [AB_METRIC=tblock] python scripts/ab_lka2d.py OUT.json [libA.so libB.so "-@KEY=VAL" ...]   ("-" = the tree's library, "@K=V,..." = environment of that measurement)
(default: the tree's library against alt_lib/libdlka_hip_prev.so)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import ctypes
    import torch
    import bench
    from deformablelka_amd import _lib as L
    path = sys.argv[2]
    if "@" in path:   # "LIB@KEY=VAL,KEY=VAL": environment of this measurement
        path, kv = path.split("@", 1)
        os.environ.update(dict(x.split("=", 1) for x in kv.split(",") if x))
    if path != "-":
        cd = ctypes.CDLL(os.path.join(ROOT, path))
        for name, (rs, args) in L.SIGNATURES.items():   # (an older build may lack this round's new exports: bind what it has)
            if hasattr(cd, name):
                fn = getattr(cd, name); fn.restype = rs; fn.argtypes = args
        L._lib = cd
    if os.environ.get("AB_METRIC") == "stack":   # the timed step of bench.py's headline: the 21-block engine, fwd + bwd from a hipGraph + SGD update (AB_DTYPE=bf16, AB_STAGES=2+3)
        import time
        from deformablelka_amd import dp
        from deformablelka_amd.stack import DLKABlockStack, SYNAPSE_STAGES
        kw = {}
        if os.environ.get("AB_STAGES"):
            kw["stages"] = tuple(SYNAPSE_STAGES[int(i)] for i in os.environ["AB_STAGES"].split("+"))
        st = DLKABlockStack(2, device=torch.device("cuda", 0), dtype=torch.bfloat16 if os.environ.get("AB_DTYPE") == "bf16" else torch.float32, seed=1234, data_seed=4321, **kw)
        st.forward_backward(); st.forward_backward()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            st.forward_backward()
        vals = []
        for _ in range(3):
            for _ in range(5):
                dp.step_single(st, 1e-12, 1, None, g.replay)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                dp.step_single(st, 1e-12, 1, None, g.replay)
            torch.cuda.synchronize()
            vals.append(round((time.perf_counter() - t0) / 30 * 1e3, 4))
        print("RESULT", json.dumps({"ms_per_step": sorted(vals)[1], "all": vals, "finite": st.health()["finite"]}))
    elif os.environ.get("AB_METRIC") == "fullnet":   # the whole D_LKA_Former trainer iteration (bench.fullnet_metric, eager)
        r = bench.fullnet_metric(2, 8, torch.device("cuda", 0))
        print("RESULT", json.dumps({"value": r["value"], "ms_per_step": r["ms_per_step"], "loss": r.get("loss")}))
    elif os.environ.get("AB_METRIC") == "tblock":
        if os.environ.get("AB_OLD_MASK") == "1":   # (round 6 A/B: the Dropout3d multipliers as dropout3d(ones) — four launches per block instead of two)
            import deformablelka_amd as dk
            dk.TransformerBlock_3D_single_deform_LKA._draw_drop_mask = lambda self, B, C, dtype, device: torch.nn.functional.dropout3d(
                torch.ones(B, C, 1, 1, 1, dtype=dtype, device=device), self.conv8[0].p, True).view(B, C)
        r = bench.tblock_metric(2, 10, 3, torch.device("cuda", 0))
        print("RESULT", json.dumps({"value": r["value"], "graph": r.get("hipgraph", {}).get("value")}))
    else:
        r = bench.lka2d_metric(10, torch.device("cuda", 0), torch.bfloat16)
        print("RESULT", json.dumps({"value": r["value"], "per_block_ms": r["ms_per_block_fwd_bwd"]}))
    sys.exit(0)
out_path = sys.argv[1]
libs = sys.argv[2:] or ["-", "alt_lib/libdlka_hip_prev.so"]
res = {l: [] for l in libs}
for rnd in range(2):
    for l in libs:
        if l.split("@")[0] != "-" and not os.path.exists(os.path.join(ROOT, l.split("@")[0])):
            continue
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", l], capture_output=True, text=True, timeout=600)
        line = [x for x in p.stdout.splitlines() if x.startswith("RESULT")]
        res[l].append(json.loads(line[0][7:]) if line else {"error": p.stderr[-300:]})
json.dump(res, open(out_path, "w"), indent=1)
for l, v in res.items():
    print("tree" if l == "-" else l, v)
