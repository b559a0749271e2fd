#!/usr/bin/env python
"""A/B of environment knobs on the timed stack step, ONE process on ONE box (box-to-box variation on the pool exceeds most single changes):
every configuration builds its own DLKABlockStack, captures fwd+bwd in a hipGraph and replays it; the configurations are measured in
interleaved rounds so that clock / thermal drift hits all of them alike.

usage: python scripts/ab_stack_knobs.py OUT.json [--dtype f32|bf16] [--rounds 3] [--steps 30] -- NAME:K=V,K=V NAME2: ...
  e.g. ... -- base: inline0:DLKA_STACK_WGRAD_OVERLAP_MIN_C=64 packed:DLKA_GOFF_PACKED=1
(knobs read once per process by the LIBRARY — `static const` getenv — cannot be A/B-ed this way; the ones used here are read per stack / per call)
pseudo-knobs handled here: _stages=0+1 (only those stages' blocks), _lib=PATH (another build of libdlka_hip.so, bound in this process)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    argv = sys.argv[1:]
    cut = argv.index("--")
    opts, specs = argv[:cut], argv[cut + 1:]
    out_path = opts[0]
    dtype = torch.bfloat16 if "--dtype" in opts and opts[opts.index("--dtype") + 1] == "bf16" else torch.float32
    rounds = int(opts[opts.index("--rounds") + 1]) if "--rounds" in opts else 3
    steps = int(opts[opts.index("--steps") + 1]) if "--steps" in opts else 30
    from deformablelka_amd.stack import DLKABlockStack
    from deformablelka_amd import dp
    dev = torch.device("cuda", 0)
    configs = []
    for s in specs:
        name, _, kv = s.partition(":")
        env = dict(p.split("=", 1) for p in kv.split(",") if p)
        configs.append((name, env))
    knobs = sorted({k for _, e in configs for k in e if not k.startswith("_")})
    built = {}
    for name, env in configs:   # build + capture under the configuration's environment; replay needs none of it
        for k in knobs:
            os.environ.pop(k, None)
        os.environ.update({k: v for k, v in env.items() if not k.startswith("_")})
        kw = {}
        if "_stages" in env:
            from deformablelka_amd.stack import SYNAPSE_STAGES
            kw["stages"] = tuple(SYNAPSE_STAGES[int(i)] for i in env["_stages"].split("+"))
        if "_lib" in env:
            import ctypes
            from deformablelka_amd import _lib as L
            L._lib = L.bind(ctypes.CDLL(os.path.join(ROOT, env["_lib"])))
        else:
            from deformablelka_amd import _lib as L
            L._lib = None   # (the default library again)
        try:
            L.get_lib().dlka_env_refresh()   # (the library caches its fork switches; builds older than round 5 have no such export)
        except AttributeError:
            pass
        st = DLKABlockStack(2, device=dev, dtype=dtype, seed=1234, data_seed=4321, **kw)
        st.forward_backward()
        st.forward_backward()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            st.forward_backward()
        built[name] = (st, g)
    for k in knobs:
        os.environ.pop(k, None)
    res = {name: [] for name, _ in configs}
    for r in range(rounds):
        for name, _ in configs:
            st, g = built[name]

            def step():
                dp.step_single(st, 1e-12, 1, None, g.replay)
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            res[name].append(round((time.perf_counter() - t0) / steps * 1e3, 4))
    kernels = {}
    if "--trace" in opts:   # per-kernel launch trace (library events behind every launch, one stream, eager) of every configuration, side by side
        import bench
        for name, env in configs:
            if "_lib" in env:
                import ctypes
                L._lib = L.bind(ctypes.CDLL(os.path.join(ROOT, env["_lib"])))
            else:
                L._lib = None
            per, total, _, _ = bench.trace_step(built[name][0], reps=3)
            kernels[name] = {"%d %s" % (stg, bench.short_kernel(k)): [round(c, 2), round(ms * 1e3, 2)] for (stg, k), (c, ms) in per.items()}
            kernels[name]["(sum ms)"] = [1, round(total, 4)]
        names = [n for n, _ in configs]
        keys = sorted({k for n in names for k in kernels[n]}, key=lambda k: -max(kernels[n].get(k, [0, 0])[0] * kernels[n].get(k, [0, 0])[1] for n in names))
        print("%-78s %s" % ("kernel (stage name): us per launch", " ".join("%9s" % n[:9] for n in names)))
        for k in keys[:int(os.environ.get("AB_TRACE_ROWS", "40"))]:
            print("%-78s %s" % (k[:78], " ".join("%9.1f" % kernels[n].get(k, [0, float("nan")])[1] for n in names)))
    out = {"dtype": "bf16" if dtype == torch.bfloat16 else "f32", "steps": steps, "ms_per_step": res, "kernels": kernels,
           "median": {n: sorted(v)[len(v) // 2] for n, v in res.items()}, "configs": {n: e for n, e in configs},
           "finite": {n: built[n][0].health()["finite"] for n, _ in configs}}
    json.dump(out, open(out_path, "w"), indent=1)
    for n, v in res.items():
        print("%-24s median %.4f ms  %s" % (n, out["median"][n], v))


if __name__ == "__main__":
    main()
