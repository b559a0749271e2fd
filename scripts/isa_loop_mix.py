#!/usr/bin/env python
"""Instruction mix of the hot loop (largest backward-branch region) of every kernel in a .hip file, compiled for gfx950.
usage: python scripts/isa_loop_mix.py deformablelka_amd/csrc/cl_deform_fwd.hip [name-substring]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/_isa_mix.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-I" + ROOT + "/include",
                "-I" + ROOT + "/deformablelka_amd/csrc", "-S", "--cuda-device-only", "-o", out, src], capture_output=True)
lines = open(out).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN4dlka\w+:", l)]
for i, name in starts:
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("void dlka::", "").split("(")[0]
    if pat not in dem:
        continue
    end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
    body = lines[i:end]
    labels = {}
    for k, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = k
    best = None
    for k, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.search(r"s_branch (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            span = (labels[m.group(1)], k)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    if best is None:
        continue
    cnt = collections.Counter()
    ops = collections.Counter()
    for l in body[best[0]:best[1]]:
        l = l.strip()
        if not l or l.startswith((".", ";", "/")) or l.endswith(":"):
            continue
        op = l.split()[0]
        if op.startswith("v_mfma"): c = "mfma"
        elif op.startswith("v_"): c = "valu"
        elif op.startswith("s_waitcnt"): c = "waitcnt"
        elif op.startswith("s_barrier"): c = "barrier"
        elif op.startswith("s_"): c = "salu"
        elif op.startswith("ds_"): c = "lds"
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")): c = "vmem"
        else: c = op
        cnt[c] += 1
        if c in ("valu", "lds", "vmem"): ops[op] += 1
    print(f"{dem[:80]:80s} loop {best[1]-best[0]:5d} lines  " + " ".join(f"{k}={v}" for k, v in sorted(cnt.items())))
    if os.environ.get("OPS"):
        print("    " + " ".join(f"{k}:{v}" for k, v in ops.most_common(25)))
