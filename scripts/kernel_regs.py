#!/usr/bin/env python
"""VGPR / SGPR / spill / LDS figures of every kernel in libdlka_hip.so's object files (from the AMDGPU metadata notes).
usage: python scripts/kernel_regs.py [substring ...]"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RE = "/opt/rocm/lib/llvm/bin/llvm-readelf"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
pats = sys.argv[1:]
for o in sorted(glob.glob(os.path.join(ROOT, "deformablelka_amd/csrc/_build/*.o"))):
    tmp = "/tmp/_k.co"
    fat = "/tmp/_k.fat"
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", o, fat], capture_output=True)
    r = subprocess.run([BUNDLER, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={tmp}"], capture_output=True)
    if r.returncode != 0 or not os.path.exists(tmp) or os.path.getsize(tmp) == 0:
        continue
    txt = subprocess.run([RE, "--notes", tmp], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = g("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("void dlka::", "").replace("dlka::", "")
        dem = dem.split("(")[0]
        if pats and not any(p in dem for p in pats):
            continue
        print(f"{dem[:90]:90s} vgpr {g('vgpr_count'):>4} agpr {blk.split()[0]:>3} sgpr {g('sgpr_count'):>4} spill {g('vgpr_spill_count'):>3} lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size'):>5}")
    os.remove(tmp)
