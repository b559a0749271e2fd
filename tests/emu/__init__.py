"""TEST INFRASTRUCTURE — builds and loads a HOST build of the kernel sources against the wavefront emulator
(tests/emu/include/hip/hip_runtime.h).  Lets the CPU-only container check kernel logic before GPU time is spent.
Never used by the product."""
import ctypes
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(_HERE))
SO = os.path.join(_HERE, "_build", "libdlka_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build(force=False):
    srcs = sorted(glob.glob(os.path.join(ROOT, "deformablelka_amd", "csrc", "*.hip")))
    deps = srcs + glob.glob(os.path.join(ROOT, "deformablelka_amd", "csrc", "*.h")) + [
        os.path.join(ROOT, "include", "dlka.h"), os.path.join(_HERE, "include", "hip", "hip_runtime.h")]
    def fresh():
        return os.path.exists(SO) and all(os.path.getmtime(d) <= os.path.getmtime(SO) for d in deps)

    if not force and fresh():
        return SO
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    import fcntl
    with open(os.path.join(_HERE, "_build", ".lock"), "w") as lk:   # pytest-xdist workers / spawned ranks: ONE of them builds, the others wait
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not force and fresh():
            return SO
        return _build_locked(srcs)


def _build_locked(srcs):
    cxx = CLANG if os.path.exists(CLANG) else "clang++"
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(_HERE, "_build", os.path.basename(s) + ".o")
        objs.append(o)
        procs.append(subprocess.Popen([cxx, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-c", s, "-o", o,
                                       "-I" + os.path.join(_HERE, "include"), "-I" + os.path.join(ROOT, "include"),
                                       "-I" + os.path.join(ROOT, "deformablelka_amd", "csrc")]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("emulator build failed")
    tmp = SO + ".tmp%d" % os.getpid()   # linked aside, then renamed: a process that has the old library mapped keeps its (unlinked) file
    subprocess.check_call([cxx, "-shared", "-o", tmp] + objs + ["-lpthread"])
    os.replace(tmp, SO)
    return SO


def load():
    return ctypes.CDLL(build())
