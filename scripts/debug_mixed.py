"""debug: the wrapper block's mixed bf16 mode at (32, 32^3) — which build / switch / scratch content moves grad_x"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deformablelka_amd import _lib as L
which = sys.argv[1]
poison = len(sys.argv) > 2 and sys.argv[2] == "poison"
if which == "prev":
    L._lib = L.bind(ctypes.CDLL(os.path.join(ROOT, "alt_lib/libdlka_hip_prev.so")))
if poison:
    def scratch(nbytes, like):
        b = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=like.device)
        b.fill_(0xFF)   # NaN in fp32 and bf16
        return b
    L.scratch = scratch
from tests import parity
import oracle  # noqa
for C, dims in ((32, (32, 32, 32)), (64, (16, 16, 16))):
    try:
        e = parity.check_tblock3d_mixed_bf16("cuda:0", 2, C, dims, report=True)
        print(which, poison, C, "PASS", sorted(e.items(), key=lambda kv: -kv[1])[:3])
    except AssertionError as ex:
        print(which, poison, C, "FAIL", str(ex)[:200])
