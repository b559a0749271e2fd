#!/usr/bin/env python
"""One case of scripts/fuzz_tokens.py again, verbose (cell flips, per-parameter errors on own offsets / on the kernels' cells), with the environment given as KEY=VAL arguments.
usage: python scripts/debug_fuzz_case.py B C d0 d1 d2 std seed [KEY=VAL ...]"""
import os
import sys
sys.path.insert(0, ".")
B, C, d0, d1, d2 = (int(v) for v in sys.argv[1:6])
std, seed = float(sys.argv[6]), int(sys.argv[7])
for a in sys.argv[8:]:
    k, v = a.split("=", 1)
    os.environ[k] = v
from tests import parity  # noqa: E402
try:
    parity.check_lka3d_tokens("cuda:0", B, C, (d0, d1, d2), seed=seed, offset_std=std, report_offsets=True)
    print("PASS")
except AssertionError as e:
    print("FAIL", str(e)[:300])
