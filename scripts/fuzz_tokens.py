"""Random-shape check of the token-layout block against the oracle (GPU): ragged volumes, both channel widths with the second-generation
grad_input kernel, both dtypes.  python scripts/fuzz_tokens.py [n] [seed] [small]"""
import random, sys
sys.path.insert(0, ".")
import torch
from tests import parity

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
small = len(sys.argv) > 3 and sys.argv[3] == "small"   # "small": volumes of at most 512 voxels with W in {4, 8} — the fused depthwise pair (cl_dwpair.hip) takes them
bad = 0
pair0 = None
if small:
    from deformablelka_amd import _lib
    pair0 = _lib.get_lib().dlka_dwpair_launch_count()
for k in range(n):
    C = rng.choice([32, 32, 64, 128] + ([256] if small else []))
    dims = tuple(rng.randint(3, 14) for _ in range(3))
    if small:
        w = rng.choice([4, 8])
        while True:
            a, b = rng.randint(1, 12), rng.randint(1, 12)
            if a * b * w <= 512:
                break
        dims = (a, b, w)
    B = rng.choice([1, 2])
    std = rng.choice([0.02, 0.2, 0.6])
    try:
        if k % 3 == 2 and C <= 64:
            parity.check_lka3d_tokens_bf16("cuda:0", B, C, dims, seed=k, offset_std=std)
            kind = "bf16"
        else:
            parity.check_lka3d_tokens("cuda:0", B, C, dims, seed=k, offset_std=std)
            kind = "f32"
        print(f"ok   {kind} B={B} C={C} dims={dims} offset_std={std}")
    except AssertionError as e:
        bad += 1
        print(f"FAIL B={B} C={C} dims={dims} offset_std={std}: {str(e)[:200]}")
if small:
    print("fused depthwise pair launches:", _lib.get_lib().dlka_dwpair_launch_count() - pair0)
print("failures:", bad)
sys.exit(1 if bad else 0)
