"""Random-shape check of the 2-D block against the oracle (GPU): ragged images, the widths of the channels-last fast path, both dtypes, both grad_input generations
(DLKA_DDW2D_GX=tiles | window | per shape).  python scripts/fuzz_lka2d.py [n] [seed]"""
import os, random, sys
sys.path.insert(0, ".")
import torch
from tests import parity

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for k in range(n):
    C = rng.choice([32, 64, 96, 96, 128, 192])
    H, W = rng.randint(5, 40), rng.randint(5, 40)
    B = rng.choice([1, 2, 3])
    std = rng.choice([0.02, 0.1, 0.3])
    gx = rng.choice([None, "tiles", "window"])
    if gx is None:
        os.environ.pop("DLKA_DDW2D_GX", None)
    else:
        os.environ["DLKA_DDW2D_GX"] = gx
    try:
        if k % 3 == 2:
            parity.check_lka2d_attention_bf16("cuda:0", B, C, H, W, seed=k, offset_std=std)
            kind = "bf16"
        else:
            parity.check_lka2d_attention("cuda:0", B, C, H, W, seed=k, offset_std=std)
            kind = "f32"
        print(f"ok   {kind} B={B} C={C} {H}x{W} offset_std={std} gx={gx}")
    except AssertionError as e:
        bad += 1
        print(f"FAIL B={B} C={C} {H}x{W} offset_std={std} gx={gx}: {str(e)[:200]}")
print("failures:", bad)
sys.exit(1 if bad else 0)
