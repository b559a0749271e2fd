#!/usr/bin/env python
"""Diagnostic for tests/test_nets_gpu.py::test_assembled_net_train_mode_vs_oracle_assembled_net: where does the worst parameter gradient differ?  A LeakyReLU-kink
mismatch (one pre-activation within rounding of 0) confines the error of a conv weight gradient to ONE output channel; a wrong kernel spreads it.
usage: python scripts/debug_net_kink.py [DLKA_DWPAIR=0 ...]   (KEY=VAL arguments are put in the environment first)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for a in sys.argv[1:]:
    k, v = a.split("=", 1)
    os.environ[k] = v
import torch  # noqa: E402
from tests import netoracle  # noqa: E402

res = netoracle.run_pair(torch.device("cuda", 0), (32, 64, 64), B=2, training=True)
s = netoracle.summarize(res, top=6)
print("flipped", s["flipped"], "worst same-cells:", [(k, round(v, 5)) for k, v in s["same_grad_worst"]])
for k, _ in s["same_grad_worst"][:3]:
    a, b = res["hip_grads"][k].double(), res["same_grads"][k].double()
    d = (a - b).abs()
    scale = float(b.abs().max())
    if d.dim() >= 2:
        per = d.reshape(d.shape[0], -1).max(1).values / scale
        top = torch.topk(per, min(4, per.numel()))
        print(k, tuple(a.shape), "per-output-channel max err / scale: top", [(int(i), round(float(v), 5)) for v, i in zip(top.values, top.indices)], "median", float(per.median()))
    else:
        top = torch.topk(d / scale, min(4, d.numel()))
        print(k, tuple(a.shape), "top elements", [(int(i), round(float(v), 5)) for v, i in zip(top.values, top.indices)], "median", float((d / scale).median()))
