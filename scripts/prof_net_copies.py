"""Which torch ops of the full-net trainer iteration end in device-to-device copies / plain elementwise kernels: torch.profiler over one eager iteration,
aten::copy_ / aten::add_ / aten::add / aten::clone / aten::contiguous grouped by input shapes and by Python call site."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from deformablelka_amd import training

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = training.initialize_network(1, 14, (64, 128, 128), device=dev)
opt = training.initialize_optimizer(net, initial_lr=1e-6)
x = torch.randn(2, 1, 64, 128, 128, device=dev)
tgt = torch.randint(0, 14, (2, 64, 128, 128), device=dev)
net.train()
for _ in range(3):
    training.run_iteration(net, opt, x, tgt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    training.run_iteration(net, opt, x, tgt)
    torch.cuda.synchronize()
names = ("aten::copy_", "aten::clone", "aten::contiguous", "aten::add_", "aten::add", "aten::mul", "aten::fill_", "aten::zero_", "aten::cat", "aten::_to_copy")
print("== by input shape")
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in names]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:40]:
    print(f"{e.key:18s} n={e.count:4d} dev={e.device_time_total:9.1f} us  {str(e.input_shapes)[:150]}")
print("== by call site")
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key in names]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:30]:
    st = [s for s in e.stack if "site-packages/torch" not in s and "<built-in" not in s][:3]
    print(f"{e.key:18s} n={e.count:4d} dev={e.device_time_total:9.1f} us  {' <- '.join(s.split('/')[-1][:70] for s in st)}")
print("== memcpy / memset events")
acc = {}
for ev in prof.events():
    if "emcpy" in ev.name or "emset" in ev.name:
        a = acc.setdefault(ev.name, [0, 0.0]); a[0] += 1; a[1] += ev.device_time
for k, v in acc.items():
    print(k, v)
print("== large copies / adds and the op chain that issued them (cpu_parent names, innermost first)")
seen = {}
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::add", "aten::add_", "aten::contiguous") and ev.device_time_total > 25.0:
        chain, p = [], ev.cpu_parent
        while p is not None and len(chain) < 6:
            chain.append(p.name[:60])
            p = p.cpu_parent
        key = (ev.name, str(ev.input_shapes)[:60], " <- ".join(chain))
        a = seen.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += ev.device_time_total
for (n, shp, ch), (c, t) in sorted(seen.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n:16s} n={c:3d} dev={t:8.1f} us  {shp}  {ch}")
