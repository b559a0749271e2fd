"""Times the net's full-resolution 3^3 plumbing conv (2 x 16 x 64 x 128 x 128, 16 -> 16 channels, fp32) through ops.conv3d_forward / conv3d_backward with the
launch trace (per-kernel durations without rocprof) and checks forward + data gradient against torch's own conv at a slab of the volume."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import byref, c_float, create_string_buffer
from deformablelka_amd import ops, _lib as L

if os.environ.get("LIB"):   # another build of the library (e.g. a timing-only ablation: NOCHECK=1 skips the comparisons)
    import ctypes
    L._lib = L.bind(ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["LIB"])))
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(2, 16, 64, 128, 128, device=dev)
w = torch.randn(16, 16, 3, 3, 3, device=dev) * 0.05
b = torch.randn(16, device=dev)
go = torch.randn(2, 16, 64, 128, 128, device=dev)
lib = L.get_lib()
for _ in range(2):
    y = ops.conv3d_forward(x, w, b, 1, 1, 1, 1)
    gi, gw, gb = ops.conv3d_backward(x, w, go, 1, 1, 1, 1)
torch.cuda.synchronize()
L.check(lib.dlka_trace_start(256, L.stream_ptr(x)), "trace_start")
for _ in range(5):
    y = ops.conv3d_forward(x, w, b, 1, 1, 1, 1)
    gi, gw, gb = ops.conv3d_backward(x, w, go, 1, 1, 1, 1)
L.check(lib.dlka_trace_stop(), "trace_stop")
buf, ms, acc = create_string_buffer(512), c_float(), {}
for i in range(lib.dlka_trace_count()):
    L.check(lib.dlka_trace_get(i, buf, 512, byref(ms)), "trace_get")
    k = buf.value.decode().replace("void dlka::", "").replace("dlka::", "").split("(")[0]
    e = acc.setdefault(k, [0, 0.0]); e[0] += 1; e[1] += ms.value
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k[:70]:70s} x{v[0] // 5:<3d} {v[1] / v[0] * 1e3:8.1f} us")
if os.environ.get("NOCHECK"):
    sys.exit(0)
# slab check against torch (fp64 on the CPU): d = 0..3 of sample 1 (needs input d = 0..4)
xs, gos = x[1:2, :, :5].cpu().double(), go[1:2, :, :5].cpu().double()
ref = torch.nn.functional.conv3d(xs, w.cpu().double(), b.cpu().double(), 1, 1)
err = (y[1:2, :, :4].cpu().double() - ref[:, :, :4]).abs().max().item()
print("forward max abs err (slab)", err, "ref max", ref.abs().max().item())
refg = torch.nn.functional.conv_transpose3d(gos, w.cpu().double(), None, 1, 1)
errg = (gi[1:2, :, :4].cpu().double() - refg[:, :, :4]).abs().max().item()
print("grad_input max abs err (slab)", errg, "ref max", refg.abs().max().item())
assert err < 2e-4 and errg < 2e-4
# weight gradient (full size) against torch's own fp32 conv weight gradient on the device (MIOpen / ATen): the split contraction's error budget is ~1e-5 of max|gW|
try:
    refw = torch.nn.grad.conv3d_weight(x, w.shape, go, stride=1, padding=1)
    errw = (gw - refw).abs().max().item() / refw.abs().max().item()
    print("grad_weight max err / max|ref| (full size, vs torch fp32)", errw)
    assert errw < 1e-4
except RuntimeError as e:
    print("torch conv3d_weight unavailable:", repr(e)[:200])
