import os, torch, ctypes
from ctypes import byref
os.environ["DLKA_WGRAD_GATHER"] = "1"
from deformablelka_amd import _lib as L, ops
from tests import emu
import sys
if "--emu" in sys.argv:
    L._set_backend_for_tests(emu.load()); dev = "cpu"
else:
    dev = "cuda:0"
import deformablelka_amd as dk
from oracle import blocks
for (B, C, dims) in [(1, 32, (4, 4, 4)), (2, 32, (8, 8, 8)), (1, 64, (3, 4, 5)), (2, 32, (16, 16, 16))] + ([] if dev == "cpu" else [(2, 32, (32, 32, 32)), (2, 64, (16, 16, 16)), (2, 128, (8, 8, 8)), (2, 256, (4, 4, 4))]):
    torch.manual_seed(0)
    H, W, D = dims
    m = dk.LKA_Attention3d_deform(C); blocks.randomize_offsets_(m, std=0.3); m = m.to(dev)
    x = torch.randn(B, H * W * D, C, device=dev); gy = torch.randn_like(x)
    params = [p.detach().contiguous() for p in m.block_params()]
    y, saved = ops.lka3d_attention_tokens_forward(x, params, (H, W, D))
    lib = L.get_lib(); dt = L.dtype_code(x)
    wb = lib.dlka_lka3d_tokens_workspace_bytes(B, C, H, W, D, dt)
    ws = torch.full((wb + (1 << 20),), 0xAB, dtype=torch.uint8, device=dev)
    gx = torch.empty_like(x); grads = [torch.empty_like(t) for t in params]
    ps = ops._ptr_struct(L.Lka3dPtrs, L.LKA3D_FIELDS, params); gs = ops._ptr_struct(L.Lka3dPtrs, L.LKA3D_FIELDS, grads)
    rc = lib.dlka_lka3d_attention_tokens_backward(L.ptr(x), byref(ps), L.ptr(gy), L.ptr(saved), saved.numel(), L.ptr(gx), byref(gs), L.ptr(ws), wb + (1 << 20), B, C, H, W, D, dt, L.stream_ptr(x))
    assert rc == 0, rc
    if dev != "cpu": torch.cuda.synchronize()
    tail = ws[wb - 4096:].cpu()
    bad = (tail != 0xAB).nonzero()
    print(B, C, dims, "wb", wb, "touched bytes past the scratch region:", bad.numel(), (bad.min().item(), bad.max().item()) if bad.numel() else "")
