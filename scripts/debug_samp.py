"""Debug aid: the sample tensor S[tap][m][c] the grad_offset kernel stores for the weight gradient, read back from the workspace and compared with
a plain torch restatement of the sampling rule; plus a canary behind the workspace."""
import os, sys, torch
from ctypes import byref
sys.path.insert(0, ".")
from deformablelka_amd import _lib as L, ops
import deformablelka_amd as dk

emu_mode = "--emu" in sys.argv
if emu_mode:
    from tests import emu
    L._set_backend_for_tests(emu.load())
dev = "cpu" if emu_mode else "cuda:0"


def ref_samples(t, off, dims):
    """t [B,N,C] tokens (D,H,W order = dims), off [B,81,N] -> S [27][B*N][C] (3^3, pad 1, stride 1, dil 1; guard q > -1 && q < size)."""
    D, H, W = dims
    B, N, C = t.shape
    vol = t.reshape(B, D, H, W, C).double()
    idx = torch.arange(N, device=t.device)
    w0, h0, d0 = idx % W, (idx // W) % H, idx // (W * H)
    out = torch.zeros(27, B, N, C, dtype=torch.float64, device=t.device)
    for tap in range(27):
        ti, tj, tk = tap // 9, (tap // 3) % 3, tap % 3
        qd = (d0 + ti - 1).float()[None] + off[:, 3 * tap]
        qh = (h0 + tj - 1).float()[None] + off[:, 3 * tap + 1]
        qw = (w0 + tk - 1).float()[None] + off[:, 3 * tap + 2]
        inside = (qd > -1) & (qh > -1) & (qw > -1) & (qd < D) & (qh < H) & (qw < W)
        fd, fh, fw = qd.floor(), qh.floor(), qw.floor()
        ld, lh, lw = (qd - fd).double(), (qh - fh).double(), (qw - fw).double()
        zd, zh, zw = fd.long(), fh.long(), fw.long()
        acc = torch.zeros(B, N, C, dtype=torch.float64, device=t.device)
        bi = torch.arange(B, device=t.device)[:, None].expand(B, N)
        for q in range(8):
            cd, ch, cw = (q >> 2) & 1, (q >> 1) & 1, q & 1
            xd, xh, xw = zd + cd, zh + ch, zw + cw
            ok = inside & (xd >= 0) & (xd < D) & (xh >= 0) & (xh < H) & (xw >= 0) & (xw < W)
            wq = (ld if cd else 1 - ld) * (lh if ch else 1 - lh) * (lw if cw else 1 - lw)
            val = vol[bi, xd.clamp(0, D - 1), xh.clamp(0, H - 1), xw.clamp(0, W - 1)]
            acc += (wq * ok)[..., None] * val
        out[tap] = acc
    return out.reshape(27, B * N, C)


def align256(n):
    return (n + 255) // 256 * 256


cases = [(1, 32, (4, 4, 4)), (2, 32, (8, 8, 8))] if emu_mode else [(2, 32, (32, 32, 32)), (2, 64, (16, 16, 16)), (2, 128, (8, 8, 8)), (2, 256, (4, 4, 4)), (1, 64, (5, 6, 7))]
from oracle import blocks
for (B, C, dims) in cases:
    for mode in ("gather", "samp"):
        L.get_lib().dlka_lka3d_force_wgrad_gather(1 if mode == "gather" else 0)   # (the switch is process-wide and read once: use the setter)
        torch.manual_seed(0)
        H, W, D = dims
        N = H * W * D
        m = dk.LKA_Attention3d_deform(C); blocks.randomize_offsets_(m, std=0.3); m = m.to(dev)
        x = torch.randn(B, N, C, device=dev); gy = torch.randn_like(x)
        params = [p.detach().contiguous() for p in m.block_params()]
        y, saved = ops.lka3d_attention_tokens_forward(x, params, (H, W, D))
        lib = L.get_lib(); dt = L.dtype_code(x)
        wb = lib.dlka_lka3d_tokens_workspace_bytes(B, C, H, W, D, dt)
        ws = torch.full((wb + (1 << 20),), 0xAB, dtype=torch.uint8, device=dev)
        gx = torch.empty_like(x); grads = [torch.empty_like(t) for t in params]
        ps = ops._ptr_struct(L.Lka3dPtrs, L.LKA3D_FIELDS, params); gs = ops._ptr_struct(L.Lka3dPtrs, L.LKA3D_FIELDS, grads)
        rc = lib.dlka_lka3d_attention_tokens_backward(L.ptr(x), byref(ps), L.ptr(gy), L.ptr(saved), saved.numel(), L.ptr(gx), byref(gs), L.ptr(ws),
                                                      wb + (1 << 20), B, C, H, W, D, dt, L.stream_ptr(x))
        assert rc == 0, rc
        if dev != "cpu":
            torch.cuda.synchronize()
        tail = ws[wb - 4096:].cpu()
        bad = (tail != 0xAB).nonzero()
        print(f"[{mode}] B={B} C={C} dims={dims}: bytes touched behind the last region: {bad.numel()}")
        gW = grads[L.LKA3D_FIELDS.index("deform_w")].clone() if "deform_w" in L.LKA3D_FIELDS else None
        if mode == "gather":
            gW_ref = gW
            continue
        E4 = align256(B * N * C * 4)
        t = saved[3 * E4: 3 * E4 + B * N * C * 4].view(torch.float32).reshape(B, N, C)
        off = saved[4 * E4: 4 * E4 + B * 81 * N * 4].view(torch.float32).reshape(B, 81, N)
        sbytes = 27 * B * N * C * 4
        s0 = wb - 4096 - align256(sbytes)
        S = ws[s0: s0 + sbytes].view(torch.float32).reshape(27, B * N, C)
        Sr = ref_samples(t, off, (H, W, D))
        err = (S.double() - Sr).abs()
        print(f"    S vs torch restatement: max abs err {err.max().item():.3e} (max |S| {Sr.abs().max().item():.3f}); per-tap max err:",
              [f"{e:.1e}" for e in err.amax(dim=(1, 2)).tolist()][:27:3])
        if err.max() > 1e-4:
            bad = err > 1e-4
            print("    bad fraction", bad.float().mean().item(), "by 32-channel chunk:", [round(bad[:, :, 32 * k: 32 * k + 32].float().mean().item(), 4) for k in range(C // 32)])
            print("    by tap:", [round(bad[k].float().mean().item(), 3) for k in range(27)])
            rows = bad.any(dim=2).any(dim=0)
            print("    bad rows by (row % 32):", [round(rows[k::32].float().mean().item(), 2) for k in range(32)])
            print("    bad rows by 128-row block (first 16):", [round(rows[128 * k: 128 * k + 128].float().mean().item(), 2) for k in range(16)])
            i = bad.nonzero()[0].tolist()
            print("    first bad element (tap, row, c):", i, "got", S[i[0], i[1], i[2]].item(), "want", Sr[i[0], i[1], i[2]].item())
            # is the value some OTHER sample?  look for it among the reference samples of the same tap
            same = (Sr[i[0]].float() - S[i[0], i[1], i[2]]).abs() < 1e-6
            print("    same value found in the reference of that tap at (row, c):", same.nonzero()[:4].tolist())
        if gW is not None:
            print(f"    deform weight gradient, stored samples vs gathering kernel: rel err {((gW - gW_ref).abs().max() / gW_ref.abs().max()).item():.3e}")
