"""Cases shared by tests/test_ref_d3d_gpu.py (-m gpu: the reference's own native op, oracle/_ref/D3D.so, against the C oracle
and the HIP kernels) and tests/golden/make_ref_golden.py (records the reference's outputs as committed fixtures).

Shapes come from the reference's smoke scripts and the four stage shapes of BASELINE.json config 3."""
import torch

# name: (B, C, Cout, (D,H,W), k, stride, pad, dil, group, dg, offset mode, offset scale, im2col_step)
FULL = {
    # 3D/dcn/test.py:16-22,65 — DeformConvPack / DeformConv at (2,32,32,32,32), k=5 p=2 dense
    "test_py_k5_dense": (2, 32, 32, (32, 32, 32), 5, 1, 2, 1, 1, 1, "normal", 1.0, 64),
    # 3D/dcn/test.py:28 — k=5 p=2 groups=dim
    "test_py_k5_depthwise": (2, 32, 32, (32, 32, 32), 5, 1, 2, 1, 32, 1, "normal", 1.0, 64),
    # 3D/dcn/test_deform_conv_speed.py:155-156 — (1,32,32,32,32) k=3
    "speed_py_k3": (1, 32, 32, (32, 32, 32), 3, 1, 1, 1, 1, 1, "normal", 1.0, 64),
    # 3D/dcn/test_3d_deform_conv_params.py:17 — (1,64,16,16,16)
    "params_py_k3": (1, 64, 64, (16, 16, 16), 3, 1, 1, 1, 1, 1, "normal", 1.0, 64),
    # BASELINE.json config 3: the four stage shapes at B=2, offsets ~ 1 voxel (the regime bench.py times)
    "stage0_headline": (2, 32, 32, (32, 32, 32), 3, 1, 1, 1, 1, 1, "normal", 1.0, 64),
    "stage1": (2, 64, 64, (16, 16, 16), 3, 1, 1, 1, 1, 1, "normal", 1.0, 64),
    "stage2": (2, 128, 128, (8, 8, 8), 3, 1, 1, 1, 1, 1, "normal", 1.0, 64),
    "stage3": (2, 256, 256, (4, 4, 4), 3, 1, 1, 1, 1, 1, "normal", 1.0, 64),
    "stage0_wild": (2, 32, 32, (32, 32, 32), 3, 1, 1, 1, 1, 1, "wild", 1.0, 64),
}
SMALL = {
    "k3_normal": (2, 8, 8, (7, 6, 5), 3, 1, 1, 1, 1, 1, "normal", 1.0, 64),
    "k3_wild": (2, 8, 12, (6, 6, 6), 3, 1, 1, 1, 1, 1, "wild", 1.0, 64),
    "k3_integer": (1, 8, 8, (6, 6, 6), 3, 1, 1, 1, 1, 1, "integer", 1.0, 64),
    "k3_zero": (1, 8, 8, (6, 6, 6), 3, 1, 1, 1, 1, 1, "zero", 1.0, 64),
    "k5_dense": (1, 4, 6, (7, 7, 7), 5, 1, 2, 1, 1, 1, "normal", 1.0, 64),
    "k5_depthwise": (1, 8, 8, (7, 7, 7), 5, 1, 2, 1, 8, 1, "normal", 1.0, 64),
    "g2_dg2_ragged": (2, 8, 12, (9, 7, 11), (3, 2, 3), (2, 1, 1), (1, 0, 1), (1, 2, 1), 2, 2, "wild", 1.0, 64),
    "dil3": (1, 8, 8, (8, 8, 8), 3, 1, 3, 3, 1, 1, "normal", 1.0, 64),
    "im2col_step2": (4, 4, 4, (5, 5, 5), 3, 1, 1, 1, 1, 1, "normal", 1.0, 2),
    # Q1: deformable_col2im_cuda forwards pad_h in place of pad_w (cuh:447) — visible only when pad_h != pad_w
    "q1_pad_h_ne_pad_w": (1, 4, 4, (6, 6, 6), 3, 1, (1, 1, 2), 1, 1, 1, "normal", 1.0, 64),
    "single_voxel": (3, 4, 4, (1, 1, 1), 3, 1, 1, 1, 1, 1, "normal", 1.0, 64),
}


def make(case, seed=0):
    from tests import parity
    B, C, Cout, dims, k, s, p, d, g, dg, mode, scale, step = case
    x, off, w, b, go, (k3, s3, p3, d3) = parity.make_deform3d(B, C, Cout, dims, k, s, p, d, g, dg, mode, seed, scale)
    return dict(x=x, off=off, w=w, b=b, go=go, s=s3, p=p3, d=d3, g=g, dg=dg, step=step)


def run_ref(t, dev):
    """The reference's D3D.deform_conv_forward / _backward on the GPU (oracle/_ref/D3D.so)."""
    from oracle import ref
    x, w, b, off, go = (t[k].to(dev).contiguous() for k in ("x", "w", "b", "off", "go"))
    out = ref.deform_conv3d_forward(x, w, b, off, t["s"], t["p"], t["d"], t["g"], t["dg"], t["step"])
    gi, goff, gw, gb = ref.deform_conv3d_backward(x, w, b, off, go, t["s"], t["p"], t["d"], t["g"], t["dg"], t["step"])
    torch.cuda.synchronize()
    return [v.cpu() for v in (out, gi, goff, gw, gb)]


# ---------------------------------------------------------------------------------------------------------------------
# 2-D: the reference's OWN 3-D op pins the 2-D (torchvision, un-vendored) arithmetic.  With D = 1, kd = 1, pad_d = 0 and a zero depth
# offset the D3D kernels compute qd = 0 exactly -> floor 0, ld = 0, the upper-depth corner is outside the volume and dropped
# (3D/dcn/src/cuda/deform_im2col_cuda.cuh:26-72, 245-259): trilinear collapses to bilinear with the SAME guard (-1 < q < size), the same
# per-corner zeroing and the same floor-based one-sided derivative as torchvision 0.12's deform_conv2d.  The one torchvision deviation that
# stays restated (oracle/dlka_oracle_impl.h): its coordinate weight is unguarded, which differs from D3D's at q == -1 EXACTLY — those
# positions are excluded from the grad_offset comparison (`reach_only` below) and checked against the restatement separately.
# name: (B, C, Cout, H, W, (kh, kw), stride, pad, dil, group, offset groups, offset mode)
CASES_2D = {
    # the two depthwise deformable convs of the 2-D D-LKA block (2D/deformable_LKA/deformable_LKA.py:93-94), fast-path width (C % 32 == 0)
    "dlka_dw5": (2, 32, 32, 14, 14, (5, 5), 1, 2, 1, 32, 1, "normal"),
    "dlka_dw7_dil3": (2, 32, 32, 14, 14, (7, 7), 1, 9, 3, 32, 1, "normal"),
    "dlka_dw5_wild": (1, 64, 64, 12, 10, (5, 5), 1, 2, 1, 64, 1, "wild"),
    "dlka_dw7_dil3_integer": (1, 32, 32, 12, 12, (7, 7), 1, 9, 3, 32, 1, "integer"),
    # speed-test shapes of the reference (2D/deformable_LKA/deform_conv_speed.py:35-58): dense / depthwise 3x3
    "dense3": (2, 8, 12, 9, 11, (3, 3), 1, 1, 1, 1, 1, "normal"),
    "dense3_integer": (1, 8, 8, 7, 7, (3, 3), 1, 1, 1, 1, 1, "integer"),
    "dense3_zero": (1, 4, 4, 6, 6, (3, 3), 1, 1, 1, 1, 1, "zero"),
    "dw3": (1, 8, 8, 8, 8, (3, 3), 1, 1, 1, 8, 1, "normal"),
    "strided_g2_og2_wild": (2, 8, 12, 11, 9, (3, 3), 2, 1, 1, 2, 2, "wild"),
}
SMALL_2D = [n for n in CASES_2D]   # all of them are small enough to be recorded as fixtures


def make2d(case, seed=0):
    from tests import parity
    B, C, Cout, H, W, k, s, p, d, g, og, mode = case
    x, off, w, go = parity.make_deform2d(B, C, Cout, H, W, k, s, p, d, g, og, mode, seed)
    return dict(x=x, off=off, w=w, go=go, k=k, s=s, p=p, d=d, g=g, og=og, H=H, W=W)


def embed2d(t):
    """The D3D call that computes the 2-D case: depth axis of size 1, zero depth offsets, zero bias (D3D always adds one)."""
    from tests import parity
    kh, kw = t["k"]
    K = kh * kw
    s, p, d = t["s"], t["p"], t["d"]
    return dict(x=t["x"].unsqueeze(2), off=parity.embed_offsets_2d_in_3d(t["off"], K, t["og"]), w=t["w"].unsqueeze(2),
                b=torch.zeros(t["w"].shape[0]), go=t["go"].unsqueeze(2), s=(1, s, s), p=(0, p, p), d=(1, d, d), g=t["g"], dg=t["og"], step=64)


def project2d(t, ref3):
    """D3D outputs of the embedded call -> the 2-D op's (out, grad_input, grad_offset (dy, dx), grad_weight); also returns the depth-offset
    gradient, which has no 2-D counterpart."""
    out, gi, goff, gw, gb = ref3
    kh, kw = t["k"]
    K = kh * kw
    B, _, _, Ho, Wo = goff.shape
    g3 = goff.reshape(B, t["og"], K, 3, Ho, Wo)
    goff2 = g3[:, :, :, 1:3].reshape(B, t["og"] * 2 * K, Ho, Wo).contiguous()
    return [out.squeeze(2), gi.squeeze(2), goff2, gw.squeeze(2)], g3[:, :, :, 0]


def run_ref2d(t, dev):
    return project2d(t, run_ref(embed2d(t), dev))


def q_minus_one_mask(t):
    """[B, og*2K, Ho, Wo] bool: offset channels of samples with a coordinate at q == -1 exactly (torchvision's unguarded coordinate weight
    differs from D3D's guarded one there, and only there)."""
    kh, kw = t["k"]
    K, og = kh * kw, t["og"]
    off = t["off"]
    B, _, Ho, Wo = off.shape
    s, p, d = t["s"], t["p"], t["d"]
    o2 = off.reshape(B, og, K, 2, Ho, Wo)
    tj = (torch.arange(K) // kw).view(1, 1, K, 1, 1)
    tk = (torch.arange(K) % kw).view(1, 1, K, 1, 1)
    qy = (torch.arange(Ho).view(1, 1, 1, Ho, 1) * s - p + tj * d).float() + o2[:, :, :, 0]
    qx = (torch.arange(Wo).view(1, 1, 1, 1, Wo) * s - p + tk * d).float() + o2[:, :, :, 1]
    edge = (qy == -1) | (qx == -1)
    return edge[:, :, :, None].expand(B, og, K, 2, Ho, Wo).reshape(B, og * 2 * K, Ho, Wo)
