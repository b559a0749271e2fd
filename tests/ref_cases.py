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
