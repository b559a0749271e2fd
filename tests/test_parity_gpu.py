"""-m gpu: the parity tests proper.  HIP kernels on a real MI355X, called through the C-ABI, against the CPU oracle
on the same seeded inputs, plus the golden vectors of the reference's own modules and size-independent properties
at BASELINE.json's full sizes."""
import pytest
import torch

from tests import parity

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def hip_backend(oracle):
    from deformablelka_amd import _lib
    _lib._set_backend_for_tests(None)
    assert torch.cuda.is_available()
    lib = _lib.get_lib()   # fails loudly if libdlka_hip.so is missing — no fallback
    assert lib.dlka_abi_version() == 1
    yield


def _refresh_env():
    from deformablelka_amd import _lib
    _lib.get_lib().dlka_env_refresh()


@pytest.fixture(autouse=True)
def _fork_switches_follow_monkeypatch():
    """monkeypatch restores the environment when a test ends; the library's cached fork switches follow it."""
    yield
    _refresh_env()


D3 = [
    # reference smoke-script shapes and the four 3-D stage shapes, shrunk where the CPU oracle is slow
    (2, 32, 32, (12, 12, 12), 3, 1, 1, 1, 1, 1, "normal"),     # stage-0 config (3D/dcn/test_deform_conv_speed.py:155)
    (2, 64, 64, (8, 8, 8), 3, 1, 1, 1, 1, 1, "normal"),        # stage-1 config
    (2, 128, 128, (8, 8, 8), 3, 1, 1, 1, 1, 1, "uniform3"),    # stage-2: real size
    (2, 256, 256, (4, 4, 4), 3, 1, 1, 1, 1, 1, "wild"),        # stage-3: real size
    (1, 16, 16, (10, 10, 10), 5, 1, 2, 1, 1, 1, "normal"),     # k=5 p=2 dense (3D/dcn/test.py:65)
    (1, 16, 16, (10, 10, 10), 5, 1, 2, 1, 16, 1, "normal"),    # k=5 depthwise (3D/dcn/test.py:28)
    (2, 8, 12, (9, 7, 11), (3, 2, 3), (2, 1, 1), (1, 0, 1), (1, 2, 1), 2, 2, "wild"),  # ragged everything
    (1, 8, 8, (9, 9, 9), 3, 1, 1, 1, 1, 1, "integer"),         # exact integers incl. -1 and size
    (1, 8, 8, (9, 9, 9), 3, 1, 1, 1, 1, 1, "zero"),            # zero-init conv_offset (Q5)
    (1, 8, 8, (8, 8, 8), 3, 1, 3, 3, 1, 1, "normal"),          # dilation 3
    (3, 4, 4, (1, 1, 1), 3, 1, 1, 1, 1, 1, "normal"),          # degenerate single voxel
]


@pytest.mark.parametrize("case", D3)
def test_deform3d_vs_oracle(case):
    *cfg, mode = case
    parity.check_deform3d(DEV, *cfg, off_mode=mode)


D2 = [
    (2, 96, 96, 20, 20, (5, 5), 1, 2, 1, 96, 1, "normal"),     # 2-D stage (96, 56^2) shrunk
    (2, 96, 96, 20, 20, (7, 7), 1, 9, 3, 96, 1, "normal"),
    (1, 384, 384, 14, 14, (7, 7), 1, 9, 3, 384, 1, "wild"),    # (384, 14^2): real size
    (2, 16, 24, 13, 9, (3, 3), 2, 1, 1, 2, 2, "normal"),
    (1, 8, 8, 9, 9, (3, 3), 1, 1, 1, 1, 1, "integer"),
]


@pytest.mark.parametrize("case", D2)
def test_deform2d_vs_oracle(case):
    *cfg, mode = case
    parity.check_deform2d(DEV, *cfg, off_mode=mode, with_bias=(cfg[1] == 16))


CONV = [
    (2, 32, 32, (12, 12, 12), 5, 1, 2, 1, 32),                # dw 5^3
    (2, 32, 32, (12, 12, 12), 7, 1, 9, 3, 32),                # dw 7^3 dil 3
    (2, 32, 81, (10, 10, 10), 3, 1, 1, 1, 1),                 # offset-predict conv
    (2, 64, 64, (8, 8, 8), 1, 1, 0, 1, 1),                    # pointwise
    (2, 256, 81, (4, 4, 4), 3, 1, 1, 1, 1),                   # stage-3 offset conv, real size
    (1, 16, 16, (7, 8, 6), (3, 5, 5), 1, (1, 6, 6), (1, 3, 3), 16),   # ACDC anisotropic dw
    (1, 8, 12, (7, 6, 5), 3, 2, 1, 1, 2),                     # strided grouped
    (2, 96, 50, (1, 20, 20), (1, 5, 5), 1, (0, 2, 2), 1, 1),  # 2-D offset net 5x5 -> 50
    (1, 96, 98, (1, 20, 20), (1, 7, 7), 1, (0, 9, 9), (1, 3, 3), 1),  # 2-D offset net 7x7 dil 3 -> 98
    (2, 16, 16, (6, 9, 128), 3, 1, 1, 1, 1),                  # the net's full-resolution plumbing convs: 128-wide rows in registers (conv3_row_mfma_kernel)
    (1, 12, 16, (5, 4, 40), 3, 1, 1, 1, 1),                   # ... partial lane groups, ragged channels
    (1, 16, 16, (4, 5, 36), 3, 1, 1, 1, 1),                   # W % 8 != 0: the per-voxel-tile MFMA kernel / thread-per-voxel forward
    (2, 16, 16, (5, 32, 128), 3, 1, 1, 1, 1),                 # H % 8 == 0: the input-row-stationary weight gradient (conv3_bwd_weight_rows_b16_kernel), full 128-wide rows, four waves per plane
    (1, 5, 14, (2, 16, 40), 3, 1, 1, 1, 1),                   # ... ragged channels, a ragged second segment
]


@pytest.mark.parametrize("case", CONV)
def test_conv3d_vs_aten_cpu(case):
    parity.check_conv3d(DEV, *case)


GOLDEN = ["DeformConvPack_k3", "DeformConvPack_k5_dw_zero", "DeformConv_g2_dg2_nobias", "DeformConvPack_d_TW",
          "DeformConvPack_d_HW", "DeformConvPack_d_H", "DeformConvPack_Depth", "DeformConvPack_experimental", "LKA3d_deform", "LKA_Attention3d_deform",
          "DeformConv2d_k5_dw", "deformable_LKA_Attention", "TransformerBlock_3D_single_deform_LKA_train",
          "TransformerBlock_3D_single_deform_LKA_eval", "UnetResBlock_train"]


@pytest.mark.parametrize("name", GOLDEN)
def test_reference_module_golden(name):
    from tests.golden_checks import replay
    replay(name, DEV)


@pytest.mark.parametrize("C,dims", [(32, (16, 16, 16)), (64, (8, 8, 8)), (256, (4, 4, 4))])
def test_lka3d_block_vs_oracle(C, dims):
    """Whole fused block on the NCDHW entry point (``forward_volume``: one C-ABI call per direction, general per-op kernels) vs the oracle block
    on a real stage channel count — at the CONTRACT's tolerances (forward 1e-4 abs, gradients 1e-3 rel), with the cell-flip residual counted and
    the same-cells rerun, exactly as the token-layout test does (parity.check_lka3d_tokens)."""
    parity.check_lka3d_tokens(DEV, 2, C, dims, volume=True, report_offsets=True)


def test_full_size_stage0_properties():
    """BASELINE.json full size (C=32, 32^3, B=2): size-independent properties instead of the (slow) oracle.
    (1) zero offsets == plain conv3d computed by our own conv kernel and by the deformable kernel;
    (2) linearity of the deformable conv in (x, weight);
    (3) integer-shift offsets == shifted plain conv."""
    from deformablelka_amd import ops
    torch.manual_seed(0)
    B, C, N = 2, 32, 32
    x = torch.randn(B, C, N, N, N, device=DEV)
    w = torch.randn(C, C, 3, 3, 3, device=DEV) * 0.03
    b = torch.randn(C, device=DEV)
    off0 = torch.zeros(B, 81, N, N, N, device=DEV)
    y_def = ops.deform_conv3d_forward(x, w, b, off0, 3, 1, 1, 1, 1, 1)
    y_conv = ops.conv3d_forward(x, w, b, 1, 1, 1, 1)
    assert (y_def - y_conv).abs().max().item() < 1e-4
    # a CPU ATen slice check on one batch item keeps an independent anchor at full size
    ref = torch.nn.functional.conv3d(x[:1].cpu(), w.cpu(), b.cpu(), 1, 1)
    assert (y_conv[:1].cpu() - ref).abs().max().item() < 1e-4
    off = torch.randn(B, 81, N, N, N, device=DEV)
    x2 = torch.randn_like(x)
    zero_b = torch.zeros_like(b)
    ya = ops.deform_conv3d_forward(x, w, zero_b, off, 3, 1, 1, 1, 1, 1)
    yb = ops.deform_conv3d_forward(x2, w, zero_b, off, 3, 1, 1, 1, 1, 1)
    yab = ops.deform_conv3d_forward(x + 2 * x2, w, zero_b, off, 3, 1, 1, 1, 1, 1)
    assert (yab - (ya + 2 * yb)).abs().max().item() < 2e-4
    # integer shift (+1 along w for every tap) == conv of the volume shifted by one voxel with zero fill
    offs = torch.zeros(B, 27, 3, N, N, N, device=DEV)
    offs[:, :, 2] = 1.0
    ys = ops.deform_conv3d_forward(x, w, b, offs.reshape(B, 81, N, N, N), 3, 1, 1, 1, 1, 1)
    xs = torch.zeros_like(x)
    xs[..., :-1] = x[..., 1:]
    yref = ops.conv3d_forward(xs, w, b, 1, 1, 1, 1)
    # column w=0 differs by construction: its k=0 tap reads x[..., 0] in the deformable op but zero padding in the shifted conv
    assert (ys[..., 1:] - yref[..., 1:]).abs().max().item() < 1e-4


def test_error_behaviour():
    """Mirrors the reference's checks: contiguity (deform_conv_cuda.cu:41-42), device (:44-47), divisibility (:61-66)."""
    from deformablelka_amd import ops
    import deformablelka_amd as dk
    x = torch.randn(2, 4, 5, 5, 5, device=DEV)
    w = torch.randn(4, 4, 3, 3, 3, device=DEV)
    b = torch.randn(4, device=DEV)
    off = torch.zeros(2, 81, 5, 5, 5, device=DEV)
    with pytest.raises(RuntimeError, match="contiguous"):
        ops.deform_conv3d_forward(x.transpose(2, 3), w, b, off, 3, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError):
        ops.deform_conv3d_forward(x.cpu(), w, b, off, 3, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match="im2col_step"):
        ops.deform_conv3d_forward(x[:2].repeat(2, 1, 1, 1, 1)[:3].contiguous(), w, b, off.repeat(2, 1, 1, 1, 1)[:3].contiguous(), 3, 1, 1, 1, 1, 1, 2)
    with pytest.raises(ValueError):
        dk.DeformConv(5, 4, 3, 1, 1, groups=2)
    m = dk.DeformConv(4, 4, 3, 1, 1).to(DEV)
    with pytest.raises(AssertionError):
        m(x, off[:, :80])


# ---- channels-last fast path (MFMA implicit GEMM etc.) -------------------------------------------------------
CL_CONV = [
    (2, 32, 32, (12, 12, 12), 1, 0, 1, 1, False),
    (2, 32, 81, (10, 10, 10), 3, 1, 1, 1, True),      # offset-predict conv (planar offsets)
    (2, 64, 81, (8, 8, 8), 3, 1, 1, 1, True),
    (2, 256, 81, (4, 4, 4), 3, 1, 1, 1, True),        # stage-3, real size (split-K path)
    (2, 128, 128, (8, 8, 8), 1, 0, 1, 1, False),      # stage-2 pointwise, real size
    (2, 256, 256, (4, 4, 4), 1, 0, 1, 1, False),      # stage-3 pointwise, real size
    (2, 32, 32, (12, 12, 12), 5, 2, 1, 32, False),
    (2, 32, 32, (12, 12, 12), 7, 9, 3, 32, False),
    (2, 128, 128, (8, 8, 8), 7, 9, 3, 128, False),    # stage-2 dw7, real size
    (2, 256, 256, (4, 4, 4), 5, 2, 1, 256, False),    # stage-3 dw5, real size
]


@pytest.mark.parametrize("case", CL_CONV)
def test_conv3d_cl(case):
    *cfg, planar = case
    parity.check_conv3d_cl(DEV, *cfg, planar=planar)


@pytest.mark.parametrize("case", [(2, 32, 32, (12, 12, 12), "normal"), (2, 64, 64, (8, 8, 8), "uniform3"),
                                  (2, 128, 128, (8, 8, 8), "wild"), (2, 256, 256, (4, 4, 4), "normal"),
                                  (1, 32, 32, (9, 9, 9), "integer"), (1, 32, 32, (5, 6, 7), "zero")])
def test_deform3d_cl(case):
    B, C, Cout, dims, mode = case
    parity.check_deform3d_cl(DEV, B, C, Cout, dims, off_mode=mode)


@pytest.mark.parametrize("C,dims", [(32, (16, 16, 16)), (64, (8, 8, 8)), (64, (16, 16, 16)), (128, (8, 8, 8)), (256, (4, 4, 4)), (32, (5, 6, 7))])
def test_lka3d_tokens_block_vs_oracle(C, dims):
    parity.check_lka3d_tokens(DEV, 2, C, dims)


# offset_std = std of conv_offset.weight that makes the PREDICTED offsets ~1 voxel at that stage width — the regime bench.py times
# (deformablelka_amd/stack.py:_init_block calibration: 3 * calib / sqrt(27 C))
HEADLINE = [(32, (32, 32, 32), 0.376), (64, (16, 16, 16), 0.380), (128, (8, 8, 8), 0.490), (256, (4, 4, 4), 0.451)]


@pytest.mark.parametrize("C,dims,wstd", HEADLINE)
def test_lka3d_tokens_block_headline_shapes_vs_oracle(C, dims, wstd):
    """BASELINE.json config 3 at FULL size (B=2; stage 0 = C 32, 32^3 dominates the metric), offsets ~1 voxel: the token fast
    path against the oracle block, forward and every gradient."""
    parity.check_lka3d_tokens(DEV, 2, C, dims, offset_std=wstd, report_offsets=True)


# BASELINE.json config 5 as written: 40x224x224 tiles only divide through the ACDC stem (1,4,4) (acdc/model_components.py:21) -> per-tile stage
# shapes 40x56x56 / 20x28x28 / 10x14x14 / 5x7x7 with the ACDC variant's anisotropic depthwise pair (acdc/transformerblock.py:213-237)
@pytest.mark.parametrize("C,dims,wstd", HEADLINE)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_forward_is_bitwise_reproducible_at_the_headline_shapes(C, dims, wstd, dtype):
    """forward outputs and predicted offsets bitwise equal across 5 runs, all four config-3 stage shapes, B = 2, ~1-voxel offsets, both dtypes."""
    parity.check_forward_reproducible(DEV, 2, C, dims, dtype, runs=5, offset_std=wstd, expect_kw=(dims[0] < 32))


@pytest.mark.parametrize("C,dims,wstd", HEADLINE)
def test_tblock3d_forward_is_bitwise_reproducible_at_the_headline_shapes(C, dims, wstd):
    """The wrapper block (the path the trainers call) in TRAINING mode: output, BatchNorm batch statistics and every saved activation bitwise equal across 4 runs."""
    parity.check_tblock3d_forward_reproducible(DEV, 2, C, dims, True, runs=4)


CONFIG5 = [(32, (40, 56, 56), 0.376), (64, (20, 28, 28), 0.380), (128, (10, 14, 14), 0.490), (256, (5, 7, 7), 0.451)]


@pytest.mark.parametrize("C,dims,wstd", CONFIG5)
def test_lka3d_tokens_block_config5_acdc_stage_shapes_vs_oracle(C, dims, wstd):
    """The non-cubic stage shapes of a 40x224x224 tile (125 k voxels at C = 32: 4x stage 0 of the Synapse patch) with the ACDC depthwise kernels
    on the token fast path, B = 1 (sliding-window inference is forward-only; the gradients are checked all the same)."""
    parity.check_lka3d_tokens(DEV, 1, C, dims, offset_std=wstd, report_offsets=True, acdc=True)


@pytest.mark.parametrize("C,dims,wstd", HEADLINE)
def test_lka3d_tokens_bf16_headline_shapes_vs_oracle(C, dims, wstd):
    """The north_star dtype: bf16 activations (fp32 parameters / offsets / accumulation) on the token fast path at the FULL stage sizes
    (B=2, offsets ~1 voxel) — every output and gradient within 2e-2 of the bf16-storage oracle, and of the fp32 oracle wherever the
    quantity is not exposed to grad_offset's discontinuity (tests/parity.py:check_lka3d_tokens_bf16)."""
    parity.check_lka3d_tokens_bf16(DEV, 2, C, dims, offset_std=wstd, report=True)


def test_lka3d_tokens_bf16_autocast_policy():
    """fp32 tensors inside torch.autocast(dtype=bfloat16): the block's policy moves the activations to bf16, parameters stay fp32."""
    parity.check_lka3d_tokens_bf16(DEV, 2, 64, (8, 8, 8), via_autocast=True, report=True)


@pytest.mark.parametrize("C,hw", [(96, 56), (192, 28), (384, 14)])
def test_lka2d_attention_real_shapes_vs_oracle(C, hw):
    """The three decoder shapes of the 224^2 2-D net (B = 2 here, 24 in training): the channels-last 2-D block — MFMA offset nets,
    gather-layout depthwise deformable convs (cl_ddw2d.hip) — forward and every gradient against the oracle block."""
    parity.check_lka2d_attention(DEV, 2, C, hw, hw, report=True)


@pytest.mark.parametrize("sel,C,hw", [("tiles", 192, 28), ("tiles", 384, 14), ("window", 96, 56)])
def test_lka2d_grad_input_forced_generation_real_shapes(sel, C, hw, monkeypatch):
    """The grad_input kernel the launcher would NOT pick at this shape (DLKA_DDW2D_GX): both generations hold the contract at all three decoder shapes."""
    monkeypatch.setenv("DLKA_DDW2D_GX", sel)
    parity.check_lka2d_attention(DEV, 2, C, hw, hw)


def test_lka2d_attention_config2_batch24_vs_oracle():
    """BASELINE.json config 2's own batch (B = 24) at the widest-image decoder shape (96, 56^2): contract tolerances, flips counted, same-cells rerun."""
    parity.check_lka2d_attention(DEV, 24, 96, 56, 56, report=True)


@pytest.mark.parametrize("C,hw", [(384, 14), (192, 28), (96, 56)])
def test_lka2d_attention_bf16_real_shapes_vs_oracle(C, hw):
    """BASELINE.json config 2 ("224x224 bf16 training"): the 2-D block with bf16 activations (fp32 parameters / offsets / accumulation, fp32
    offset-determining chain) at the three decoder shapes — forward and every gradient within 2e-2 of the fp32 oracle block."""
    parity.check_lka2d_attention_bf16(DEV, 2, C, hw, hw, report=True)


def test_lka2d_attention_bf16_autocast_policy():
    """fp32 tensors inside torch.autocast(dtype=bfloat16): the 2-D block's policy hands the kernels bf16 activations (DLKA_BF16) and fp32 parameters —
    the output comes back bf16, the parameter gradients fp32."""
    import deformablelka_amd as dk
    from oracle import blocks
    torch.manual_seed(0)
    m = dk.deformable_LKA_Attention(96).to(DEV)
    blocks.randomize_offsets_(m, std=0.03)
    x = torch.randn(2, 96, 20, 17, device=DEV, requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
    assert y.dtype == torch.bfloat16
    y.float().sum().backward()
    assert x.grad.dtype == torch.float32 and all(p.grad.dtype == torch.float32 and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    y32 = m(x.detach())
    assert y32.dtype == torch.float32 and parity.rel_err(y, y32) < 2e-2


def test_lka2d_attention_fast_path_equals_general_path():
    """Same inputs through the channels-last fast path and (dlka_lka2d_force_general) the general NCHW kernels."""
    import deformablelka_amd as dk
    from deformablelka_amd import _lib
    from oracle import blocks
    torch.manual_seed(0)
    m = dk.deformable_LKA_Attention(96).to(DEV)
    blocks.randomize_offsets_(m, std=0.03)
    x = torch.randn(3, 96, 20, 17, device=DEV)
    gy = torch.randn_like(x)

    def run():
        xs = x.clone().requires_grad_(True)
        for p in m.parameters():
            p.grad = None
        y = m(xs)
        y.backward(gy)
        return [y.detach(), xs.grad] + [p.grad.clone() for p in m.parameters()]
    fast = run()
    old = _lib.get_lib().dlka_lka2d_force_general(1)
    try:
        gen = run()
    finally:
        _lib.get_lib().dlka_lka2d_force_general(old)
    assert not torch.equal(fast[0], gen[0])
    parity.assert_close("y", fast[0], gen[0], atol=2e-4)
    for i, (a, b) in enumerate(zip(fast[1:], gen[1:])):
        parity.assert_close(f"grad {i}", a, b, rtol=8e-3)


def test_deform3d_cl_headline_shape_vs_oracle():
    """The deformable conv of the headline shape (C=32, 32^3, B=2, offsets N(0,1)): forward + all four gradients vs the oracle."""
    parity.check_deform3d_cl(DEV, 2, 32, 32, (32, 32, 32), off_mode="normal")


@pytest.mark.parametrize("C,dims", [(32, (32, 32, 32)), (64, (16, 16, 16)), (32, (9, 7, 11))])
def test_gx_fixed_point_window_worst_case(C, dims):
    """Provable overflow bound of the fixed-point grad_input window at the full stage sizes: R*K same-sign maximal contributions per cell."""
    parity.check_deform3d_cl_gx_worst_case(DEV, C, dims, B=2 if dims[0] > 9 else 1)


@pytest.mark.parametrize("C,dims", [(32, (32, 32, 32)), (64, (16, 16, 16))])
def test_gx_fixed_point_window_error_vs_fp64_window(C, dims):
    parity.check_deform3d_cl_gx_fixed_vs_fp64(DEV, 2, C, dims)


def test_tokens_full_size_stage0_matches_general_path():
    """BASELINE.json full size (C=32, 32^3, B=2): the token-layout fast path against our own general NCDHW path
    (which is pinned to the oracle above) — forward and all gradients."""
    import deformablelka_amd as dk
    from oracle import blocks
    torch.manual_seed(0)
    B, C, n = 2, 32, 32
    m = dk.LKA_Attention3d_deform(C)
    blocks.randomize_offsets_(m, std=0.02)
    m = m.to(DEV)
    x = torch.randn(B, n * n * n, C, device=DEV)
    gy = torch.randn(B, n * n * n, C, device=DEV)
    xa = x.clone().requires_grad_(True)
    ya = m(xa, B, C, n, n, n)                      # token fast path
    ya.backward(gy)
    ga = {k: p.grad.clone() for k, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = None
    xb = x.clone().requires_grad_(True)
    vb = xb.permute(0, 2, 1).reshape(B, C, n, n, n)
    yb = m.forward_volume(vb).reshape(B, C, n * n * n).permute(0, 2, 1)   # general NCDHW path
    yb.backward(gy)
    parity.assert_close("y", ya, yb.detach(), atol=2e-4)
    parity.assert_close("gx", xa.grad, xb.grad, rtol=2e-3)
    for k, p in m.named_parameters():
        parity.assert_close("grad " + k, ga[k], p.grad, rtol=2e-3)


def test_hipgraph_replay_reproduces_eager():
    """bench.py replays the 21-block step from a hipGraph: every replay must reproduce the eager result (a graph whose
    zero-fills were hipMemsetAsync nodes did not, from the second replay on — the round-2 investigation, profiles/design_history_r01_r03.md)."""
    from deformablelka_amd.stack import DLKABlockStack
    st = DLKABlockStack(2, stages=((32, (8, 8, 8), 1), (64, (8, 8, 8), 1), (256, (4, 4, 4), 1)), device="cuda:0", seed=3)
    st.forward_backward()
    torch.cuda.synchronize()
    ref = [g.clone() for b in st.blocks for g in b.grads] + [b.gx.clone() for b in st.blocks] + [b.y.clone() for b in st.blocks]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        st.forward_backward()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        st.forward_backward()
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        cur = [g_.clone() for b in st.blocks for g_ in b.grads] + [b.gx.clone() for b in st.blocks] + [b.y.clone() for b in st.blocks]
        for a, c in zip(ref, cur):
            assert torch.isfinite(c).all()
            scale = max(float(a.abs().max()), 1e-6)
            assert float((a - c).abs().max()) <= 2e-3 * scale   # atomics order only
    assert st.health()["finite"]


def test_stack_grouped_folds_on_the_side_stream_equal_the_single_launch(monkeypatch):
    """With a sealed plan and the weight-gradient stream, DLKABlockStack.backward folds the partial sums of every few blocks right behind their weight
    gradients on the side stream (DLKA_STACK_FINALIZE_GROUP, default 1) instead of one launch after the join (= 0): same folds, same gradients — incl. a
    ragged last group (7 blocks) and groups of one."""
    from deformablelka_amd.stack import DLKABlockStack
    stages = ((32, (8, 8, 8), 3), (64, (4, 4, 4), 2), (128, (4, 4, 4), 2))
    res = {}
    for grp in ("0", "3", "2", "1"):
        monkeypatch.setenv("DLKA_STACK_FINALIZE_GROUP", grp)
        st = DLKABlockStack(2, stages=stages, device="cuda:0", seed=5)
        for _ in range(3):   # the first pass records the plan; the later ones run it sealed
            st.forward_backward()
        torch.cuda.synchronize()
        assert st._fin_sealed
        res[grp] = [g.clone() for b in st.blocks for g in b.grads]
    for grp in ("3", "2", "1"):
        for a_, c_ in zip(res["0"], res[grp]):
            assert torch.isfinite(c_).all()
            scale = max(float(a_.abs().max()), 1e-6)
            assert float((a_ - c_).abs().max()) <= 2e-3 * scale, grp   # atomics order only


def test_stack_prepare_then_backward_is_stream_safe():
    """``DLKABlockStack.prepare()`` is a public method ("after every parameter update"): used on its own and followed by ``backward()`` it must
    leave nothing in flight on the side stream (the split preparation is private to ``forward()``; round-3 advice).  New parameters -> prepare() ->
    backward() with the saved activations of the earlier forward pass must equal the same sequence with everything on one stream."""
    import os
    from deformablelka_amd.stack import DLKABlockStack
    stages = ((32, (8, 8, 8), 2), (128, (4, 4, 4), 1), (256, (4, 4, 4), 1))

    def run(overlap):
        st = DLKABlockStack(2, stages=stages, device="cuda:0", seed=5, overlap_wgrad=overlap)
        st.forward_backward()
        st.flat_params.mul_(1.01)            # "parameter update"
        st.prepare()
        st.backward()
        torch.cuda.synchronize()
        return [g.clone() for b in st.blocks for g in b.grads] + [b.gx.clone() for b in st.blocks]

    a, c = run(True), run(False)
    for u, v in zip(a, c):
        assert torch.isfinite(u).all()
        assert float((u - v).abs().max()) <= 2e-3 * max(float(v.abs().max()), 1e-6)


# ---- the wrapper block TransformerBlock_3D_single_deform_LKA -----------------------------------------------
@pytest.mark.parametrize("case", [(2, 32, 4099, True, True), (1, 64, 5000, False, False), (3, 128, 999, True, False), (1, 256, 2100, False, True)])
def test_layernorm_tokens(case):
    B, C, N, planar, pos = case
    parity.check_layernorm_tokens(DEV, B, C, N, planar, pos)


@pytest.mark.parametrize("case", [(30011, 32, True, True), (7777, 64, True, False), (13000, 256, False, True), (6400, 128, False, False)])
def test_batchnorm_cl(case):
    parity.check_batchnorm_cl(DEV, *case)


@pytest.mark.parametrize("ratio", [20.0, 500.0])
def test_batchnorm_cl_large_mean(ratio):
    """|mean| >> std: the one-pass E[x^2] - mean^2 form would lose the variance (24 % off at ratio 500); the pivoted sums must not."""
    parity.check_batchnorm_cl(DEV, 30011, 32, True, True, mean_over_std=ratio)


def test_scale_residual_and_channel_scale():
    parity.check_scale_residual(DEV, 20003, 64)


@pytest.mark.parametrize("C,dims,training,pos", [(32, (16, 16, 16), True, True), (64, (8, 8, 8), False, False), (128, (6, 5, 7), True, False),
                                                  (256, (4, 4, 4), True, True)])
def test_tblock3d_vs_oracle(C, dims, training, pos):
    parity.check_tblock3d(DEV, 2, C, dims, training, pos)


@pytest.mark.parametrize("C,dims", [(32, (32, 32, 32)), (64, (16, 16, 16)), (128, (8, 8, 8)), (256, (4, 4, 4))])
def test_tblock3d_mixed_bf16_real_shapes(C, dims):
    """The wrapper block under torch.autocast(bfloat16) at the four stage shapes of the 64x128x128 patch: fp32 wrapper, DLKA_BF16 attention inside
    (dlka_tblock3d_* dtype = DLKA_BF16) — output and every gradient within 2e-2 of the fp32 oracle block and of the bf16-storage model."""
    parity.check_tblock3d_mixed_bf16(DEV, 2, C, dims, report=True)
    parity.check_tblock3d_mixed_bf16(DEV, 2, C, dims, report=True, bn_bias=0.0)   # BatchNorm bias 0: the regime a freshly initialised net trains in


def test_tblock3d_chain():
    parity.check_tblock3d(DEV, 1, 32, (6, 8, 10), True, True, chain=True)


def test_tblock3d_acdc_variant_config5_stage():
    """The wrapper block around the ACDC variant at a config-5 stage shape (C = 64, 20x28x28), eval mode as in sliding-window inference."""
    parity.check_tblock3d(DEV, 1, 64, (20, 28, 28), False, True, offset_std=CONFIG5[1][2], report=True, acdc=True)


def test_tblock3d_headline_stage_one_voxel_offsets():
    """The wrapper block at the stage the metric is dominated by — (C = 32, 32^3, B = 2), training mode, pos_embed — with offsets ~1 voxel
    (the wrapper tests ran small volumes and offset_std = 0.02 only)."""
    parity.check_tblock3d(DEV, 2, 32, (32, 32, 32), True, True, offset_std=HEADLINE[0][2], report=True)


@pytest.mark.parametrize("C,dims,dtype", [(32, (32, 32, 32), torch.float32), (64, (16, 16, 16), torch.float32), (256, (4, 4, 4), torch.float32),
                                          (32, (16, 16, 16), torch.bfloat16), (64, (5, 6, 7), torch.float32), (128, (3, 5, 7), torch.bfloat16)])
def test_lka3d_tokens_weight_gradient_from_stored_samples(C, dims, dtype):
    parity.check_lka3d_tokens_sample_handover(DEV, 2, C, dims, dtype=dtype)


@pytest.mark.parametrize("C,dims,mode", [(32, (32, 32, 32), "normal"), (64, (16, 16, 16), "normal"), (32, (16, 16, 16), "wild"), (32, (8, 8, 8), "uniform3"),
                                         (64, (5, 6, 7), "normal")])
def test_deform3d_cl_gx_second_generation_fixed_point_kernel(C, dims, mode):
    parity.check_deform3d_cl_gx_fx2_vs_fx1(DEV, 2, C, dims, mode)


@pytest.mark.parametrize("dims,dtype", [((32, 32, 32), torch.float32), ((5, 6, 7), torch.float32), ((16, 16, 16), torch.bfloat16)])
def test_lka3d_tokens_pointwise_pair_equals_two_launches(dims, dtype):
    parity.check_lka3d_tokens_pointwise_pair(DEV, 2, dims, dtype)



# ---- planar plumbing of the full net at its real shapes (csrc/planar_ops.hip) against torch's own GPU ops --------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("C", [16])
def test_batchnorm_planar_full_resolution_vs_torch(C):
    """The five BatchNorm3d layers of encoder1 / decoder2 at 2 x C x 64 x 128 x 128 through network.BatchNorm3d (HIP) against nn.BatchNorm3d
    (torch / MIOpen) on the same tensors: output, running statistics, all gradients."""
    import torch.nn as nn
    from deformablelka_amd.network import BatchNorm3d
    torch.manual_seed(0)
    x = (torch.randn(2, C, 64, 128, 128, device="cuda") * 1.7 + 0.4)
    gy = torch.randn_like(x)
    a, b = BatchNorm3d(C).cuda(), nn.BatchNorm3d(C).cuda()
    with torch.no_grad():
        a.weight.normal_(1, 0.2); a.bias.normal_(0, 0.2)
    b.load_state_dict(a.state_dict())
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = a(xa), b(xb)
    ya.backward(gy); yb.backward(gy)
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-4)
    assert torch.allclose(a.running_mean, b.running_mean, rtol=1e-5, atol=1e-6) and torch.allclose(a.running_var, b.running_var, rtol=1e-4, atol=1e-6)
    assert int(a.num_batches_tracked) == 1
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-3, atol=1e-5)
    assert torch.allclose(a.weight.grad, b.weight.grad, rtol=2e-3, atol=0.5) and torch.allclose(a.bias.grad, b.bias.grad, rtol=2e-3, atol=0.5)
    a.eval(); b.eval()
    assert torch.allclose(a(x), b(x), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_instancenorm_planar_full_resolution_vs_torch():
    """network.InstanceNorm3d (the net's default norm: the UnetResBlock norms of encoder1 / decoder2 at 2 x 16 x 64 x 128 x 128) against
    nn.InstanceNorm3d: output and input gradient, training and evaluation mode."""
    import torch.nn as nn
    from deformablelka_amd.network import InstanceNorm3d
    torch.manual_seed(0)
    x = (torch.randn(2, 16, 64, 128, 128, device="cuda") * 1.7 + 0.4)
    gy = torch.randn_like(x)
    a, b = InstanceNorm3d(16).cuda(), nn.InstanceNorm3d(16).cuda()
    assert not list(a.state_dict().keys())
    for mode in (True, False):
        a.train(mode); b.train(mode)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = a(xa), b(xb)
        ya.backward(gy); yb.backward(gy)
        assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-4)
        assert torch.allclose(xa.grad, xb.grad, rtol=1e-3, atol=1e-5)


@pytest.mark.gpu
def test_instancenorm_followed_by_inplace_ops_vs_torch():
    """UnetResBlock's pattern around the planar InstanceNorm3d (dynunet_block.py:55-68): in-place LeakyReLU on the norm's output, ``out += residual``, in-place LeakyReLU
    again.  The norm returns its result in x's own shape from inside its autograd Function (no caller-side view of the op's output), so the in-place ops neither fail nor
    detour through CopySlices; output and both input gradients against the stock layers."""
    import torch.nn as nn
    from deformablelka_amd.network import InstanceNorm3d
    torch.manual_seed(0)
    x = torch.randn(2, 16, 16, 32, 32, device="cuda") * 1.3 + 0.2
    res = torch.randn_like(x)
    gy = torch.randn_like(x)
    outs = []
    for norm in (InstanceNorm3d(16).cuda(), nn.InstanceNorm3d(16).cuda()):
        act = nn.LeakyReLU(0.01, inplace=True)
        xa, ra = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        out = act(norm(xa * 1.0))
        out = norm(out * 1.0)
        out += ra * 1.0
        out = act(out)
        out.backward(gy)
        outs.append((out.detach(), xa.grad, ra.grad))
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("groups,shape", [(1, (2, 32, 32, 32, 32)), (2, (3, 8, 16, 64, 64)), (32, (2, 64, 16, 16, 16))])
def test_groupnorm_long_rows_vs_torch(groups, shape):
    """network.GroupNorm (the stem's one-group norm over 2 x 32 x 32^3: planar statistics kernels + affine; the short-row case stays on the stock
    layer) against nn.GroupNorm with the same parameters: output, input gradient, affine gradients."""
    import torch.nn as nn
    from deformablelka_amd.network import GroupNorm
    torch.manual_seed(0)
    C = shape[1]
    a, b = GroupNorm(groups, C).cuda(), nn.GroupNorm(groups, C).cuda()
    with torch.no_grad():
        a.weight.copy_(torch.randn(C) * 0.3 + 1.0); a.bias.copy_(torch.randn(C) * 0.2)
    b.load_state_dict(a.state_dict())
    x = torch.randn(*shape, device="cuda") * 1.7 + 0.4
    x[:, 0] += 30.0                      # a channel far from the group's mean
    gy = torch.randn_like(x)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = a(xa), b(xb)
    ya.backward(gy); yb.backward(gy)
    # fp64 restatement as the arbiter of both
    xd = x.double().view(shape[0], groups, -1)
    xh = ((xd - xd.mean(-1, keepdim=True)) / torch.sqrt(xd.var(-1, unbiased=False, keepdim=True) + a.eps)).view(shape)
    wv = a.weight.detach().double().view(1, C, 1, 1, 1)
    yr = xh * wv + a.bias.detach().double().view(1, C, 1, 1, 1)
    assert torch.allclose(ya.double(), yr, rtol=1e-4, atol=1e-4)
    assert torch.allclose(ya, yb, rtol=1e-4, atol=2e-4)
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-3, atol=2e-5)
    gw = (gy.double() * xh).sum((0, 2, 3, 4))
    assert torch.allclose(a.weight.grad.double(), gw, rtol=1e-4, atol=1e-2)
    assert torch.allclose(a.bias.grad, b.bias.grad, rtol=1e-4, atol=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,dims", [(16, 14, (64, 128, 128)), (32, 14, (32, 32, 32)), (1, 16, (64, 128, 128))])
def test_pointwise_planar_conv_real_shapes_vs_torch(cin, cout, dims):
    """The 1x1x1 convs of the plumbing (output heads, conv3 of encoder1) through network.Convolution (HIP) against the same weights as one fp64 matmul."""
    from deformablelka_amd.network import Convolution
    torch.manual_seed(1)
    conv = Convolution(cin, cout, 1, 1, bias=True).cuda()
    x = torch.randn(2, cin, *dims, device="cuda", requires_grad=True)
    gy = torch.randn(2, cout, *dims, device="cuda")
    y = conv(x)
    y.backward(gy)
    w, b = conv.conv.weight.detach().double().reshape(cout, cin), conv.conv.bias.detach().double()
    xr = x.detach().double().reshape(2, cin, -1)
    yr = torch.matmul(w, xr) + b.view(1, -1, 1)
    gyr = gy.double().reshape(2, cout, -1)
    assert torch.allclose(y.double().reshape(2, cout, -1), yr, rtol=1e-5, atol=1e-5)
    assert torch.allclose(x.grad.double().reshape(2, cin, -1), torch.matmul(w.t(), gyr), rtol=1e-5, atol=1e-5)
    gw = torch.einsum("bon,bin->oi", gyr, xr)
    assert torch.allclose(conv.conv.weight.grad.double().reshape(cout, cin), gw, rtol=1e-4, atol=1e-4 * float(gw.abs().max()))
    assert torch.allclose(conv.conv.bias.grad.double(), gyr.sum((0, 2)), rtol=1e-4, atol=1e-4 * float(gyr.sum((0, 2)).abs().max()))


def test_pmc_traffic_file_names_the_step_kernels():
    """bench.py quotes `roofline.traffic` from profiles/pmc_traffic_block.json (committed rocprofv3 --pmc passes), keyed by kernel name: the file is only
    evidence while the names at HEAD match it (round-3 verdict, weak #8).  The three deformable kernels of the stage-0 fp32 block as THIS build launches
    them must each have an entry."""
    import json
    import os
    from ctypes import byref, c_float, create_string_buffer
    from deformablelka_amd import _lib as L
    from deformablelka_amd.stack import DLKABlockStack
    blob = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic_block.json")))
    st = DLKABlockStack(2, stages=((32, (32, 32, 32), 1),), device="cuda:0", seed=1, overlap_wgrad=False)
    st.forward_backward()
    torch.cuda.synchronize()
    lib = L.get_lib()
    stream = torch.cuda.current_stream().cuda_stream
    L.check(lib.dlka_trace_start(512, stream), "trace_start")
    try:
        st.forward_backward()
    finally:
        rc = lib.dlka_trace_stop()
    L.check(rc, "trace_stop")
    buf, ms = create_string_buffer(512), c_float()
    names = set()
    for i in range(lib.dlka_trace_count()):
        L.check(lib.dlka_trace_get(i, buf, 512, byref(ms)), "trace_get")
        names.add(buf.value.decode().replace("void dlka::", "").replace("dlka::", "").split("(")[0])
    have = set(blob["stage0_f32"])
    for frag in ("cl_deform_gx_fx2_kernel", "cl_deform_goff16_kernel", "cl_deform_fwd16_kernel"):
        mine = [n for n in names if n.startswith(frag)]
        assert mine, (frag, sorted(names))
        assert all(n in have for n in mine), f"profiles/pmc_traffic_block.json is stale: {mine} not in {sorted(k for k in have if k.startswith(frag))} — re-run scripts/pmc_block.sh"


@pytest.mark.parametrize("waves", [None, "4", "8", "42"])
def test_conv_brick_data_gradient_stage0(waves, monkeypatch):
    """cl_conv_brick_kernel at the shape it exists for — the offset-predict conv's data gradient at (32, 32^3), B = 2, planar grad_out — against the fp64 conv
    (1e-3 relative) and against cl_conv_wave_kernel (same products, another summation order), for the default tile and each alternative (DLKA_CONV_BRICK_WAVES);
    the launch counter proves which kernel produced the result."""
    from deformablelka_amd import _lib, ops
    lib = _lib.get_lib()
    if waves:
        monkeypatch.setenv("DLKA_CONV_BRICK_WAVES", waves)
    monkeypatch.setenv("DLKA_CONV_BRICK", "2")   # the data gradient's brick kernel only
    n0 = lib.dlka_conv_brick_launch_count()
    parity.check_conv3d_cl(DEV, 2, 32, 81, (32, 32, 32), 3, 1, 1, 1, planar=True, seed=11)
    assert lib.dlka_conv_brick_launch_count() == n0 + 1
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 32, 32, 32, generator=gen).to(DEV)
    w = (torch.randn(81, 32, 3, 3, 3, generator=gen) * 0.05).to(DEV)
    go = torch.randn(2, 81, 32, 32, 32, generator=gen).to(DEV)
    g_brick = ops.conv3d_backward_cl(x, w, go, 1, 1, 1, grad_out_planar=True)[0]
    monkeypatch.setenv("DLKA_CONV_BRICK", "0")
    n1 = lib.dlka_conv_brick_launch_count()
    g_wave = ops.conv3d_backward_cl(x, w, go, 1, 1, 1, grad_out_planar=True)[0]
    assert lib.dlka_conv_brick_launch_count() == n1
    assert (g_brick - g_wave).abs().max().item() <= 2e-5 * g_wave.abs().max().item()


def test_conv_brick_forward_stage0(monkeypatch):
    """cl_conv_brick3_kernel at the shape it exists for — the offset-predict conv's FORWARD at (32, 32^3), B = 2, planar output — against the fp64 conv at the forward
    contract (1e-4; check_conv3d_cl also holds the data / weight gradients, so the data gradient's brick kernel runs in the same call) and against
    cl_igemm_kernel<0,1,3,3> (same six products per term pair, another summation order); the launch counter proves which kernels ran."""
    from deformablelka_amd import _lib, ops
    lib = _lib.get_lib()
    n0 = lib.dlka_conv_brick_launch_count()
    parity.check_conv3d_cl(DEV, 2, 32, 81, (32, 32, 32), 3, 1, 1, 1, planar=True, seed=12)
    assert lib.dlka_conv_brick_launch_count() == n0 + 2
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(2, 32, 32, 32, 32, generator=gen).to(DEV)
    w = (torch.randn(81, 32, 3, 3, 3, generator=gen) * 0.05).to(DEV)
    bias = torch.randn(81, generator=gen).to(DEV)
    y_brick = ops.conv3d_forward_cl(x, w, bias, 1, 1, 1, out_planar=True)
    monkeypatch.setenv("DLKA_CONV_BRICK", "0")
    n1 = lib.dlka_conv_brick_launch_count()
    y_igemm = ops.conv3d_forward_cl(x, w, bias, 1, 1, 1, out_planar=True)
    assert lib.dlka_conv_brick_launch_count() == n1
    assert (y_brick - y_igemm).abs().max().item() <= 2e-6 * y_igemm.abs().max().item()


def test_stack_grad_input_fork_equals_one_stream(monkeypatch):
    """DLKABlockStack's data-chain pass runs the deformable conv's grad_input on the library's internal stream beside grad_offset (fork / join inside the block, also under
    hipGraph capture).  Same kernels, same saved activations — ONE forward pass, so that no sampling cell can move between the runs —: every gradient of a three-stage stack
    equals the one-stream backward pass (DLKA_GX_FORK_MIN_ROWS=huge) up to the order of the few fp32 global atomics (far samples), eager and replayed from a graph."""
    from deformablelka_amd.stack import DLKABlockStack
    stages = ((32, (8, 8, 8), 3), (64, (4, 4, 4), 2), (128, (4, 4, 4), 2))
    st = DLKABlockStack(2, stages=stages, device="cuda:0", seed=5)
    for _ in range(3):   # the first pass records the fold plan; the later ones run it sealed
        st.forward_backward()
    st.forward()
    torch.cuda.synchronize()
    res = {}

    def grads():
        return [g.clone() for b in st.blocks for g in b.grads] + [b.gx.clone() for b in st.blocks]

    for mode, val in (("one", "1000000000"), ("fork", "0")):
        monkeypatch.setenv("DLKA_GX_FORK_MIN_ROWS", val)
        _refresh_env()   # (the fork switches are read once per process)
        st.backward()
        torch.cuda.synchronize()
        res[mode] = grads()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            st.backward()
        g.replay()
        torch.cuda.synchronize()
        res[mode + "_graph"] = grads()
    for mode in ("fork", "one_graph", "fork_graph"):
        for a_, c_ in zip(res["one"], res[mode]):
            assert torch.isfinite(c_).all()
            scale = max(float(a_.abs().max()), 1e-6)
            assert float((a_ - c_).abs().max()) <= 2e-3 * scale, mode   # atomics order only


@pytest.mark.parametrize("C,hw", [(96, 56), (192, 28)])
def test_lka2d_backward_forks_equal_one_stream(C, hw, monkeypatch):
    """The 2-D block's backward with its two internal streams (each depthwise deformable conv's grad_input beside its grad_offset kernel; the offset nets' weight gradients
    beside the data chain) against the same pass on one stream (DLKA_LKA2D_FORK=0): every gradient equal up to atomics order — at a shape of each grad_input generation."""
    import deformablelka_amd as dk
    from oracle import blocks
    torch.manual_seed(3)
    m = dk.deformable_LKA_Attention(C)
    blocks.randomize_offsets_(m, std=0.03)
    m = m.to(DEV)
    x = torch.randn(4, C, hw, hw, device=DEV, requires_grad=True)
    gy = torch.randn(4, C, hw, hw, device=DEV)
    y = m(x)   # ONE forward pass: the three backward passes below share its saved offsets (a second forward could flip a sampling cell through atomics order)
    leaves = [x] + list(m.parameters())
    res = {}
    for mode in ("0", "1", None):
        if mode is None:
            monkeypatch.delenv("DLKA_LKA2D_FORK", raising=False)
        else:
            monkeypatch.setenv("DLKA_LKA2D_FORK", mode)
        _refresh_env()   # (the fork switches are read once per process)
        res[mode] = [g.clone() for g in torch.autograd.grad(y, leaves, gy, retain_graph=True)]
        torch.cuda.synchronize()
    for mode in ("1", None):
        for a_, c_ in zip(res["0"], res[mode]):
            assert torch.isfinite(c_).all()
            scale = max(float(a_.abs().max()), 1e-6)
            assert float((a_ - c_).abs().max()) <= 2e-3 * scale, mode


# ---- fork contexts: per device, per caller (include/dlka.h) --------------------------------------------------------------------
def _fork_stats(dev=0):
    import ctypes
    from deformablelka_amd import _lib
    c, l = ctypes.c_int64(0), ctypes.c_int64(0)
    assert _lib.get_lib().dlka_fork_stats(dev, ctypes.byref(c), ctypes.byref(l)) == 0
    return c.value, l.value


def _close(a_, c_, bitwise):
    if bitwise:
        return torch.equal(a_, c_)
    scale = max(float(a_.abs().max()), 1e-6)
    return float((a_ - c_).abs().max()) <= 2e-3 * scale


def test_fork_contexts_two_threads_two_streams(monkeypatch):
    """Two host threads, each on its own torch.cuda.Stream, inside the library AT THE SAME TIME — one in the 2-D block's backward (two internal streams), the other in the
    3-D token block's backward with grad_input forked beside grad_offset — reproduce what each pass gives alone: bit for bit wherever the pass is bitwise reproducible
    alone (everything that involves no floating-point global atomics), to atomics order elsewhere.  The calls go straight through the ops layer (ctypes releases the GIL
    for the duration of a C call; autograd would serialise both on the device's one backward thread).  Every call leases its own context from the device's pool."""
    import threading
    import deformablelka_amd as dk
    from deformablelka_amd import ops
    from oracle import blocks
    monkeypatch.setenv("DLKA_GX_FORK_MIN_ROWS", "0")   # the one-call 3-D backward forks too
    monkeypatch.delenv("DLKA_LKA2D_FORK", raising=False)
    _refresh_env()
    torch.manual_seed(11)
    # problem A: the 2-D block at the shape of its tile grad_input kernel
    m2 = dk.deformable_LKA_Attention(96)
    blocks.randomize_offsets_(m2, std=0.03)
    m2 = m2.to(DEV)
    params2 = [p.detach() for p in m2.block_params()]
    x2 = torch.randn(4, 96, 56, 56, device=DEV)
    g2 = torch.randn(4, 96, 56, 56, device=DEV)
    # problem B: the 3-D token block, stage-1 shape
    m3 = dk.LKA_Attention3d_deform(64)
    blocks.randomize_offsets_(m3, std=0.05)
    m3 = m3.to(DEV)
    params3 = [p.detach() for p in m3.block_params()]
    x3 = torch.randn(2, 16 * 16 * 16, 64, device=DEV)
    g3 = torch.randn(2, 16 * 16 * 16, 64, device=DEV)
    y2, saved2 = ops.lka2d_attention_forward(x2, params2)
    y3, saved3 = ops.lka3d_attention_tokens_forward(x3, params3, (16, 16, 16))
    torch.cuda.synchronize()

    def run2():
        gx, gs = ops.lka2d_attention_backward(x2, params2, g2, saved2)
        return [gx] + list(gs)

    def run3():
        gx, gs = ops.lka3d_attention_tokens_backward(x3, params3, g3, saved3, (16, 16, 16))
        return [gx] + list(gs)

    # alone, twice: the reference results and which tensors are bitwise reproducible
    ref = {}
    for name, fn in (("2d", run2), ("3d", run3)):
        a = [t.clone() for t in fn()]
        torch.cuda.synchronize()
        b = [t.clone() for t in fn()]
        torch.cuda.synchronize()
        ref[name] = (a, [torch.equal(u, v) for u, v in zip(a, b)])
        assert all(torch.isfinite(t).all() for t in a)
    c0, l0 = _fork_stats(0)
    iters = 24
    bar = threading.Barrier(2)
    out = {"2d": [], "3d": []}
    err = []

    def worker(name, fn):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(iters):
                    bar.wait()
                    out[name].append([t.clone() for t in fn()])
                s.synchronize()
        except Exception as e:   # noqa: BLE001
            err.append(repr(e))
            bar.abort()

    ts = [threading.Thread(target=worker, args=("2d", run2)), threading.Thread(target=worker, args=("3d", run3))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not err, err
    for name in ("2d", "3d"):
        a, bitwise = ref[name]
        assert len(out[name]) == iters
        for it, res in enumerate(out[name]):
            for k, (u, v) in enumerate(zip(a, res)):
                assert _close(u, v, bitwise[k]), (name, it, k, bitwise[k], float((u - v).abs().max()))
    c1, l1 = _fork_stats(0)
    assert l1 - l0 == 2 * iters          # every call forked
    assert c1 >= 2                        # ... and two calls that overlapped held two different contexts


def test_fork_contexts_follow_the_device():
    """A backward call issued with device 1 current (an nn.DataParallel replica: 2D/trainer_MaxViT_deform_LKA.py:107-108) takes its streams and events from device 1's
    pool — device 0's counters do not move — and computes what device 0 computes."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the 1-GPU box cannot exercise it; the 8-GPU node runs it)")
    import deformablelka_amd as dk
    from deformablelka_amd import ops
    from oracle import blocks
    torch.manual_seed(12)
    m2 = dk.deformable_LKA_Attention(96)
    blocks.randomize_offsets_(m2, std=0.03)
    x = torch.randn(2, 96, 56, 56)
    g = torch.randn(2, 96, 56, 56)
    res = {}
    for d in (0, 1):
        with torch.cuda.device(d):
            dev = f"cuda:{d}"
            md = m2.to(dev)
            params = [p.detach() for p in md.block_params()]
            before = (_fork_stats(0), _fork_stats(1))
            y, saved = ops.lka2d_attention_forward(x.to(dev), params)
            gx, gs = ops.lka2d_attention_backward(x.to(dev), params, g.to(dev), saved)
            torch.cuda.synchronize()
            after = (_fork_stats(0), _fork_stats(1))
            assert after[d][1] == before[d][1] + 1 and after[1 - d] == before[1 - d], (d, before, after)
            res[d] = [t.cpu() for t in [y, gx] + list(gs)]
    for u, v in zip(res[0], res[1]):
        assert _close(u, v, False)
    # a tensor of another device than the current one never reaches a launch
    with torch.cuda.device(0):
        with pytest.raises(RuntimeError, match="current device"):
            ops.lka2d_attention_forward(x.to("cuda:1"), [p.detach() for p in m2.to("cuda:1").block_params()])


@pytest.mark.parametrize("C,dims,bf", [(32, (32, 32, 32), False), (64, (16, 16, 16), True), (128, (8, 8, 8), False), (256, (4, 4, 4), False)])
def test_tblock3d_phased_backward_equals_one_call(C, dims, bf):
    parity.check_tblock3d_phased_backward(DEV, 2, C, dims, lka_bf16=bf)


def test_tblock3d_wgrad_overlap_equals_one_stream():
    """``module.wgrad_overlap = True`` (transformerblock.WgradOverlap): a chain of three wrapper blocks whose weight gradients run on the side stream and are joined once,
    at the end of backward(), against the same backward pass on one stream (``WgradOverlap.disabled``) — ONE forward pass (a second one could flip a sampling cell
    through atomics order), every gradient equal up to atomics order, eager and replayed from a hipGraph; and a block whose parameters already carry a .grad
    (accumulation) takes the one-stream pass."""
    import deformablelka_amd as dk
    from deformablelka_amd.transformerblock import WgradOverlap
    from oracle import blocks
    torch.manual_seed(5)
    C, (H, W, D) = 64, (16, 16, 16)
    mods = []
    for _ in range(3):
        m = dk.TransformerBlock_3D_single_deform_LKA(H * W * D, C, C, 4, dropout_rate=0.1, pos_embed=True)
        blocks.randomize_offsets_(m, std=0.3)
        with torch.no_grad():
            m.gamma.normal_(0.5, 0.2)
        m.keep_channels_last = True
        m.wgrad_overlap = True
        m._draw_drop_mask = lambda B_, C_, dtype, device: torch.ones(B_, C_, dtype=dtype, device=device)   # (the same mask in every run)
        mods.append(m.to(DEV).train())
    x = torch.randn(2, H, W, D, C, device=DEV).permute(0, 4, 1, 2, 3).requires_grad_(True)
    gy = torch.randn(2, H, W, D, C, device=DEV).permute(0, 4, 1, 2, 3)
    params = [p for m in mods for p in m.parameters()]
    names = ["x"] + [f"m{i}.{k}" for i, m in enumerate(mods) for k, _ in m.named_parameters()]
    ov = WgradOverlap.get(torch.device(DEV))
    y = x
    for m in mods:
        y = m(y)

    def backward(disabled):
        WgradOverlap.disabled = disabled
        try:
            for p in params + [x]:
                p.grad = None
            y.backward(gy, retain_graph=True)
            torch.cuda.synchronize()
        finally:
            WgradOverlap.disabled = False
        return [x.grad.clone()] + [p.grad.clone() for p in params]

    def same(tag, ref, got):
        bad = [(n, float((a_ - b_).abs().max()) / max(float(a_.abs().max()), 1e-30)) for n, a_, b_ in zip(names, ref, got)
               if not bool(torch.isfinite(b_).all()) or float((a_ - b_).abs().max()) > 2e-3 * max(float(a_.abs().max()), 1e-30)]
        assert not bad, (tag, bad[:8])

    ref = backward(True)
    for rep in range(3):
        same(f"eager {rep}", ref, backward(False))
        assert not ov.pending   # joined, nothing kept alive
    # accumulation into existing .grad tensors: the one-stream pass (2 x the gradient afterwards)
    for p in params + [x]:
        p.grad = None
    y.backward(gy, retain_graph=True)
    y.backward(gy)
    torch.cuda.synchronize()
    k = names.index("m0.conv8.1.weight")
    assert float((params[k - 1].grad - 2 * ref[k]).abs().max()) <= 4e-3 * max(float(ref[k].abs().max()), 1e-6)
    del y   # (capturing a backward pass while an eager autograd graph of the same modules is still retained crashes inside run_backward on this torch / ROCm build,
    #          with or without the side stream: scripts/debug_overlap3.py, profiles/r08_notes.md)
    # replayed from a graph (forward + backward captured: the join is the last node of the backward pass)
    for p in params + [x]:
        p.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        yy = x
        for m in mods:
            yy = m(yy)
        yy.backward(gy)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    got = [x.grad.clone()] + [p.grad.clone() for p in params]
    # (its forward pass is another one: the cells may differ from `ref`'s in a sample or two — measured 2.2e-2 on conv_offset.weight.grad of the middle block; 5e-2 for the gradients that collect grad_offset)
    for n, a_, b_ in zip(names, ref, got):
        lim = (5e-2 if any(t in n for t in ("conv0", "conv_spatial", "conv_offset", "proj_1", "norm.", "pos_embed", "x")) else 2e-3) * max(float(a_.abs().max()), 1e-30)
        assert bool(torch.isfinite(b_).all()) and float((a_ - b_).abs().max()) <= lim, ("graph", n, float((a_ - b_).abs().max()), lim)


@pytest.mark.parametrize("C,dims,dtype", [(32, (32, 32, 32), torch.float32), (128, (8, 8, 8), torch.float32), (64, (16, 16, 16), torch.bfloat16)])
def test_lka3d_tokens_phased_backward_equals_one_call(C, dims, dtype):
    parity.check_lka3d_tokens_phased_backward(DEV, 2, C, dims, dtype)


def test_lka3d_module_wgrad_overlap_equals_one_stream():
    """``LKA_Attention3d_deform.wgrad_overlap = True``: a chain of three BARE D-LKA modules through autograd with their weight gradients on the side stream (one join at
    the end of backward(): transformerblock.WgradOverlap) against the same backward pass on one stream — one forward pass, every gradient equal up to atomics order."""
    import deformablelka_amd as dk
    from deformablelka_amd.transformerblock import WgradOverlap
    from oracle import blocks
    torch.manual_seed(3)
    C, (H, W, D) = 64, (16, 16, 16)
    mods = []
    for _ in range(3):
        m = dk.LKA_Attention3d_deform(C)
        blocks.randomize_offsets_(m, std=0.3)
        m.wgrad_overlap = True
        mods.append(m.to(DEV))
    x = torch.randn(2, H * W * D, C, device=DEV, requires_grad=True)
    gy = torch.randn(2, H * W * D, C, device=DEV)
    params = [p for m in mods for p in m.parameters()]
    y = x
    for m in mods:
        y = m(y, 2, C, H, W, D)

    def backward(disabled):
        WgradOverlap.disabled = disabled
        try:
            for p in params + [x]:
                p.grad = None
            y.backward(gy, retain_graph=True)
            torch.cuda.synchronize()
        finally:
            WgradOverlap.disabled = False
        return [x.grad.clone()] + [p.grad.clone() for p in params]

    ref = backward(True)
    ov = WgradOverlap.get(torch.device(DEV))
    for rep in range(3):
        got = backward(False)
        assert not ov.pending
        for a_, b_ in zip(ref, got):
            assert bool(torch.isfinite(b_).all()) and float((a_ - b_).abs().max()) <= 2e-3 * max(float(a_.abs().max()), 1e-30), rep


def test_tblock3d_wgrad_overlap_frozen_parameter_and_shared_block_take_one_stream():
    """the side-stream pass is only safe when autograd keeps every returned weight gradient until backward() returns.  A FROZEN parameter's gradient is
    dropped at once (its memory would be re-used by the main stream while the side stream still writes it) and a block applied TWICE in one graph has its two
    gradients added on the main stream before the side stream has finished: both must take the one-stream pass (`WgradOverlap.eligible`) — observed through
    `ops.tblock3d_backward`'s side_stream argument — and give the one-stream gradients."""
    import deformablelka_amd as dk
    from deformablelka_amd import ops
    from deformablelka_amd.transformerblock import WgradOverlap
    from oracle import blocks
    torch.manual_seed(7)
    C, (H, W, D) = 32, (16, 16, 16)
    mods = []
    for _ in range(2):
        m = dk.TransformerBlock_3D_single_deform_LKA(H * W * D, C, C, 4, dropout_rate=0.1, pos_embed=True)
        blocks.randomize_offsets_(m, std=0.3)
        with torch.no_grad():
            m.gamma.normal_(0.5, 0.2)
        m.keep_channels_last = True
        m.wgrad_overlap = True
        m._draw_drop_mask = lambda B_, C_, dtype, device: torch.ones(B_, C_, dtype=dtype, device=device)
        mods.append(m.to(DEV).train())
    x = torch.randn(2, H, W, D, C, device=DEV).permute(0, 4, 1, 2, 3).requires_grad_(True)
    gy = torch.randn(2, H, W, D, C, device=DEV).permute(0, 4, 1, 2, 3)
    used = []
    orig = ops.tblock3d_backward

    def spy(*a, **k):
        used.append(k.get("side_stream") is not None)
        return orig(*a, **k)

    def run(frozen, twice, disabled):
        used.clear()
        for m in mods:
            for p in m.parameters():
                p.requires_grad_(True)
                p.grad = None
        x.grad = None
        if frozen:
            mods[0].conv51.conv1.conv.weight.requires_grad_(False)
        WgradOverlap.disabled = disabled
        ops.tblock3d_backward = spy
        try:
            y = mods[1](mods[0](x))
            if twice:
                y = mods[1](y)
            # main-stream allocations right behind each block's backward would land in a dropped gradient's memory
            y.backward(gy)
            torch.cuda.synchronize()
        finally:
            ops.tblock3d_backward = orig
            WgradOverlap.disabled = False
        return list(used), [x.grad.clone()] + [None if p.grad is None else p.grad.clone() for m in mods for p in m.parameters()]

    for frozen, twice in ((True, False), (False, True)):
        side_ref, ref = run(frozen, twice, True)
        side, got = run(frozen, twice, False)
        assert not any(side_ref)
        # backward order: mods[1] (twice: both applications), then mods[0]
        if frozen:
            assert side == [True, False], side      # the block with the frozen parameter: one stream; the other one keeps the side stream
        else:
            assert side == [False, False, True], side   # the block applied twice: one stream both times
        for a_, b_ in zip(ref, got):
            assert (a_ is None) == (b_ is None)
            if a_ is not None:
                assert bool(torch.isfinite(b_).all()) and float((a_ - b_).abs().max()) <= 2e-3 * max(float(a_.abs().max()), 1e-30)
    for m in mods:
        for p in m.parameters():
            p.requires_grad_(True)


@pytest.mark.parametrize("case", [(2, 32, 81, (32, 32, 32), 3, 1, 1), (2, 128, 81, (8, 8, 8), 3, 1, 1), (24, 96, 98, (1, 56, 56), (1, 7, 7), (0, 9, 9), (1, 3, 3)),
                                  (24, 384, 50, (1, 14, 14), (1, 5, 5), (0, 2, 2), 1), (3, 64, 81, (5, 6, 7), 3, 1, 1)])
def test_wgrad_from_padded_copy_equals_unpadded(case):
    """The padded dense weight gradient at the real shapes (3-D offset conv at 32^3 and 8^3, the 2-D offset nets at 56^2 and 14^2) and a ragged one: bitwise equal to the
    unpadded kernels, 1e-3 of the fp64 conv."""
    parity.check_wgrad_pad_equals_unpadded(DEV, *case)


@pytest.mark.parametrize("C,dims,bf", [(128, (8, 8, 8), False), (256, (4, 4, 4), False), (128, (8, 8, 8), True), (256, (4, 4, 4), True), (64, (5, 3, 8), False),
                                       (32, (7, 9, 4), True)])
def test_dwpair_equals_unfused(C, dims, bf):
    """cl_dwpair.hip (round 5): the 8^3 / 4^3 stages' two depthwise convs (and their data gradients) in ONE launch == one launch per conv (DLKA_DWPAIR=0): the whole
    wrapper block's outputs and gradients, fp32 and DLKA_BF16 storage, at the real stage shapes and two ragged ones.  (That block is also held to the oracle by the
    tblock parity tests, which run through the fused kernel at these shapes.)"""
    parity.check_dwpair_equals_unfused(DEV, 2, C, dims, lka_bf16=bf)


@pytest.mark.parametrize("case", [(2, 32, (32, 32, 32), 7, 9, 3), (2, 32, (32, 32, 32), 5, 2, 1), (2, 64, (16, 16, 16), 7, 9, 3), (1, 32, (13, 7, 9), 7, 9, 3), (2, 64, (5, 6, 16), 5, 2, 1)])
def test_dwconv_pipelined_row_pairs_equal_row_kernel(case, monkeypatch):
    """cl_dwconv_rows2p_kernel (round 6, opt-in DLKA_DW_2P: software-pipelined rows, output rows on the halves of v_pk_fma_f32, two output planes) at the stage-0 / stage-1
    shapes and two ragged ones: bit for bit the row kernel's result (same FMA chain per output), forward and data gradient, and the fp64 conv's to 1e-4."""
    from deformablelka_amd import ops
    from deformablelka_amd._lib import get_lib
    lib = get_lib()
    B, C, dims, k, p, d = case
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(B, *dims, C, generator=gen).to(DEV)
    g = torch.randn(B, *dims, C, generator=gen).to(DEV)
    w = (torch.randn(C, 1, k, k, k, generator=gen) * 0.1).to(DEV)
    bias = torch.randn(C, generator=gen).to(DEV)
    res = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("DLKA_DW_2P", mode)
        n0 = lib.dlka_dwconv_2p_launch_count()
        y = ops.conv3d_forward_cl(x, w, bias, p, d, C)
        gi = ops.conv3d_backward_cl(x, w, g, p, d, C)[0]
        assert (lib.dlka_dwconv_2p_launch_count() - n0 >= 2) == (mode != "0"), (mode, n0, lib.dlka_dwconv_2p_launch_count())
        res[mode] = (y, gi)
    for mode in ("1", "2"):
        assert torch.equal(res["0"][0], res[mode][0]) and torch.equal(res["0"][1], res[mode][1]), mode
    ref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double().cpu(), w.double().cpu(), bias.double().cpu(), padding=p, dilation=d, groups=C).permute(0, 2, 3, 4, 1)
    assert float((res["1"][0].double().cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("bf", [False, True])
def test_weight_preparation_tiled_equals_elementwise(bf):
    """cl_igemm.hip prep_job_tile (round 5): the LDS-tiled weight re-layout is bitwise the element-per-lane one — every prepared form of one block of each Synapse width."""
    parity.check_prep_tiled_equals_elementwise(DEV, ((32, (8, 8, 8), 1), (64, (4, 4, 4), 1), (128, (4, 4, 4), 1), (256, (4, 4, 4), 1)), torch.bfloat16 if bf else torch.float32)
