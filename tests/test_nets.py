"""SURVEY.md §8f rows 2-4 on the CPU: the full 3-D net assembly against the reference's own class (keys / shapes of the real assembly, the
plumbing around the D-LKA blocks at the full 64x128x128 patch), the 2-D decoder pieces against the reference's classes (kernel sources on
the emulator), the sliding-window predictor, and the data-parallel trainer hook (2 gloo ranks)."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import golden_checks

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


# ---- D_LKA_Former ---------------------------------------------------------------------------------------------------------------------
def test_d_lka_former_state_dict_matches_the_reference_class():
    """Every key and shape of the reference's D_LKA_Former(trans_block=TransformerBlock_3D_single_deform_LKA) — 699 entries, 42.35 M
    parameters — so that a reference checkpoint loads with strict=True."""
    import deformablelka_amd as dk
    g = golden_checks.gold_nets()
    net = dk.D_LKA_Former(in_channels=1, out_channels=14, img_size=[64, 128, 128], feature_size=16, num_heads=4, depths=[3, 3, 3, 3],
                          dims=[32, 64, 128, 256], do_ds=True)
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert mine == g["D_LKA_Former_keys"]
    assert sum(p.numel() for p in net.parameters()) == g["D_LKA_Former_params"] == 42353327
    assert len(net.dlka_blocks()) == 21


def test_d_lka_former_plumbing_matches_the_reference_at_full_size():
    """Everything AROUND the transformer blocks — stem, down-sampling + GroupNorm, token reshapes, transposed convs, skip additions,
    encoder1 / decoder2 (InstanceNorm conv blocks), deep-supervision heads — against the reference's class at the full patch, with the same
    cheap block plugged into both (``trans_block`` is a constructor argument of both)."""
    import deformablelka_amd as dk
    from make_golden_nets import LiteBlock
    case = golden_checks.gold_nets()["D_LKA_Former_plumbing"]
    net = dk.D_LKA_Former(trans_block=LiteBlock, **case["ctor"])
    net.load_state_dict(case["state_dict"], strict=True)
    net.eval()
    x = torch.randn(1, 1, 64, 128, 128, generator=torch.Generator().manual_seed(case["input_seed"]))
    with torch.no_grad():
        outs = net(x)
    assert [tuple(o.shape) for o in outs] == case["out_shapes"]
    for o, sub, mean, amean in zip(outs, case["out_sub"], case["out_mean"], case["out_abs_mean"]):
        assert (o[..., ::8, ::8, ::8] - sub).abs().max().item() < 2e-5
        assert abs(float(o.double().mean()) - mean) < 1e-6 and abs(float(o.double().abs().mean()) - amean) < 1e-6


# ---- 2-D decoder ----------------------------------------------------------------------------------------------------------------------
@pytest.fixture()
def emu_backend():
    from deformablelka_amd import _lib
    from tests import emu
    _lib._set_backend_for_tests(emu.load())
    yield
    _lib._set_backend_for_tests(None)


def test_assembled_net_with_real_dlka_blocks_vs_oracle_assembled_net(emu_backend):
    """D_LKA_Former(trans_block=TransformerBlock_3D_single_deform_LKA) — the 21 real D-LKA blocks, kernel sources on the emulator — against the
    oracle-assembled net (tests/netoracle.py: same assembly, every block = oracle.blocks.transformer_block_3d) on the smallest image the Synapse stem
    admits (16x32x32 -> stages 8^3 / 4^3 / 2^3 / 1^3): logits of all three heads, argmax of the full-resolution head, the deep-supervision loss and
    EVERY parameter gradient (d_lka_former_synapse.py:144-167, model_components.py:52-66).  Evaluation-mode normalisation: at this size the
    deepest stage holds ONE voxel per sample, and batch statistics over two values are not a well-conditioned comparison (train mode runs at
    32x64x64 on the GPU, tests/test_nets_gpu.py)."""
    from tests import netoracle
    res = netoracle.run_pair("cpu", (16, 32, 32), B=1, training=False)
    s = netoracle.summarize(res)
    print({k: v for k, v in s.items() if not k.endswith("grad_errs")})
    for tag in ("ref", "same"):
        assert max(s[tag + "_logit_abs"]) <= 1e-4, s[tag + "_logit_abs"]
        assert s[tag + "_argmax_agree"] >= 0.999
        assert s[tag + "_loss_abs"] <= 1e-5
    assert len(s["same_grad_errs"]) > 500   # (every parameter of the net with a non-zero gradient: 21 blocks x 26 + the plumbing)
    lim = 1e-3 if s["flipped"] == 0 else 8e-3
    assert all(v <= lim for v in s["ref_grad_errs"].values()), s["ref_grad_worst"]
    assert all(v <= 1e-3 for v in s["same_grad_errs"].values()), s["same_grad_worst"]   # identical sampling cells: the contract's 1e-3, no exception


@pytest.mark.parametrize("name", ["deformableLKABlock", "MyDecoderLayer", "MyDecoderLayer_last", "MyDecoderLayer_noskip"])
def test_decoder2d_golden_on_emulator(name, emu_backend):
    golden_checks.replay(name, "cpu")


# ---- sliding-window inference -----------------------------------------------------------------------------------------------------------
def test_sliding_window_steps():
    from deformablelka_amd.inference import compute_steps_for_sliding_window as steps, num_tiles
    assert steps((64,), (110,), 0.5) == [[0, 23, 46]]                  # the example in neural_network.py:270-271
    assert steps((64, 128, 128), (64, 128, 128), 0.5) == [[0], [0], [0]]
    assert steps((96, 96, 96), (240, 240, 160), 0.5) == [[0, 48, 96, 144], [0, 48, 96, 144], [0, 32, 64]]
    assert num_tiles((240, 240, 160), (96, 96, 96), 16, 16) == 10 * 10 * 5   # SURVEY §8d cfg 5


def test_gaussian_importance_map_matches_scipy():
    """neural_network.py:250-263 uses scipy.ndimage.gaussian_filter on a unit impulse; the separable closed form must agree."""
    from scipy.ndimage import gaussian_filter
    from deformablelka_amd.inference import gaussian_importance_map
    for ps in [(8, 12, 10), (16, 16, 16), (5, 24, 24)]:
        tmp = np.zeros(ps)
        tmp[tuple(i // 2 for i in ps)] = 1
        ref = gaussian_filter(tmp, [i / 8 for i in ps], 0, mode="constant", cval=0)
        ref = (ref / ref.max()).astype(np.float32)
        ref[ref == 0] = ref[ref != 0].min()
        got = gaussian_importance_map(ps).numpy()
        assert np.abs(got - ref).max() < 1e-6 and got.min() > 0 and got.max() == 1.0


class _PointwiseNet(torch.nn.Module):
    """Per-voxel network: its tiled prediction must equal its whole-volume prediction whatever the blending weights are."""

    def __init__(self, k=3):
        super().__init__()
        self.c = torch.nn.Conv3d(1, k, 1)

    def forward(self, x):
        return [self.c(x), self.c(x)[..., ::2, ::2, ::2]]   # deep-supervision list: the predictor takes the first head


def test_tiled_prediction_equals_whole_volume_for_a_pointwise_net():
    from deformablelka_amd import inference as inf
    torch.manual_seed(0)
    net = _PointwiseNet()
    x = torch.randn(1, 21, 30, 19)
    whole = torch.softmax(net(x[None])[0], 1)[0]
    for gauss in (True, False):
        seg, probs = inf.predict_3d_tiled(net, x, (8, 16, 8), step_size=0.5, use_gaussian=gauss, tile_batch=3)
        assert probs.shape == whole.shape and (probs - whole).abs().max().item() < 1e-5
        assert torch.equal(seg, whole.argmax(0))
    # smaller than the patch in one axis: symmetric zero padding, cropped again
    seg, probs = inf.predict_3d_tiled(net, x[:, :5], (8, 16, 8), step_size=0.5)
    assert probs.shape == (3, 5, 30, 19) and (probs - torch.softmax(net(x[None, :, :5])[0], 1)[0]).abs().max().item() < 1e-5
    # pancreas procedure (test_util.py:45-111)
    lab, score = inf.predict_single_case(net, x[0], stride_xy=6, stride_z=5, patch_size=(8, 16, 8), num_classes=3, tile_batch=2)
    assert (score - whole).abs().max().item() < 1e-5 and torch.equal(lab, whole.argmax(0))


def test_predict_single_case_matches_the_reference_loop():
    """A network whose output depends on the tile CONTENT AND POSITION-in-tile (3^3 conv with zero padding): the restated numpy loop of
    test_util.py:73-106 must give the same blended score map."""
    from deformablelka_amd import inference as inf
    torch.manual_seed(1)
    net = torch.nn.Conv3d(1, 2, 3, padding=1)
    image = torch.randn(13, 17, 11)
    ps, sxy, sz = (8, 8, 8), 4, 3
    lab, score = inf.predict_single_case(net, image, sxy, sz, ps, num_classes=2, tile_batch=5)
    img = image.numpy()
    ww, hh, dd = img.shape
    nx, ny, nz = math.ceil((ww - ps[0]) / sxy) + 1, math.ceil((hh - ps[1]) / sxy) + 1, math.ceil((dd - ps[2]) / sz) + 1
    sm, cnt = np.zeros((2,) + img.shape, np.float32), np.zeros(img.shape, np.float32)
    for a in range(nx):
        xs = min(sxy * a, ww - ps[0])
        for b in range(ny):
            ys = min(sxy * b, hh - ps[1])
            for c in range(nz):
                zs = min(sz * c, dd - ps[2])
                t = torch.from_numpy(img[xs:xs + 8, ys:ys + 8, zs:zs + 8])[None, None]
                y = torch.softmax(net(t), 1)[0].detach().numpy()
                sm[:, xs:xs + 8, ys:ys + 8, zs:zs + 8] += y
                cnt[xs:xs + 8, ys:ys + 8, zs:zs + 8] += 1
    sm = sm / cnt[None]
    assert np.abs(score.numpy() - sm).max() < 1e-5 and np.array_equal(lab.numpy(), sm.argmax(0))


# ---- trainer hook: DistributedDataParallel over two gloo ranks ------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_block():
    import deformablelka_amd as dk
    from oracle.blocks import randomize_offsets_
    torch.manual_seed(5)
    m = dk.deformableLKABlock(dim=8)
    randomize_offsets_(m, std=0.05)
    return m


def _ddp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIPEMU_THREADS="2")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deformablelka_amd import _lib, training
    from tests import emu
    _lib._set_backend_for_tests(emu.load())
    net = training.wrap_data_parallel(_make_block(), "cpu")
    assert isinstance(net, torch.nn.parallel.DistributedDataParallel)
    g = torch.Generator().manual_seed(100 + rank)                      # this rank's shard of the batch
    x, tgt = torch.randn(1, 20, 8, generator=g), torch.randn(1, 20, 8, generator=g)
    opt = torch.optim.SGD(net.parameters(), 0.1)
    loss = training.run_iteration(net, opt, x, tgt, loss_fn=lambda o, t: ((o - t) ** 2).mean(), clip_norm=1e9, forward=lambda t: net(t, 4, 5))
    if rank == 0:
        torch.save({"params": {k: v.detach().clone() for k, v in net.module.state_dict().items()}, "loss": loss}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_ddp_trainer_hook_two_ranks_equals_one_process_on_both_shards(tmp_path, emu_backend):
    """wrap_data_parallel + run_iteration on 2 gloo ranks (1 sample each) == one process that sees both samples (mean loss)."""
    from deformablelka_amd import training
    out = str(tmp_path / "r0.pt")
    mp.spawn(_ddp_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    from deformablelka_amd import _lib
    from tests import emu
    _lib._set_backend_for_tests(emu.load())
    net = _make_block()
    xs, ts = [], []
    for r in range(2):
        g = torch.Generator().manual_seed(100 + r)
        xs.append(torch.randn(1, 20, 8, generator=g)); ts.append(torch.randn(1, 20, 8, generator=g))
    opt = torch.optim.SGD(net.parameters(), 0.1)
    training.run_iteration(net, opt, torch.cat(xs), torch.cat(ts), loss_fn=lambda o, t: ((o - t) ** 2).mean(), clip_norm=1e9, forward=lambda t: net(t, 4, 5))
    for k, v in net.state_dict().items():
        assert torch.allclose(got["params"][k], v, rtol=1e-4, atol=1e-6), k


@pytest.mark.parametrize("momentum", [0.1, 0.3, None])
def test_bn_running_statistics_bookkeeping_equals_torchs(momentum):
    """dynunet_block.bn_update_running (the wrapper block's two norms in one foreach launch) against nn.BatchNorm3d's own training-mode bookkeeping
    (running mean / unbiased variance / step counter), single and grouped call forms, over three steps."""
    import torch.nn as nn
    from deformablelka_amd.dynunet_block import bn_update_running
    torch.manual_seed(0)
    mine = [nn.BatchNorm3d(6, momentum=momentum) for _ in range(3)]
    ref = [nn.BatchNorm3d(6, momentum=momentum).train() for _ in range(3)]
    for step in range(3):
        xs = [torch.randn(3, 6, 2, 3, 4) * (1 + k) + step for k in range(3)]
        stats = []
        for k, x in enumerate(xs):
            ref[k](x)
            stats.append(torch.cat([x.mean((0, 2, 3, 4)), torch.zeros(6), x.var((0, 2, 3, 4), unbiased=True)]))
        bn_update_running(mine[0], stats[0])
        bn_update_running((mine[1], stats[1]), (mine[2], stats[2]))
    for a, b in zip(mine, ref):
        assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == 3
        assert torch.allclose(a.running_mean, b.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(a.running_var, b.running_var, rtol=1e-5, atol=1e-6)


def _drop_mask_equals_dropout3d(device):
    """TransformerBlock_3D_single_deform_LKA._draw_drop_mask (bernoulli_ + div_ on an empty tensor: two launches) makes the draw F.dropout3d makes for its
    (B, C, 1, 1, 1) noise tensor — same values AND the same generator state afterwards, so a net that mixes both stays in step with the reference's
    (transformerblock.py:598 conv8 = Dropout3d(0.1) + 1x1x1 conv)."""
    import torch.nn.functional as F
    import deformablelka_amd as dk
    m = dk.TransformerBlock_3D_single_deform_LKA(8, 32, 32, 4, dropout_rate=0.1, pos_embed=True)
    for seed, (B, C) in enumerate([(2, 32), (2, 256), (3, 64), (24, 128)]):
        torch.manual_seed(seed)
        ref = F.dropout3d(torch.ones(B, C, 1, 1, 1, device=device), 0.1, True).view(B, C)
        ref_next = torch.rand(4, device=device)
        torch.manual_seed(seed)
        mine = m._draw_drop_mask(B, C, torch.float32, torch.device(device))
        mine_next = torch.rand(4, device=device)
        assert torch.equal(ref, mine) and torch.equal(ref_next, mine_next), (seed, B, C)
        assert set(mine.unique().tolist()) <= {0.0, float(torch.tensor(1.0) / torch.tensor(0.9))}


def test_drop_mask_draw_equals_dropout3d():
    _drop_mask_equals_dropout3d("cpu")


def test_rows_matmul_chunked_weight_gradient():
    """network._RowsMatmul (the GEMM behind the kernel == stride convs: weight gradient as a batched GEMM over 1024-row chunks + a sum) against plain matmul autograd, fp64:
    chunked sizes, a size below the threshold, a size that is not a multiple of the chunk."""
    from deformablelka_amd.network import _RowsMatmul
    torch.manual_seed(0)
    for M, K, N in ((16384, 32, 48), (20480, 8, 16), (4096, 32, 64), (16384 + 512, 16, 8)):
        a = torch.randn(M, K, dtype=torch.float64, requires_grad=True)
        w = torch.randn(K, N, dtype=torch.float64, requires_grad=True)
        gy = torch.randn(M, N, dtype=torch.float64)
        y = _RowsMatmul.apply(a, w)
        y.backward(gy)
        a2, w2 = a.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
        y2 = a2 @ w2
        y2.backward(gy)
        assert torch.equal(y, y2) and torch.equal(a.grad, a2.grad)
        assert (w.grad - w2.grad).abs().max().item() <= 1e-12 * w2.grad.abs().max().item()
    a = torch.randn(16384, 4, dtype=torch.float64)   # a frozen weight: no weight gradient is formed
    w = torch.randn(4, 4, dtype=torch.float64)
    a.requires_grad_(True)
    _RowsMatmul.apply(a, w).sum().backward()
    assert a.grad is not None


def test_eval_reference_checkpoint_helpers(tmp_path):
    """scripts/eval_reference_checkpoint.py (the "DSC vs ref" half of the metric, runnable once the published weights / data are mounted): an nnU-Net-style checkpoint
    file with DataParallel prefixes loads into D_LKA_Former with strict key agreement; Dice per class on a hand-made pair."""
    import importlib.util
    import numpy as np
    import deformablelka_amd as dk
    spec = importlib.util.spec_from_file_location("eval_ref", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "eval_reference_checkpoint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    net = dk.D_LKA_Former(in_channels=1, out_channels=14, img_size=[64, 128, 128], feature_size=16, num_heads=4, depths=[3, 3, 3, 3], dims=[32, 64, 128, 256], do_ds=True)
    f = str(tmp_path / "model_final_checkpoint.model")
    torch.save({"state_dict": {"module." + k: v for k, v in net.state_dict().items()}, "epoch": 1000}, f)
    sd = mod.load_reference_state_dict(f)
    assert set(sd) == set(net.state_dict()) and len(sd) == 699
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    pred = np.zeros((4, 4, 4), dtype=np.int64)
    tgt = np.zeros((4, 4, 4), dtype=np.int64)
    pred[:2] = 1; tgt[:2, :2] = 1          # |P| = 32, |T| = 16, |P & T| = 16 -> 2 * 16 / 48
    tgt[3] = 2                             # class 2 only in the target -> 0
    d = mod.dice_per_class(pred, tgt, [1, 2, 3])
    assert abs(d[1] - 2 * 16 / 48) < 1e-12 and d[2] == 0.0 and 3 not in d
