"""N>1 logic of the data-parallel path on CPU: two `gloo` ranks, each with its own batch shard, run the block stack
through the emulator build of the real kernels, all-reduce the flat gradient buffer and take the SGD step
(`DLKABlockStack.reduce_and_update`, the code bench.py runs over RCCL).  The result must equal one process that sees
both shards."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

STAGES = ((32, (2, 2, 3), 2),)   # two chained C=32 blocks on a 2x2x3 volume (the emulator runs every lane as a fiber)
LR = 0.1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


STAGES2 = ((32, (2, 2, 3), 1), (64, (2, 2, 2), 1))   # two stages: the overlapped schedule cuts the backward pass between them


def _make_stack(data_seed, stages=STAGES):
    from deformablelka_amd import _lib
    from deformablelka_amd.stack import DLKABlockStack
    from tests import emu
    _lib._set_backend_for_tests(emu.load())
    st = DLKABlockStack(1, stages=stages, device="cpu", seed=7)            # same parameters on every rank
    g = torch.Generator().manual_seed(1000 + data_seed)                    # rank-specific inputs / grad_outputs
    for chain in st.chains:
        chain[0].x.copy_(torch.randn(chain[0].x.shape, generator=g))
        chain[-1].gy.copy_(torch.randn(chain[-1].gy.shape, generator=g))
    return st


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIPEMU_THREADS="2")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = _make_stack(rank)
    st.forward_backward()
    st.reduce_and_update(LR, world, dist)
    if rank == 0:
        torch.save({"params": st.flat_params.clone(), "grads": st.flat_grads.clone()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_data_parallel_step_matches_single_process(tmp_path):
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    # single process: both shards, gradients summed by hand
    gsum, p0 = None, None
    for r in range(2):
        st = _make_stack(r)
        p0 = st.flat_params.clone()
        st.forward_backward()
        gsum = st.flat_grads.clone() if gsum is None else gsum + st.flat_grads
    from deformablelka_amd import _lib
    _lib._set_backend_for_tests(None)
    assert gsum.abs().max() > 0
    assert torch.allclose(got["grads"], gsum, rtol=1e-5, atol=1e-6)
    assert torch.allclose(got["params"], p0 - LR / 2 * gsum, rtol=1e-5, atol=1e-6)


def test_single_rank_update_needs_no_process_group():
    st = _make_stack(0)
    p0 = st.flat_params.clone()
    st.forward_backward()
    st.reduce_and_update(LR, 1, None)
    from deformablelka_amd import _lib
    _lib._set_backend_for_tests(None)
    assert torch.allclose(st.flat_params, p0 - LR * st.flat_grads)


def _worker_overlapped(rank, world, port, out):
    """The schedule bench.py runs for N > 1: backward of the late (parameter-heavy) blocks, their all-reduce asynchronously, the rest of
    the backward pass, the second all-reduce, both waits, then the update."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIPEMU_THREADS="2")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = _make_stack(rank, STAGES2)
    split = st.split_index(0.5)   # the C = 64 block holds 72 % of the gradient bytes of this two-stage stack
    cut = st.grad_offset_of(split)
    assert 0 < split < len(st.blocks) and 0 < cut < st.flat_grads.numel()
    st.forward()
    st.backward(split, None)
    w1 = dist.all_reduce(st.flat_grads[cut:], async_op=True)
    st.backward(0, split)
    w2 = dist.all_reduce(st.flat_grads[:cut], async_op=True)
    w1.wait()
    w2.wait()
    st.flat_params.add_(st.flat_grads, alpha=-LR / world)
    if rank == 0:
        torch.save({"params": st.flat_params.clone(), "grads": st.flat_grads.clone()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_overlapped_allreduce_schedule_matches_single_process(tmp_path):
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker_overlapped, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    gsum, p0 = None, None
    for r in range(2):
        st = _make_stack(r, STAGES2)
        p0 = st.flat_params.clone()
        st.forward_backward()
        gsum = st.flat_grads.clone() if gsum is None else gsum + st.flat_grads
    from deformablelka_amd import _lib
    _lib._set_backend_for_tests(None)
    assert torch.allclose(got["grads"], gsum, rtol=1e-5, atol=1e-6)
    assert torch.allclose(got["params"], p0 - LR / 2 * gsum, rtol=1e-5, atol=1e-6)


def _worker_schedule(rank, world, port, out, failing_rank, fail_at):
    """Unhappy path of the schedule choice: one rank's split capture (or trial step) fails -> EVERY rank must fall back to the single
    all-reduce; the step then still produces the single-process result (mismatched collectives would hang or corrupt instead)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIPEMU_THREADS="2")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deformablelka_amd import dp
    st = _make_stack(rank, STAGES2)
    split = st.split_index(0.5)
    cut = st.grad_offset_of(split)

    def prepare():
        if rank == failing_rank and fail_at == "prepare":
            raise RuntimeError("simulated capture failure")
        return True

    def trial():
        if rank == failing_rank and fail_at == "trial":
            return False
        st.forward(); st.backward(split, None); st.backward(0, split)
        return True

    sched = dp.choose_schedule(True, prepare, trial, dist, world, "cpu")
    if sched == "overlap":
        dp.step_overlap(st, LR, world, dist, lambda: (st.forward(), st.backward(split, None)), lambda: st.backward(0, split), cut)
    else:
        dp.step_single(st, LR, world, dist, st.forward_backward)
    torch.save({"params": st.flat_params.clone(), "grads": st.flat_grads.clone(), "sched": sched}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("failing_rank,fail_at,expect", [(1, "prepare", "single"), (0, "trial", "single"), (-1, "", "overlap")])
def test_all_ranks_take_the_same_allreduce_schedule(tmp_path, failing_rank, fail_at, expect):
    out = str(tmp_path / "rank")
    mp.spawn(_worker_schedule, args=(2, _free_port(), out, failing_rank, fail_at), nprocs=2, join=True)
    got = [torch.load(out + f".{r}") for r in range(2)]
    assert got[0]["sched"] == got[1]["sched"] == expect
    gsum, p0 = None, None
    for r in range(2):
        st = _make_stack(r, STAGES2)
        p0 = st.flat_params.clone()
        st.forward_backward()
        gsum = st.flat_grads.clone() if gsum is None else gsum + st.flat_grads
    from deformablelka_amd import _lib
    _lib._set_backend_for_tests(None)
    for g in got:
        assert torch.allclose(g["grads"], gsum, rtol=1e-5, atol=1e-6)
        assert torch.allclose(g["params"], p0 - LR / 2 * gsum, rtol=1e-5, atol=1e-6)
