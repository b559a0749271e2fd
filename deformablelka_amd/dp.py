"""Data-parallel plumbing for the D-LKA block stack: one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on
the MI355X node, ``gloo`` in the CPU tests).  The batch shards across ranks with no data-path collective; the only
exchange is the gradient all-reduce, issued either once per step or in two pieces so that the first overlaps the rest of the
backward pass.  Which of the two a step uses has to be THE SAME on every rank — mismatched collectives hang or corrupt — so every
local decision (did my graph capture work? did my trial step run?) is reduced over the ranks before anyone acts on it."""
from __future__ import annotations

from typing import Callable

import torch


def all_ranks_agree(ok: bool, dist, world: int, device) -> bool:
    """True iff ``ok`` holds on EVERY rank (MIN all-reduce of a flag).  A collective: all ranks must call it, in the same order."""
    if world == 1 or dist is None:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def choose_schedule(want_overlap: bool, prepare_overlap: Callable[[], bool], trial_overlap: Callable[[], bool], dist, world: int, device) -> str:
    """Returns "overlap" or "single", identically on all ranks.

    prepare_overlap: local set-up of the two-piece schedule (e.g. the split hipGraph capture) -> success flag.
    trial_overlap:   one trial of the two compute pieces WITHOUT any collective (a failure after an issued all-reduce would leave
                     it dangling on the peers) -> success flag.
    Both are only called when every rank is still a candidate; their exceptions count as failure."""
    def safe(fn):
        try:
            return bool(fn())
        except Exception:
            return False
    if not all_ranks_agree(want_overlap, dist, world, device):
        return "single"
    if not all_ranks_agree(safe(prepare_overlap), dist, world, device):
        return "single"
    if not all_ranks_agree(safe(trial_overlap), dist, world, device):
        return "single"
    return "overlap"


def step_single(stack, lr: float, world: int, dist, compute: Callable[[], None]) -> None:
    compute()
    stack.reduce_and_update(lr, world, dist)


def step_overlap(stack, lr: float, world: int, dist, compute_a: Callable[[], None], compute_b: Callable[[], None], cut: int) -> None:
    """compute_a = forward + backward of blocks[split:], whose gradients (flat_grads[cut:]) are final afterwards; their all-reduce runs
    while compute_b (backward of blocks[:split]) computes; then the remaining piece, both waits, the SGD update."""
    compute_a()
    w1 = dist.all_reduce(stack.flat_grads[cut:], async_op=True) if world > 1 else None
    compute_b()
    w2 = dist.all_reduce(stack.flat_grads[:cut], async_op=True) if world > 1 else None
    if w1 is not None:
        w1.wait()
        w2.wait()
    stack.flat_params.add_(stack.flat_grads, alpha=-lr / world)
