"""The trainer-side hooks of the 3-D net: what ``d_lka_former_trainer_synapse.py`` does around the network —
``initialize_network`` (:158-198), ``initialize_optimizer_and_scheduler`` (:200-205) and ``run_iteration`` (:259-309) — plus the
data-parallel wrapper the reference leaves to ``nn.DataParallel`` (``run_training.py``): one process per GPU, ``torch.distributed``
(backend ``nccl`` = RCCL over xGMI), gradients averaged with bucketed all-reduces that overlap the backward pass
(``DistributedDataParallel``), batch sharded by rank with no other collective."""
from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.nn as nn

from .network import D_LKA_Former
from .transformerblock import TransformerBlock_3D_single_deform_LKA


def initialize_network(input_channels: int = 1, num_classes: int = 14, crop_size: Sequence[int] = (64, 128, 128), depths=(3, 3, 3, 3),
                       skip_connections=(True, True, True, True), trans_block=TransformerBlock_3D_single_deform_LKA, device=None,
                       patch_size=(2, 4, 4), wgrad_overlap: bool = True) -> D_LKA_Former:
    """d_lka_former_trainer_synapse.py:158-185 (the fvcore FLOP count of :186-193 is logging only).
    wgrad_overlap: the D-LKA transformer blocks issue their weight gradients on a side stream joined at the end of ``backward()`` (transformerblock.WgradOverlap: the
    block-stack engine's schedule for the nn.Module path).  Right for this trainer's loop (``run_iteration``: zero_grad(set_to_none) -> backward -> clip -> step);
    ``wrap_data_parallel`` switches it off again, DistributedDataParallel reads gradients while the pass is still running."""
    net = D_LKA_Former(in_channels=input_channels, out_channels=num_classes, img_size=crop_size, feature_size=16, num_heads=4, depths=list(depths),
                       dims=[32, 64, 128, 256], do_ds=True, trans_block=trans_block, skip_connections=list(skip_connections), patch_size=patch_size)
    if device is not None:
        net = net.to(device)
    net.inference_apply_nonlin = lambda x: torch.softmax(x, 1)
    set_wgrad_overlap(net, wgrad_overlap)
    return net


def set_wgrad_overlap(net: nn.Module, on: bool) -> int:
    """Switch the side-stream weight-gradient schedule of every D-LKA transformer block of `net` (see transformerblock.WgradOverlap for its contract); returns how many."""
    n = 0
    for m in net.modules():
        if hasattr(type(m), "wgrad_overlap"):
            m.wgrad_overlap = bool(on)
            n += 1
    return n


def initialize_optimizer(net: nn.Module, initial_lr: float = 1e-2, weight_decay: float = 3e-5, fused=None):
    """:200-205: SGD, momentum 0.99, Nesterov; the reference drives the learning rate with nnU-Net's poly schedule.
    ``fused`` (default: on when every parameter is a floating-point GPU tensor) selects torch.optim.SGD's single-pass implementation — the same update,
    weight decay, momentum, Nesterov step and parameter write in one kernel per tensor list instead of the four ``_foreach`` passes over the 42 M parameters
    (0.9 ms of a 30 ms iteration)."""
    params = list(net.parameters())
    if fused is None:
        fused = bool(params) and all(p.is_cuda and p.is_floating_point() for p in params)
    kw = dict(weight_decay=weight_decay, momentum=0.99, nesterov=True)
    if fused:
        try:
            return torch.optim.SGD(params, initial_lr, fused=True, **kw)
        except (RuntimeError, TypeError, ValueError):   # a torch build without the fused implementation for this device
            pass
    return torch.optim.SGD(params, initial_lr, **kw)


def wrap_data_parallel(net: nn.Module, device, find_unused_parameters: bool = False, bucket_cap_mb: int = 25) -> nn.Module:
    """One replica per process / GPU.  ``find_unused_parameters`` is needed for the 2-D net only (its ``decoder_3`` owns two D-LKA blocks
    that never run).  ``gradient_as_bucket_view`` lets RCCL reduce in place.  Without an initialised process group (single
    GPU) the network is returned as is."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return net
    dev = torch.device(device)
    ids = [dev.index] if dev.type == "cuda" else None
    set_wgrad_overlap(net, False)   # (DDP's bucket hooks read a gradient as soon as autograd hands it over: it must be complete by then)
    return nn.parallel.DistributedDataParallel(net, device_ids=ids, find_unused_parameters=find_unused_parameters, bucket_cap_mb=bucket_cap_mb,
                                               gradient_as_bucket_view=True)


def deep_supervision_loss(outputs, target, weights=None, base_loss: Callable = None):
    """The trainer weights the deep-supervision heads 1, 1/2, 1/4, ... (normalised; nnU-Net's ``MultipleOutputLoss2``).  ``base_loss``
    defaults to cross-entropy (the reference adds a soft-Dice term from nnU-Net, outside the hot path).  target: the list of label volumes at
    the heads' resolutions, or one full-resolution volume that is nearest-neighbour down-sampled here."""
    base_loss = base_loss or nn.functional.cross_entropy
    if not isinstance(outputs, (list, tuple)):
        return base_loss(outputs, target)
    n = len(outputs)
    weights = weights or [1.0 / (2 ** i) for i in range(n)]
    s = sum(weights)
    total = 0.0
    for i, out in enumerate(outputs):
        tgt = target[i] if isinstance(target, (list, tuple)) else target
        if tgt.shape[-3:] != out.shape[-3:]:
            tgt = nn.functional.interpolate(tgt[:, None].float(), size=out.shape[-3:], mode="nearest")[:, 0].long()
        total = total + (weights[i] / s) * base_loss(out, tgt)
    return total


def run_iteration(net: nn.Module, optimizer, data: torch.Tensor, target, loss_fn: Callable = deep_supervision_loss, do_backprop: bool = True,
                  clip_norm: float = 12.0, bf16_autocast: bool = False, forward: Callable = None):
    """:259-309: zero_grad, forward, loss, backward, clip_grad_norm_(12), step.  ``bf16_autocast`` wraps forward + loss in
    ``torch.autocast(dtype=bfloat16)`` (bf16 needs no gradient scaler — the reference's fp16 branch does, :281-290).  What that changes HERE: torch's own
    layers (norms, activations, the loss) follow autocast; each of the 21 ``TransformerBlock_3D_single_deform_LKA`` wrappers runs MIXED — its D-LKA
    attention on bf16 activations (``dlka_tblock3d_*`` with dtype = DLKA_BF16: the token kernels' bf16 path), its own residual stream, LayerNorm /
    BatchNorm statistics and UnetResBlock convs in fp32 — and the conv re-expressions of ``network.Convolution`` stay fp32 GEMMs."""
    fwd = forward if forward is not None else net      # (modules whose forward takes more than the data tensor)
    optimizer.zero_grad()
    if bf16_autocast:
        with torch.autocast(data.device.type, dtype=torch.bfloat16):
            output = fwd(data)
            loss = loss_fn(output, target)
    else:
        output = fwd(data)
        loss = loss_fn(output, target)
    if do_backprop:
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), clip_norm)
        optimizer.step()
    return loss.detach()


class GraphedIteration:
    """``run_iteration`` captured ONCE in a hipGraph and replayed: forward, loss, backward, ``clip_grad_norm_`` and the optimizer step become one
    graph launch per iteration (the D-LKA autograd Functions launch on the capturing stream; every tensor the iteration creates comes from the
    graph's private pool).  Static shapes: ``data`` / ``target`` are copied into the captured input buffers.  The learning rate is read when the graph
    is captured — build a new one after changing it (the reference's poly schedule changes it once per EPOCH); a call with a changed rate raises
    instead of silently stepping with the old one.  Returns a COPY of the loss (the captured buffer is overwritten by the next replay).  BatchNorm
    layers with ``momentum=None`` are rejected (host read of ``num_batches_tracked``).  Single process; with ``DistributedDataParallel`` use the
    eager ``run_iteration``."""

    def __init__(self, net: nn.Module, optimizer, data: torch.Tensor, target: torch.Tensor, loss_fn: Callable = deep_supervision_loss,
                 clip_norm: float = 12.0, warmup: int = 3):
        if not data.is_cuda:
            raise RuntimeError("GraphedIteration needs GPU tensors")
        for name, m in net.named_modules():   # cumulative-average BatchNorm reads num_batches_tracked on the HOST every step: a device sync inside the
            if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.track_running_stats and m.momentum is None:   # capture, or a value baked into the graph
                raise RuntimeError(f"GraphedIteration: BatchNorm layer {name!r} has momentum=None (cumulative moving average), which cannot be captured "
                                   "in a hipGraph; give it a momentum or use the eager run_iteration")
        self.net, self.optimizer = net, optimizer
        self._captured_lrs = [float(g["lr"]) for g in optimizer.param_groups]   # the graph holds these values (see __call__)
        self.data, self.target = data.clone(), target.clone()
        side = torch.cuda.Stream(device=data.device)
        side.wait_stream(torch.cuda.current_stream(data.device))
        with torch.cuda.stream(side):   # warm-up off the default stream: optimizer state, lazily built tables and workspaces exist before the capture
            for _ in range(max(1, warmup)):
                run_iteration(net, optimizer, self.data, self.target, loss_fn, True, clip_norm)
        torch.cuda.current_stream(data.device).wait_stream(side)
        optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = run_iteration(net, optimizer, self.data, self.target, loss_fn, True, clip_norm)

    def __call__(self, data: torch.Tensor = None, target: torch.Tensor = None) -> torch.Tensor:
        if data is not None:
            self.data.copy_(data, non_blocking=True)
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        lrs = [float(g["lr"]) for g in self.optimizer.param_groups]
        if lrs != self._captured_lrs:
            raise RuntimeError(f"GraphedIteration: the optimizer's learning rate changed ({self._captured_lrs} -> {lrs}) but the captured graph holds the "
                               "old value; build a new GraphedIteration after changing it")
        self.graph.replay()
        return self.loss.clone()   # (a copy: ``self.loss`` is the graph's static output buffer, overwritten by the next replay)
