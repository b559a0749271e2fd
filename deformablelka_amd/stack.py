"""Static-memory engine for a stack of 3-D D-LKA blocks (the 21 blocks one 64x128x128 patch traverses in
D_LKA_Former, 3D/d_lka_former/network_architecture/synapse/model_components.py:33-39,127-131).

MI355X-first layout: every parameter lives in ONE flat HBM buffer (``flat_params``), every gradient in one flat
buffer of the same layout (``flat_grads``) — a single RCCL all-reduce per step with no bucketing copies —, saved
activations of all blocks stay resident (288 GB HBM makes recomputation pointless at this size), and the whole
fwd+bwd is a fixed sequence of C-ABI calls on one stream, so it can be captured in a hipGraph and replayed.
"""
from __future__ import annotations

from ctypes import byref
from typing import List, Sequence, Tuple

import os

import torch

from . import _lib as L

# (C, (H, W, D), number of blocks) for a 64x128x128 patch with the (2,4,4) stem — SURVEY.md §8 stage table
SYNAPSE_STAGES: Tuple[Tuple[int, Tuple[int, int, int], int], ...] = (
    (32, (32, 32, 32), 6), (64, (16, 16, 16), 6), (128, (8, 8, 8), 6), (256, (4, 4, 4), 3))
CHAIN = 3  # blocks per stage instance are applied back to back (model_components.py:33-39)


def _param_shapes(C: int):
    return [(C, C, 1, 1, 1), (C,), (C, 1, 5, 5, 5), (C,), (C, 1, 7, 7, 7), (C,), (81, C, 3, 3, 3), (81,),
            (C, C, 3, 3, 3), (C,), (C, C, 1, 1, 1), (C,), (C, C, 1, 1, 1), (C,)]


def _offset_std_for(C: int, offset_std_voxels: float = 1.0) -> float:
    """std of conv_offset.weight ~ N(0, .) that makes the PREDICTED offsets ~offset_std_voxels at stage width C.  The input of conv_offset
    (after proj_1, GELU and the two default-initialised depthwise convs) shrinks with C, so the gain is calibrated per stage width:
    measured on MI355X with gain 3/sqrt(fan_in) the predicted offsets had std 0.272 / 0.190 / 0.104 / 0.080 at C = 32 / 64 / 128 / 256
    (scripts/stack_stats.py); the table brings all four to ~1.0 and is frozen."""
    import math
    calib = {32: 3.68, 64: 5.26, 128: 9.6, 256: 12.5}.get(C, 1.0)
    return offset_std_voxels * calib * 3.0 / math.sqrt(C * 27)


def chain_order(stages: Sequence, order: str = "unet"):
    """The chains (stage instances of up to CHAIN blocks) of a patch in EXECUTION order.

    "unet" (default): the order D_LKA_Former's own forward pass visits them — the encoder's stage instances from the widest volume to the bottleneck, then the
    decoder's back up (model_components.py:33-39 the encoder's four stages, :127-131 the decoder's; network_architecture/synapse/d_lka_former_synapse.py chains them):
    of a stage's chains the first is the encoder's, the others the decoder's.  The backward pass runs it in reverse, i.e. it STARTS with the decoder's 32^3 blocks —
    whose weight gradients (the largest of the step) then trail on the side stream under the small stages' data chains, which leave most of the chip idle.
    "stages": stage by stage as the table lists them (rounds 1 - 4; DLKA_STACK_ORDER=stages), small stages first in the backward pass."""
    per_stage = [[(C, dims, min(CHAIN, n - c0)) for c0 in range(0, n, CHAIN)] for C, dims, n in stages]
    if order == "stages":
        return [c for cs in per_stage for c in cs]
    return [cs[0] for cs in per_stage if cs] + [c for cs in reversed(per_stage) for c in cs[1:]]


class _Block:
    __slots__ = ("C", "dims", "params", "grads", "pstruct", "gstruct", "saved", "x", "y", "gx", "gy", "saved_bytes", "partials", "partials_bytes")


class DLKABlockStack:
    def __init__(self, batch: int, stages: Sequence = SYNAPSE_STAGES, device="cuda:0", dtype=torch.float32, seed: int = 0,
                 offset_std_voxels: float = 1.0, data_seed=None, defer_finalize: bool = True, overlap_wgrad: bool = None):
        """seed: parameters (identical on every data-parallel rank); data_seed: the synthetic inputs / grad_outputs of THIS rank's
        batch shard (None: drawn from the parameter generator, single-process use)."""
        self.B, self.device, self.dtype = batch, torch.device(device), dtype
        self.lib = L.get_lib()
        self.dt = L.DLKA_F32 if dtype == torch.float32 else L.DLKA_BF16
        gen = torch.Generator().manual_seed(seed)
        self.order = os.environ.get("DLKA_STACK_ORDER", "unet")
        chain_specs = chain_order(stages, self.order)
        shapes_all = []
        for C, dims, n in chain_specs:
            for _ in range(n):
                shapes_all.append((C, dims, _param_shapes(C)))
        total = sum(sum(int(torch.Size(s).numel()) for s in sh) for _, _, sh in shapes_all)
        # dtype is the ACTIVATION storage type (float32, or bfloat16 = DLKA_BF16); parameters and gradients are fp32 masters either way
        self.flat_params = torch.empty(total, dtype=torch.float32, device=self.device)
        self.flat_grads = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.blocks: List[_Block] = []
        self.chains: List[List[_Block]] = []
        off = 0
        ws_bytes = 0
        for C, dims, shapes in shapes_all:
            blk = _Block()
            blk.C, blk.dims = C, dims
            blk.params, blk.grads = [], []
            for s in shapes:
                n = int(torch.Size(s).numel())
                blk.params.append(self.flat_params[off:off + n].view(s))
                blk.grads.append(self.flat_grads[off:off + n].view(s))
                off += n
            self._init_block(blk, gen, offset_std_voxels)
            blk.pstruct = L.Lka3dPtrs(*[p.data_ptr() for p in blk.params])
            blk.gstruct = L.Lka3dPtrs(*[g.data_ptr() for g in blk.grads])
            H, W, D = dims
            if not self.lib.dlka_lka3d_tokens_supported(batch, C, H, W, D, self.dt):
                raise RuntimeError(f"token-layout D-LKA block does not support C={C}, dtype={dtype}")
            blk.saved_bytes = self.lib.dlka_lka3d_tokens_saved_bytes(batch, C, H, W, D, self.dt)
            blk.saved = torch.empty(blk.saved_bytes, dtype=torch.uint8, device=self.device)
            ws_bytes = max(ws_bytes, self.lib.dlka_lka3d_tokens_workspace_bytes(batch, C, H, W, D, self.dt))
            self.blocks.append(blk)
        self.ws_bytes = ws_bytes
        self.ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        self._build_prepare_plan()
        self._build_finalize_plan(defer_finalize)
        self._build_wgrad_overlap(overlap_wgrad)
        # activations: blocks of one stage instance are chained x -> y -> ... ; each chain has a synthetic input and
        # a synthetic grad_output (the layers between chains — down/up-sampling, UnetResBlock — are not D-LKA).
        if data_seed is not None:
            gen = torch.Generator().manual_seed(int(data_seed))
        i = 0
        for C, dims, n in chain_specs:
            chain = self.blocks[i:i + n]
            shape = (batch, dims[0] * dims[1] * dims[2], C)   # token layout [B, N, C]
            acts = [torch.randn(shape, generator=gen).to(self.device, dtype)] + \
                   [torch.empty(shape, dtype=dtype, device=self.device) for _ in chain]
            gacts = [torch.empty(shape, dtype=dtype, device=self.device) for _ in chain] + \
                    [torch.randn(shape, generator=gen).to(self.device, dtype)]
            for j, blk in enumerate(chain):
                blk.x, blk.y = acts[j], acts[j + 1]
                blk.gx, blk.gy = gacts[j], gacts[j + 1]
            self.chains.append(chain)
            i += n

    # reference initialisers: nn.Conv3d default (kaiming_uniform a=sqrt(5) + uniform bias), DeformConv weight/bias
    # (3D/dcn/modules/deform_conv.py:44-50).  conv_offset would be zero (deform_conv.py:86-88) which makes every
    # sample integer-aligned; for timing it is drawn so that predicted offsets have ~offset_std_voxels std.
    def _init_block(self, blk, gen, offset_std):
        import math
        for idx in range(0, 14, 2):
            w, b = blk.params[idx], blk.params[idx + 1]
            fan_in = int(w[0].numel())
            bound = 1.0 / math.sqrt(fan_in)
            w.copy_(((torch.rand(w.shape, generator=gen) * 2 - 1) * bound).to(self.device))
            b.copy_(((torch.rand(b.shape, generator=gen) * 2 - 1) * bound).to(self.device))
        ow, ob = blk.params[6], blk.params[7]
        ow.copy_((torch.randn(ow.shape, generator=gen) * _offset_std_for(blk.C, offset_std)).to(self.device))
        ob.zero_()

    def _build_prepare_plan(self):
        """The prepared (MFMA-operand-order) weight forms of ALL blocks are produced by ONE launch per step (``prepare()``), not by one launch
        inside every block's forward call: the job table is built here once — parameter and ``saved`` addresses never change — and lives on
        the device (include/dlka.h: dlka_lka3d_tokens_prepare_*)."""
        import ctypes
        n = len(self.blocks)
        nbytes = self.lib.dlka_lka3d_tokens_prepare_plan_bytes(n)
        params = (L.Lka3dPtrs * n)(*[b.pstruct for b in self.blocks])
        saved = (ctypes.c_void_p * n)(*[b.saved.data_ptr() for b in self.blocks])
        sbytes = (ctypes.c_size_t * n)(*[b.saved_bytes for b in self.blocks])
        dims = (ctypes.c_int * (5 * n))(*[v for b in self.blocks for v in (self.B, b.C, *b.dims)])
        self._plan_host = torch.zeros(nbytes, dtype=torch.uint8)
        rc = self.lib.dlka_lka3d_tokens_prepare_plan(n, params, saved, sbytes, dims, self.dt, ctypes.c_void_p(self._plan_host.data_ptr()), nbytes)
        L.check(rc, "lka3d_tokens_prepare_plan")
        self._plan_dev = self._plan_host.to(self.device) if self.device.type == "cuda" else self._plan_host

    def _build_finalize_plan(self, enable: bool):
        """The seven weight gradients of a block end in a "finalize" launch that folds their partial sums; nothing later in the backward pass
        reads its results.  With block-PRIVATE partial-sum areas (288 GB of HBM: 0.4 GB for the 21 blocks) all blocks' finalisations become ONE
        launch at the end of the backward pass (or of a slice of it) instead of 21 dependent launches of 15 - 30 us, most of each latency
        (include/dlka.h: dlka_wgrad_finalize_*).  The job table is recorded while the blocks go through their first backward pass (which still
        finalises block by block) and then lives on the device."""
        self._fin_host = self._fin_dev = None
        self._fin_sealed = False
        self._fin_recorded = set()
        import os
        if not enable or os.environ.get("DLKA_STACK_PER_BLOCK_FINALIZE"):   # (A/B switch: the per-block finalize launches)
            return
        import ctypes
        n = len(self.blocks)
        for blk in self.blocks:
            H, W, D = blk.dims
            blk.partials_bytes = self.lib.dlka_lka3d_tokens_partials_bytes_v(self.B, blk.C, H, W, D, self.dt, 0)
            blk.partials = torch.empty(blk.partials_bytes, dtype=torch.uint8, device=self.device)
        nbytes = self.lib.dlka_wgrad_finalize_plan_bytes(n)
        self._fin_host = torch.zeros(nbytes, dtype=torch.uint8)
        L.check(self.lib.dlka_wgrad_finalize_plan_init(ctypes.c_void_p(self._fin_host.data_ptr()), nbytes, n), "wgrad_finalize_plan_init")

    def _build_wgrad_overlap(self, enable):
        """The five weight-gradient launches of a block only READ what its data-gradient chain leaves in the workspace, and nothing before the
        finalisation at the end of the pass reads THEIR results: they run on a side stream behind an event, overlapping the next block's data chain
        (dlka_lka3d_attention_tokens_backward_phase_v).  Two workspaces alternate; the data chain of a block waits for the weight gradients that last
        used its workspace.  Under hipGraph capture the events become a fork / join in the graph.  Needs the deferred finalisation."""
        import os
        if enable is None:   # default: on (measured 11.95 -> 11.47 ms per step under hipGraph replay); DLKA_STACK_WGRAD_OVERLAP=0 = one stream
            enable = os.environ.get("DLKA_STACK_WGRAD_OVERLAP", "1") != "0"
        self._overlap = bool(enable) and self._fin_host is not None
        # blocks narrower than this run their weight gradients IN LINE on the calling stream (A/B knob: at the widest-volume stage the data chain fills
        # the chip by itself, so the side stream's kernels mostly stretch it; profiles/r05_notes.md has the measurement behind the default)
        self._overlap_min_c = int(os.environ.get("DLKA_STACK_WGRAD_OVERLAP_MIN_C", "0"))
        self._prep_pending, self._prep_split = None, 0
        if not self._overlap:
            return
        # Workspaces of the backward pass, used round-robin: block n's data chain writes pool[n % K], its weight gradients read it on the side stream, and the
        # data chain of block n + K waits for them.  K = 2 (rounds 2 - 4) ties the side stream to the data chain within one block; with the decoder's 32^3 blocks at
        # the head of the backward pass (chain_order) a deeper pool lets their weight gradients trail under the small stages.  288 GB of HBM: 0.44 GB each.
        # Measured (profiles/r08_notes.md, `r8e`): a deep ROUND-ROBIN pool is slower (K = 2 / 4 / 8 / 12: 10.50 / 10.71 / 10.86 / 11.03 ms per step) — the small
        # stages' working sets then rotate through K x their size and fall out of the L2 / Infinity Cache.  So: two alternating workspaces as before, plus
        # DLKA_STACK_TRAIL private ones for the FIRST blocks of the backward pass (the decoder's 32^3 chain), whose weight gradients may then trail freely.
        self._trail = max(0, int(os.environ.get("DLKA_STACK_TRAIL", "0")))
        self._pool_k = 2 + self._trail
        self._ws_pool = [self.ws] + [torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device) for _ in range(self._pool_k - 1)]
        self.ws2 = self._ws_pool[1]
        if self.device.type == "cuda":
            self._side = torch.cuda.Stream(device=self.device)
            self._ev_data = [torch.cuda.Event() for _ in range(self._pool_k)]
            self._ev_wg = [torch.cuda.Event() for _ in range(self._pool_k)]
            self._ev_prep, self._ev_prep2 = torch.cuda.Event(), torch.cuda.Event()
            self._ev_fin = torch.cuda.Event()
            c0 = self.blocks[0].C
            self._prep_split = next((i for i, b in enumerate(self.blocks) if b.C != c0), len(self.blocks))   # the leading blocks of the first width
        else:
            self._side = None

    def prepare(self):
        """Re-lay the weights of all blocks (after every parameter update): ONE launch, complete in stream order on the calling stream — safe to follow
        with ``backward()`` or any reader of the prepared weights.  (``forward()`` uses the split form below.)"""
        self._prepare(split=False)

    def _prepare(self, split: bool):
        """split (``forward()`` only): when a side stream exists, two launches — the blocks of the first stage (small weights) on the calling stream, the
        rest (95 % of the bytes) on the side stream while the first stage's forward passes run; ``forward`` joins it in front of the first block that needs
        it (and ``backward`` would, were it ever entered with the join still pending), so the fork never outlives the call that made it."""
        n, st = len(self.blocks), self._stream()
        self._join_prepare()
        k = self._prep_split if split and getattr(self, "_overlap", False) and getattr(self, "_side", None) is not None else 0
        if 0 < k < n:
            cur = torch.cuda.current_stream(self.device)
            L.check(self.lib.dlka_lka3d_tokens_prepare_run_range(L.ptr(self._plan_dev), L.ptr(self._plan_host), n, 0, k, st), "lka3d_tokens_prepare_run_range")
            self._ev_prep.record(cur)            # (behind the parameter update that precedes this call on the calling stream)
            self._side.wait_event(self._ev_prep)
            L.check(self.lib.dlka_lka3d_tokens_prepare_run_range(L.ptr(self._plan_dev), L.ptr(self._plan_host), n, k, n, self._side.cuda_stream),
                    "lka3d_tokens_prepare_run_range")
            self._ev_prep2.record(self._side)
            self._prep_pending = k
            return
        rc = self.lib.dlka_lka3d_tokens_prepare_run(L.ptr(self._plan_dev), L.ptr(self._plan_host), n, st)
        L.check(rc, "lka3d_tokens_prepare_run")

    def _join_prepare(self):
        if self._prep_pending is not None:   # weights of blocks >= _prep_pending are still in flight on the side stream
            torch.cuda.current_stream(self.device).wait_event(self._ev_prep2)
            self._prep_pending = None

    def _stream(self):
        if self.device.type != "cuda":   # only reachable through the CPU test backend (tests/emu)
            return None
        return torch.cuda.current_stream(self.device).cuda_stream

    def forward(self, on_block=None):
        """on_block(i): called before block i is issued (bench.py's launch trace separates the blocks with it)."""
        st = self._stream()
        self._prepare(split=True)
        for i, blk in enumerate(self.blocks):
            if on_block is not None:
                on_block(i)
            if self._prep_pending is not None and i == self._prep_pending:   # the weights of this and the later blocks were prepared on the side stream
                self._join_prepare()
            H, W, D = blk.dims
            rc = self.lib.dlka_lka3d_attention_tokens_forward_prepared(L.ptr(blk.x), byref(blk.pstruct), L.ptr(blk.y), L.ptr(blk.saved),
                                                                blk.saved_bytes, L.ptr(self.ws), self.ws_bytes, self.B, blk.C, H, W, D,
                                                                self.dt, st)
            L.check(rc, "lka3d_attention_tokens_forward_prepared")
        self._join_prepare()   # (a stack with no block behind the split point: never leave the fork open)

    def backward(self, lo: int = 0, hi: int = None, on_block=None):
        """Backward pass of blocks[lo:hi] in reverse order (default: all)."""
        import ctypes
        st = self._stream()
        self._join_prepare()
        idx = list(range(len(self.blocks)))[lo:hi]
        defer = self._fin_host is not None
        plan_ptr = ctypes.c_void_p(self._fin_host.data_ptr()) if defer else None
        overlap = defer and getattr(self, "_overlap", False)
        side = self._side if overlap else None
        K = self._pool_k if overlap else 2
        used = [False] * K   # workspace k has weight gradients in flight on the side stream
        # Sealed plan + side stream: the folds of a block's partial sums follow its weight gradients ON THE SIDE STREAM, a few blocks per launch, instead
        # of one launch for the whole pass after the join (330 us exposed at the end of every step, profiles/r04s): that stream has the slack (the weight
        # gradients are ~40 % of a block's backward work) and only the last group's fold is left behind the last block.  DLKA_STACK_FINALIZE_GROUP=0:
        # the single launch at the end.  Measured (profiles/r05_notes.md, ms per step): 0 -> 11.135, 1 -> 11.044, 2 -> 11.008, 3 -> 11.09, 5 -> 11.09; on round 5's tree
        # (faster folds and weight gradients: the side stream has more slack) one block per launch is ahead, 10.07 against 10.11 - 10.14 fp32 and 9.32 against 9.36 bf16
        # (profiles/r09h_ab_fin_*.json, each configuration twice in one process).
        fin_group = int(os.environ.get("DLKA_STACK_FINALIZE_GROUP", "1"))
        side_fin = bool(overlap and side is not None and self._fin_sealed and fin_group > 0)
        pending = []   # blocks (descending) whose weight gradients are issued and whose folds are not

        def fold_pending():
            rc = self.lib.dlka_wgrad_finalize_run(L.ptr(self._fin_dev), plan_ptr, pending[-1], pending[0] + 1, side.cuda_stream)
            L.check(rc, "wgrad_finalize_run (side stream)")
            pending.clear()

        for n, i in enumerate(reversed(idx)):
            blk = self.blocks[i]
            if on_block is not None:
                on_block(i)
            H, W, D = blk.dims
            if overlap:
                k = (2 + n) if n < K - 2 else ((n - (K - 2)) & 1)   # the first K - 2 blocks of the pass: a workspace of their own; then two alternate
                ws = self._ws_pool[k]
                record = not self._fin_sealed and i not in self._fin_recorded
                args = (L.ptr(blk.x), byref(blk.pstruct), L.ptr(blk.gy), L.ptr(blk.saved), blk.saved_bytes, L.ptr(blk.gx), byref(blk.gstruct), L.ptr(ws),
                        self.ws_bytes, L.ptr(blk.partials), blk.partials_bytes)
                dims7 = (self.B, blk.C, H, W, D, self.dt, 0)
                if side is not None and used[k]:
                    torch.cuda.current_stream(self.device).wait_event(self._ev_wg[k])   # the weight gradients that last read this workspace are done
                L.check(self.lib.dlka_lka3d_attention_tokens_backward_phase_v(*args, None, i, 1, *dims7, st), "backward phase 1")
                if side is not None and blk.C < self._overlap_min_c:
                    # this block's weight gradients in line; the side stream (which folds the partial sums) continues behind them
                    L.check(self.lib.dlka_lka3d_attention_tokens_backward_phase_v(*args, plan_ptr if record else None, i, 2, *dims7, st), "backward phase 2")
                    used[k] = False   # (whatever read this workspace on the side stream was waited for in front of phase 1)
                    if side_fin:
                        self._ev_data[k].record(torch.cuda.current_stream(self.device))
                        side.wait_event(self._ev_data[k])
                        pending.append(i)
                        if len(pending) >= fin_group:
                            fold_pending()
                elif side is not None:
                    self._ev_data[k].record(torch.cuda.current_stream(self.device))
                    side.wait_event(self._ev_data[k])
                    L.check(self.lib.dlka_lka3d_attention_tokens_backward_phase_v(*args, plan_ptr if record else None, i, 2, *dims7, side.cuda_stream),
                            "backward phase 2")
                    self._ev_wg[k].record(side)
                    used[k] = True
                    if side_fin:
                        pending.append(i)
                        if len(pending) >= fin_group:
                            fold_pending()
                else:
                    L.check(self.lib.dlka_lka3d_attention_tokens_backward_phase_v(*args, plan_ptr if record else None, i, 2, *dims7, st), "backward phase 2")
                if record:
                    self._fin_recorded.add(i)
                if not self._fin_sealed:
                    if side is not None:   # (the per-slot finalize of the recording pass reads the partial sums: behind the weight gradients)
                        torch.cuda.current_stream(self.device).wait_event(self._ev_wg[k])
                        used[k] = False
                    L.check(self.lib.dlka_wgrad_finalize_run_slot(plan_ptr, i, st), "wgrad_finalize_run_slot")
            elif defer:
                record = not self._fin_sealed and i not in self._fin_recorded
                rc = self.lib.dlka_lka3d_attention_tokens_backward_deferred_v(
                    L.ptr(blk.x), byref(blk.pstruct), L.ptr(blk.gy), L.ptr(blk.saved), blk.saved_bytes, L.ptr(blk.gx), byref(blk.gstruct), L.ptr(self.ws),
                    self.ws_bytes, L.ptr(blk.partials), blk.partials_bytes, plan_ptr if record else None, i, self.B, blk.C, H, W, D, self.dt, 0, st)
                L.check(rc, "lka3d_attention_tokens_backward_deferred_v")
                if record:
                    self._fin_recorded.add(i)
                if not self._fin_sealed:   # the table is still being recorded: this block's folds as the ordinary per-block launch
                    L.check(self.lib.dlka_wgrad_finalize_run_slot(plan_ptr, i, st), "wgrad_finalize_run_slot")
            else:
                rc = self.lib.dlka_lka3d_attention_tokens_backward(L.ptr(blk.x), byref(blk.pstruct), L.ptr(blk.gy), L.ptr(blk.saved),
                                                            blk.saved_bytes, L.ptr(blk.gx), byref(blk.gstruct), L.ptr(self.ws),
                                                            self.ws_bytes, self.B, blk.C, H, W, D, self.dt, st)
                L.check(rc, "lka3d_attention_tokens_backward")
        if side is not None:   # join: the finalisation (and whatever follows the pass) is behind every weight gradient
            for k in range(K):
                if used[k]:
                    torch.cuda.current_stream(self.device).wait_event(self._ev_wg[k])
        if side_fin:
            if pending:
                fold_pending()
            self._ev_fin.record(side)
            torch.cuda.current_stream(self.device).wait_event(self._ev_fin)   # the gradients are complete for whatever follows the pass
        elif defer and idx:
            if self._fin_sealed:
                rc = self.lib.dlka_wgrad_finalize_run(L.ptr(self._fin_dev), plan_ptr, idx[0], idx[-1] + 1, st)
                L.check(rc, "wgrad_finalize_run")
            elif len(self._fin_recorded) == len(self.blocks) and not self._capturing():
                # every block has been through a backward pass once: from the next pass on, ONE launch per pass (or slice)
                L.check(self.lib.dlka_wgrad_finalize_plan_seal(plan_ptr), "wgrad_finalize_plan_seal")
                self._fin_dev = self._fin_host.to(self.device) if self.device.type == "cuda" else self._fin_host
                self._fin_sealed = True

    def _capturing(self) -> bool:
        return self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()

    def forward_backward(self, on_block=None):
        self.forward(on_block)
        self.backward(on_block=on_block)

    def split_index(self, frac: float = 0.8) -> int:
        """Block index i such that blocks[i:] hold at least `frac` of the gradient bytes and start a chain (stage instance): the backward pass produces
        those gradients FIRST (it runs the blocks in reverse), so their all-reduce can overlap the rest of the backward pass.  In the network's order
        (chain_order) the split falls in front of the encoder's C = 128 chain: 92 % of the bytes, with the two widest encoder chains still to run."""
        sizes = [sum(int(p.numel()) for p in b.params) for b in self.blocks]
        starts = set()
        i = 0
        for chain in self.chains:
            starts.add(i)
            i += len(chain)
        total, acc = sum(sizes), 0
        best = len(self.blocks)
        for i in range(len(self.blocks) - 1, -1, -1):
            acc += sizes[i]
            if i in starts:
                best = i
                if acc >= frac * total:
                    break
        return best

    def grad_offset_of(self, block_index: int) -> int:
        """Offset (elements) of a block's first gradient in ``flat_grads`` (blocks are laid out in order)."""
        return sum(sum(int(p.numel()) for p in b.params) for b in self.blocks[:block_index])

    def reduce_and_update(self, lr: float, world: int = 1, dist=None):
        """Data-parallel tail of a step: ONE all-reduce of the flat gradient buffer (RCCL over xGMI when the process
        group is ``nccl``; the batch shards across ranks with no other collective), then plain SGD on the
        flat parameter buffer with the gradient averaged over ranks."""
        if world > 1:
            dist.all_reduce(self.flat_grads)
        self.flat_params.add_(self.flat_grads, alpha=-lr / world)

    def health(self) -> dict:
        """Finite-ness of parameters / gradients and the std of the predicted offsets (voxels) of the first block of each
        stage, read from the activations the last forward pass saved."""
        finite = bool(torch.isfinite(self.flat_params).all().item() and torch.isfinite(self.flat_grads).all().item())
        from . import ops
        stds, seen = [], set()
        for blk in self.blocks:
            if blk.C in seen:
                continue
            seen.add(blk.C)
            off = ops.lka3d_tokens_saved_offsets(blk.saved, self.B, blk.C, blk.dims, self.dtype)
            stds.append(round(float(off.std().item()), 3))
        return {"finite": finite, "offset_std": stds}

    def num_params(self) -> int:
        return int(self.flat_params.numel())
