"""3-D D-LKA modules — ``TransformerBlock_3D_single_deform_LKA``, ``LKA3d_deform`` and ``LKA_Attention3d_deform`` with the
constructor / forward signatures and ``state_dict`` keys of
3D/d_lka_former/network_architecture/synapse/transformerblock.py:570-673.

``LKA_Attention3d_deform.forward(x, B, C, H, W, D)`` runs the whole block (proj_1, GELU, dw 5^3, dw 7^3 dil 3,
offset-predict conv, deformable 3^3 conv, conv1, gate, proj_2, residual) as ONE C-ABI call per direction
(``dlka_lka3d_attention_forward/backward``).  ``LKA3d_deform`` alone runs through the per-op kernels.
"""
import os
import threading
import weakref

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import nn_ops, ops
from .dynunet_block import UnetResBlock, bn_eval_stats, bn_update_running
from .modules.deform_conv import DeformConvPack


class LKA3d_deform(nn.Module):
    """transformerblock.py:634-652."""

    VARIANT = 0   # dlka_lka3d_variant (include/dlka.h): which depthwise pair the fused entry points assume

    def _make_depthwise_pair(self, dim):
        """(conv0, conv_spatial): synapse/transformerblock.py:637-638."""
        return nn.Conv3d(dim, dim, 5, padding=2, groups=dim), nn.Conv3d(dim, dim, 7, stride=1, padding=9, groups=dim, dilation=3)

    def __init__(self, dim):
        super().__init__()
        self.conv0, self.conv_spatial = self._make_depthwise_pair(dim)
        self.deform_conv = DeformConvPack(in_channels=dim, out_channels=dim, kernel_size=(3, 3, 3), stride=1, padding=1)
        self.conv1 = nn.Conv3d(dim, dim, 1)

    def forward(self, x):
        u = x
        c0, cs, c1 = self.conv0, self.conv_spatial, self.conv1
        attn = nn_ops.conv3d(x, c0.weight, c0.bias, c0.stride, c0.padding, c0.dilation, c0.groups)
        attn = nn_ops.conv3d(attn, cs.weight, cs.bias, cs.stride, cs.padding, cs.dilation, cs.groups)
        attn = attn.contiguous()
        attn = self.deform_conv(attn)
        attn = nn_ops.conv3d(attn, c1.weight, c1.bias)
        return u * attn


class _LKA3dAttentionFn(Function):
    @staticmethod
    def forward(ctx, x, *params):
        y, saved = ops.lka3d_attention_forward(x, params)
        ctx.save_for_backward(x, saved, *params)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, saved, *params = ctx.saved_tensors
        gx, grads = ops.lka3d_attention_backward(x, params, gy, saved)
        return (gx, *grads)


class _LKA3dTokensFn(Function):
    """Whole block on the token tensor [B, N, C] (channels-last end to end, no permutes)."""

    @staticmethod
    def forward(ctx, x, dims, variant, *params):
        variant, owner = variant if isinstance(variant, tuple) else (variant, None)
        y, saved = ops.lka3d_attention_tokens_forward(x, params, dims, variant)
        ctx.dims, ctx.variant = dims, variant
        ctx.owner = owner   # a weak reference to the module when it asked for the side-stream schedule (WgradOverlap), else None
        ctx.param_args = [True] * len(params)
        if owner is not None and owner() is not None:
            _note_application(owner(), ctx)
        ctx.save_for_backward(x, saved, *params)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, saved, *params = ctx.saved_tensors
        mod = ctx.owner() if ctx.owner is not None else None
        if mod is not None:
            ctx.wgrad_done[0] = True
        if mod is not None and not WgradOverlap.disabled and gy.is_cuda and WgradOverlap.eligible(ctx, mod, ctx.param_args, 3):
            ov = WgradOverlap.get(gy.device)
            gx, grads, keep = ops.lka3d_attention_tokens_backward(x, params, gy, saved, ctx.dims, ctx.variant, side_stream=ov.side)
            ov.submit(keep)
        else:
            gx, grads = ops.lka3d_attention_tokens_backward(x, params, gy, saved, ctx.dims, ctx.variant)
        return (gx, None, None, *grads)


class LKA_Attention3d_deform(nn.Module):
    """transformerblock.py:655-673."""

    GATING_UNIT = LKA3d_deform
    wgrad_overlap = False   # True: the token path's weight gradients on a side stream, joined once at the end of backward() — WgradOverlap's contract

    def __init__(self, d_model):
        super().__init__()
        self.proj_1 = nn.Conv3d(d_model, d_model, 1)
        self.activation = nn.GELU()
        self.spatial_gating_unit = self.GATING_UNIT(d_model)
        self.proj_2 = nn.Conv3d(d_model, d_model, 1)

    @property
    def variant(self):
        return self.spatial_gating_unit.VARIANT

    def block_params(self):
        """The 14 tensors in ``dlka_lka3d_params`` order (include/dlka.h)."""
        s = self.spatial_gating_unit
        d = s.deform_conv
        return (self.proj_1.weight, self.proj_1.bias, s.conv0.weight, s.conv0.bias, s.conv_spatial.weight, s.conv_spatial.bias,
                d.conv_offset.weight, d.conv_offset.bias, d.weight, d.bias, s.conv1.weight, s.conv1.bias,
                self.proj_2.weight, self.proj_2.bias)

    def forward_volume(self, x):
        """x: [B, C, H, W, D] volume -> same shape (the block without the token<->volume permutes)."""
        if self.variant != 0:   # the fused NCDHW entry point assumes the Synapse depthwise pair: other variants compose the per-op kernels
            p1, p2 = self.proj_1, self.proj_2
            a = nn_ops.gelu(nn_ops.conv3d(x, p1.weight, p1.bias))
            return nn_ops.conv3d(self.spatial_gating_unit(a), p2.weight, p2.bias) + x
        return _LKA3dAttentionFn.apply(x, *self.block_params())

    def forward(self, x, B, C, H, W, D):
        # Fast path: the token tensor IS the channels-last volume; run the block on it directly.
        # Autocast policy (the reference has none — its op would raise on half inputs, deform_conv_cuda.cu:96): inside
        # torch.autocast(dtype=torch.bfloat16) the block takes bf16 activations with fp32 parameters and accumulation.
        act = ops.autocast_activation_dtype(x)
        fp32_params = self.proj_1.weight.dtype == torch.float32
        v = self.variant
        if act != x.dtype and fp32_params and ops.lka3d_tokens_supported(act, B, C, H, W, D, v):
            x = x.to(act)   # (one cast: the support query takes the dtype, not a tensor)
        if fp32_params and ops.lka3d_tokens_supported(x.dtype, B, C, H, W, D, v):
            owner = weakref.ref(self) if (self.wgrad_overlap and x.is_cuda) else None
            return _LKA3dTokensFn.apply(x, (H, W, D), (v, owner), *self.block_params())
        if x.dtype != self.proj_1.weight.dtype:   # general path: one dtype for activations and parameters
            x = x.to(self.proj_1.weight.dtype)
        # General path (any C / dtype): the reference's own data movement around the NCDHW block.
        x = x.permute(0, 2, 1).reshape(B, C, H, W, D)  # B N C --> B C N --> B C H W D   (:665)
        x = self.forward_volume(x)
        x = x.reshape(B, C, H * W * D).permute(0, 2, 1)  # (:672)
        return x


class WgradOverlap:
    """The engine's schedule for the path the trainers call (DLKABlockStack does it for the bare attention, stack.py): a wrapper block's backward pass is issued in two
    parts — the data-gradient chain on the current stream, its weight gradients (conv51's two convs, conv8, the attention's seven) on a side stream behind an event
    (``dlka_tblock3d_backward_phase_v``) — and the side stream is joined ONCE, at the end of the whole backward pass (``queue_callback``), so that a block's weight
    gradients run under the NEXT block's data chain instead of in front of it.

    Contract (why it is opt-in, ``module.wgrad_overlap = True``; ``training.initialize_network`` / ``bench.py`` set it for their single-process loops): the parameter
    gradients a block's backward returns are complete only when ``backward()`` has returned.  Nothing may read them earlier: no DistributedDataParallel / gradient hooks
    (``training.wrap_data_parallel`` switches it off), and gradients are not accumulated into existing ``.grad`` tensors (a block whose parameters already carry a
    ``.grad`` takes the one-stream pass for that call).  It also relies on autograd taking OWNERSHIP of the gradient tensors this Function returns rather than copying them
    (AccumulateGrad does so for a dense, contiguous, otherwise unreferenced gradient when ``.grad`` is None — which is what is returned); the GPU test
    ``test_tblock3d_wgrad_overlap_equals_one_stream`` would see a copy as garbage.  One instance per device; works under hipGraph capture (the events become a fork /
    join in the graph)."""
    _by_device = {}
    disabled = os.environ.get("DLKA_TBLOCK_WGRAD_OVERLAP", "1") == "0"   # process-wide kill switch (A/B runs; tests compare both passes on ONE forward pass)

    @classmethod
    def get(cls, device):
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        o = cls._by_device.get(key)
        if o is None:
            o = cls._by_device[key] = cls(torch.device(*key))
        return o

    def __init__(self, device):
        self.device = device
        self.side = torch.cuda.Stream(device=device)
        self.pending = {}   # (thread, caller stream) -> [keep-alive lists]: two threads running backward on one device join only their own submissions

    def _key(self):
        return (threading.get_ident(), torch.cuda.current_stream(self.device).cuda_stream)

    def submit(self, keep):
        key = self._key()
        self.pending.setdefault(key, []).append(keep)
        # one callback per SUBMIT, not one per pass guarded by a flag: a backward pass that raised after a block's submit never runs its callbacks, and a flag
        # left set by it would keep every later pass from joining.  The first callback of a pass joins; the others find nothing pending.
        torch.autograd.Variable._execution_engine.queue_callback(lambda: self.join(key))

    def join(self, key=None):
        key = self._key() if key is None else key
        if not self.pending.get(key):
            self.pending.pop(key, None)
            return
        ev = torch.cuda.Event()
        ev.record(self.side)   # (behind everything submitted so far, by any caller: a superset of this caller's work)
        torch.cuda.current_stream(self.device).wait_event(ev)
        self.pending.pop(key, None)

    @staticmethod
    def eligible(ctx, mod, params, first_param_arg):
        """The side-stream pass writes EVERY weight gradient on the side stream and keeps them alive only through autograd's ownership of the returned tensors.  That
        holds only if autograd keeps each of them until backward() returns, untouched by the main stream: every parameter must require a gradient (autograd drops the
        gradient of a frozen one at once, and the caching allocator would hand its memory to the next main-stream allocation while the side stream still writes it), carry no
        ``.grad`` yet (accumulation would read it early), and the block must be applied ONCE in the graph (two applications: the engine adds their gradients on the main
        stream before the side stream has finished).  Anything else takes the one-stream pass."""
        if not all(ctx.needs_input_grad[first_param_arg + i] for i, p in enumerate(params) if p is not None):
            return False
        if not all(p.requires_grad and p.grad is None for p in mod.parameters()):
            return False
        return not ctx.wgrad_shared[0]


def _note_application(mod, ctx):
    """Marks every not-yet-differentiated application of `mod` (incl. this one) as shared when there is more than one: see WgradOverlap.eligible."""
    live = [r for r in getattr(mod, "_wgrad_live", ()) if r() is not None and not r().wgrad_done[0]]
    ctx.wgrad_shared, ctx.wgrad_done = [False], [False]
    if live:
        ctx.wgrad_shared[0] = True
        for r in live:
            r().wgrad_shared[0] = True
    live.append(weakref.ref(ctx))
    mod._wgrad_live = live


class _TBlock3dFn(Function):
    """The whole wrapper block: one C-ABI call per direction (``dlka_tblock3d_forward/backward``)."""

    @staticmethod
    def forward(ctx, x, x_planar, dims, drop_mask, training, bn_stats, eps, variant, *params):
        variant, lka_bf16, owner = (tuple(variant) + (False, None))[:3] if isinstance(variant, tuple) else (variant, False, None)
        tparams, lka_params = params[:12], params[12:]
        y, saved = ops.tblock3d_forward(x, x_planar, tparams, lka_params, drop_mask, training, bn_stats, dims, eps[0], eps[1], variant, lka_bf16)
        ctx.cfg = (x_planar, dims, training, tuple(x.shape), [p is not None for p in tparams], (variant, lka_bf16))
        ctx.param_args = [p is not None or None for p in params]   # (which of the trailing arguments are tensors: WgradOverlap.eligible reads needs_input_grad for them)
        ctx.owner = owner   # a weak reference to the module when it asked for the side-stream schedule (WgradOverlap), else None
        if owner is not None and owner() is not None:
            _note_application(owner(), ctx)
        ctx.save_for_backward(saved, bn_stats, drop_mask, *[p for p in params if p is not None])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x_planar, dims, training, xshape, present, (variant, lka_bf16) = ctx.cfg
        saved, bn_stats, drop_mask, *ps = ctx.saved_tensors
        it = iter(ps)
        tparams = [next(it) if here else None for here in present]
        lka_params = list(it)
        mod = ctx.owner() if ctx.owner is not None else None
        if mod is not None:
            ctx.wgrad_done[0] = True
        if mod is not None and not WgradOverlap.disabled and gy.is_cuda and WgradOverlap.eligible(ctx, mod, ctx.param_args, 8):
            ov = WgradOverlap.get(gy.device)
            gx, tg, lg, keep = ops.tblock3d_backward(tparams, lka_params, drop_mask, training, bn_stats, gy, saved, dims, variant, lka_bf16, side_stream=ov.side)
            ov.submit(keep)
        else:
            gx, tg, lg = ops.tblock3d_backward(tparams, lka_params, drop_mask, training, bn_stats, gy, saved, dims, variant, lka_bf16)
        if x_planar:   # gradient w.r.t. the NCDHW input: tokens -> NCDHW.  A contiguous tensor, not the permuted view: the producer of x is a
            # torch layer whose backward (MIOpen) falls to its naive "nonpacked" kernels on a strided grad_output (profiles/archive/r03e: 61 % of a
            # full-net step)
            B, C = xshape[0], xshape[1]
            gx = ops.ndhwc_to_ncdhw(gx.view(B, *xshape[2:], C))
        else:
            gx = gx.view(xshape)
        return (gx, None, None, None, None, None, None, None, *tg, *lg)


class TransformerBlock_3D_single_deform_LKA(nn.Module):
    """transformerblock.py:570-630.  ``forward(x)`` takes the (B, C, H, W, D) volume and returns a contiguous (B, C, H, W, D)
    tensor, as the reference does (:626-630), so that any consumer's ``.view()`` keeps working.

    ``keep_channels_last = True`` (an attribute, not a constructor argument: the constructor is the reference's) skips that final
    layout copy and returns the ``torch.channels_last_3d`` view of the token memory the kernels wrote; a following block recognises
    the layout and reads the tokens in place.  ``deformablelka_amd.network`` sets it on the blocks it chains."""

    keep_channels_last = False
    wgrad_overlap = False   # True: the backward pass's weight gradients on a side stream, joined at the end of backward() — see WgradOverlap for the contract
    EPA_BLOCK = LKA_Attention3d_deform

    def __init__(self, input_size: int, hidden_size: int, proj_size: int, num_heads: int, dropout_rate: float = 0.0, pos_embed=False) -> None:
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        self.norm = nn.LayerNorm(hidden_size)
        self.gamma = nn.Parameter(1e-6 * torch.ones(hidden_size), requires_grad=True)
        self.epa_block = self.EPA_BLOCK(d_model=hidden_size)
        self.conv51 = UnetResBlock(3, hidden_size, hidden_size, kernel_size=3, stride=1, norm_name="batch")
        self.conv8 = nn.Sequential(nn.Dropout3d(0.1, False), nn.Conv3d(hidden_size, hidden_size, 1))
        self.pos_embed = None
        if pos_embed:
            self.pos_embed = nn.Parameter(torch.zeros(1, input_size, hidden_size))

    def _draw_drop_mask(self, B, C, dtype, device):
        """conv8[0] = Dropout3d: the same draw F.dropout3d makes for its (B, C, 1, 1, 1) noise tensor, bernoulli(1 - p) / (1 - p),
        from the device's generator.  The kernels only consume the [B, C] multipliers.  Drawn directly (two launches) rather than as dropout3d(ones) (four:
        fill, draw, scale, multiply — 42 launches of ~4.7 us per 21-block step); same generator consumption, same values (tests/test_nets.py, GPU suite)."""
        keep = 1.0 - self.conv8[0].p
        return torch.empty(B, C, 1, 1, 1, dtype=dtype, device=device).bernoulli_(keep).div_(keep).view(B, C)

    def wrapper_params(self):
        """The 12 tensors in ``dlka_tblock3d_params`` order (include/dlka.h)."""
        c = self.conv51
        return (self.norm.weight, self.norm.bias, self.gamma, self.pos_embed, c.conv1.conv.weight, c.conv2.conv.weight, c.norm1.weight, c.norm1.bias,
                c.norm2.weight, c.norm2.bias, self.conv8[1].weight, self.conv8[1].bias)

    def forward(self, x, keep_channels_last=None):
        """keep_channels_last: None = the module attribute; True / False = this call only (``network._chain`` passes it per call)."""
        B, C, H, W, D = x.shape
        # Autocast policy (the reference registers none): inside torch.autocast(dtype=bfloat16) — or handed a bf16 tensor — the block runs MIXED: the D-LKA
        # attention on bf16 activations (DLKA_BF16: the token kernels' bf16 path, fp32 offset-determining chain), the wrapper itself — residual stream,
        # LayerNorm / BatchNorm statistics, UnetResBlock's convs — in fp32 (include/dlka.h, dlka_tblock3d_*: dtype = DLKA_BF16).  A bf16 input is widened
        # here, explicitly; the output is fp32.
        v = self.epa_block.variant
        lka_bf16 = (x.dtype == torch.bfloat16 or ops.autocast_activation_dtype(x) == torch.bfloat16) and self.norm.weight.dtype == torch.float32 \
            and ops.tblock3d_lka_bf16_supported(B, C, H, W, D, v)
        if x.dtype == torch.bfloat16:
            x = x.float()
        if not ops.tblock3d_supported(x, B, C, H, W, D, v):
            raise NotImplementedError(f"TransformerBlock_3D_single_deform_LKA on the HIP path needs float32 and hidden_size in {{32, 64, 128, 256}}; "
                                      f"got {x.dtype}, C={C}")
        tokens = x.permute(0, 2, 3, 4, 1)
        if tokens.is_contiguous():       # e.g. the previous block's output: already [B][N][C] in memory
            xin, planar = tokens, False
        else:
            xin, planar = x.contiguous(), True
        drop = self.conv8[0]
        mask = self._draw_drop_mask(B, C, x.dtype, x.device) if drop.training and drop.p > 0 else None
        c = self.conv51
        training = c.norm1.training
        if training:
            stats = torch.empty(6 * C, dtype=torch.float32, device=x.device)
        else:
            stats = torch.cat([bn_eval_stats(c.norm1), bn_eval_stats(c.norm2)])
        owner = weakref.ref(self) if (self.wgrad_overlap and x.is_cuda) else None
        y = _TBlock3dFn.apply(xin, planar, (H, W, D), mask, training, stats, (self.norm.eps, c.norm1.eps), (v, bool(lka_bf16), owner), *self.wrapper_params(),
                              *self.epa_block.block_params())
        if training:
            bn_update_running((c.norm1, stats[:3 * C]), (c.norm2, stats[3 * C:]))
        y = y.view(B, H, W, D, C).permute(0, 4, 1, 2, 3)
        keep = self.keep_channels_last if keep_channels_last is None else keep_channels_last
        return y if keep else y.contiguous()
