"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

The oracle-assembled ``D_LKA_Former``: the net's own assembly (``deformablelka_amd.network``, whose plumbing is pinned to the reference class at
the full patch by tests/test_nets.py with a stand-in block) run on the CPU with every ``TransformerBlock_3D_single_deform_LKA`` replaced by
``OracleTransformerBlock`` — same constructor, same parameters / buffers / ``state_dict`` keys, forward = ``oracle.blocks.transformer_block_3d``
(ATen CPU convs + the C oracle for the deformable conv: 3D/d_lka_former/network_architecture/synapse/transformerblock.py:617-630).  On the CPU the
plumbing layers are the stock torch layers (``Convolution.forward`` / the norms fall through to them off the GPU).

Used by the assembled-net parity tests: D_LKA_Former(trans_block=TransformerBlock_3D_single_deform_LKA) on the HIP kernels against
D_LKA_Former(trans_block=OracleTransformerBlock) — logits, argmax agreement, loss, parameter gradients (d_lka_former_synapse.py:144-167,
model_components.py:52-66)."""
import torch
import torch.nn.functional as F

import deformablelka_amd as dk
from deformablelka_amd import ops
from oracle import blocks


class OracleTransformerBlock(dk.TransformerBlock_3D_single_deform_LKA):
    """Parameters of the HIP module, arithmetic of the oracle.  Class-level hooks (set by ``run_pair``):
    ``offsets_in``  — iterator over per-block offset VALUES to sample with (straight through; None = the block's own prediction),
    ``offsets_log`` — list receiving the offsets each block sampled with, ``masks`` — iterator over the Dropout3d multipliers to use."""
    offsets_in = None
    offsets_log = None
    masks = None
    signs_in = None    # iterator over per-block (s1, s2, s2_known) NCDHW bool tensors: the kernels' LeakyReLU activation patterns (blocks.leaky_relu_signed)
    kink_log = None    # list receiving (differing elements, largest |z| / max|z| among them, elements) per LeakyReLU

    def forward(self, x, keep_channels_last=None):
        cls = OracleTransformerBlock
        P = dict(self.named_parameters())
        P.update(dict(self.named_buffers()))
        if self.pos_embed is None:
            P["pos_embed"] = None
        B, C = x.shape[:2]
        drop = self.conv8[0]
        mask = None
        if drop.training and drop.p > 0:
            mask = next(cls.masks) if cls.masks is not None else self._draw_drop_mask(B, C, x.dtype, x.device)
        override = next(cls.offsets_in) if cls.offsets_in is not None else None
        signs = next(cls.signs_in) if cls.signs_in is not None else None
        used, pre = [], []
        y = blocks.transformer_block_3d(x.contiguous(), P, self.conv51.norm1.training, mask, offsets_override=override, offsets_out=used, act_signs=signs,
                                        pre_out=pre)
        if cls.offsets_log is not None:
            cls.offsets_log.append(used[0])
        if cls.kink_log is not None:
            cls.kink_log.extend(kink_stats(z, pat) for z, pat in pre)
        return y.contiguous()


def kink_stats(z, pat):
    """(elements whose activation pattern differs from the own z > 0, the largest |z| / max|z| among them, elements)."""
    dis = pat != (z > 0)
    k = int(dis.sum())
    return k, (float(z[dis].abs().max() / z.abs().max().clamp_min(1e-30)) if k else 0.0), z.numel()


def liven_(net, offset_std=0.05, seed=7):
    """The constructor's gamma = 1e-6 / pos_embed = 0 / zero offset predictors would hide the D-LKA branch of every block; give them training-like values
    (as tests/golden/make_golden.py does for the single block)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for blk in net.dlka_blocks():
            blk.gamma.copy_(torch.randn(blk.gamma.shape, generator=g) * 0.2 + 0.5)
            if blk.pos_embed is not None:
                blk.pos_embed.copy_(torch.randn(blk.pos_embed.shape, generator=g) * 0.3)
            for bn in (blk.conv51.norm1, blk.conv51.norm2):
                bn.weight.copy_(torch.randn(bn.weight.shape, generator=g) * 0.2 + 1.0)
                bn.bias.copy_(torch.randn(bn.bias.shape, generator=g) * 0.2)
    blocks.randomize_offsets_(net, std=offset_std, seed=seed)


def segmentation_loss(outs, target):
    """Deep-supervision cross-entropy with nnU-Net's weights 1, 1/2, 1/4 (normalised), targets down-sampled by striding — the differentiable half of
    the trainer's DC_and_CE loss (d_lka_former_trainer_synapse.py:158-205); enough to drive every head and every block with a scalar."""
    ws = [1.0, 0.5, 0.25][:len(outs)]
    tot = 0.0
    for w, o in zip(ws, outs):
        st = [t // s for t, s in zip(target.shape[1:], o.shape[2:])]
        tg = target[:, ::st[0], ::st[1], ::st[2]]
        tot = tot + w * F.cross_entropy(o.float(), tg)
    return tot / sum(ws)


def run_pair(dev, img_size, B=1, num_classes=14, seed=0, offset_std=0.05, training=True, backward=True):
    """Runs the HIP net and the oracle-assembled net on the same seeded input / parameters / dropout masks.
    Returns a dict: logits of both (own offsets / the kernels' offsets AND LeakyReLU activation patterns), losses, gradients, per-block offsets, flip counts,
    per-LeakyReLU counts of the patterns that differ."""
    import oracle
    torch.manual_seed(seed)
    kw = dict(in_channels=1, out_channels=num_classes, img_size=list(img_size), feature_size=16, num_heads=4, depths=[3, 3, 3, 3],
              dims=[32, 64, 128, 256], do_ds=True)
    net = dk.D_LKA_Former(**kw)
    liven_(net, offset_std)
    ref = dk.D_LKA_Former(trans_block=OracleTransformerBlock, **kw)
    ref.load_state_dict(net.state_dict(), strict=True)
    net.train(training)
    ref.train(training)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, 1, *img_size, generator=g)
    target = torch.randint(0, num_classes, (B, *img_size), generator=g)
    nblk = len(net.dlka_blocks())
    masks = [F.dropout3d(torch.ones(B, blk.gamma.numel(), 1, 1, 1), blk.conv8[0].p, True).view(B, -1) for blk in net.dlka_blocks()] if training else None

    # ---- HIP net ----
    net = net.to(dev)
    if training:
        it = iter(masks)
        for blk in net.dlka_blocks():
            blk._draw_drop_mask = (lambda B_, C_, dtype, device, _it=it: next(_it).to(device=device, dtype=dtype))
    saved_log, sign_log = [], []
    orig = ops.tblock3d_forward

    def spy(x_, x_planar, tparams, lka_params, drop_mask, training_, bn_stats, dims, *a, **k):
        out = orig(x_, x_planar, tparams, lka_params, drop_mask, training_, bn_stats, dims, *a, **k)
        C_ = int(lka_params[0].shape[0])
        B_ = int(x_.numel() // (C_ * dims[0] * dims[1] * dims[2]))
        saved_log.append(ops.tblock3d_saved_offsets(out[1], B_, C_, dims).cpu().clone())
        sign_log.append(tuple(t.cpu().permute(0, 2, 1).reshape(B_, C_, *dims) for t in ops.tblock3d_saved_activation_signs(out[1], B_, C_, dims)))
        return out

    # the plumbing's LeakyReLUs (encoder1 / decoder2: network.UnetResBlock.lrelu, called twice per forward): the pattern of every call, by module name
    plumb_signs, hooks = {}, []
    for name, mod in net.named_modules():
        if isinstance(mod, torch.nn.LeakyReLU) and ".conv51." not in name:   # (a block's own UnetResBlock runs inside the fused call: its pattern is in `saved`)
            hooks.append(mod.register_forward_hook(lambda _m, _i, out, _n=name: plumb_signs.setdefault(_n, []).append((out.detach() > 0).cpu())))
    ops.tblock3d_forward = spy
    try:
        xd = x.to(dev)
        if backward:
            outs = net(xd)
            loss = segmentation_loss(outs, target.to(dev))
            loss.backward()
        else:
            with torch.no_grad():
                outs = net(xd)
            loss = segmentation_loss(outs, target.to(dev))
    finally:
        ops.tblock3d_forward = orig
        for h in hooks:
            h.remove()
    assert len(saved_log) == nblk, (len(saved_log), nblk)
    res = {"hip_logits": [o.detach().float().cpu() for o in outs], "hip_loss": float(loss.detach()), "hip_offsets": saved_log,
           "hip_grads": {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters() if p.grad is not None} if backward else {}}

    # ---- oracle-assembled net: (1) on its own offsets, (2) on the kernels' offset values (identical sampling cells) ----
    def run_ref(offsets, signs=None):
        """signs: (per-block patterns, per-module plumbing patterns) of the HIP net's forward pass — every LeakyReLU of the oracle net then takes the kernels' side of
        its kink (tests/parity.py, the kink protocol); the elements where that differs from the oracle's own z > 0 are counted in kink_log."""
        cls = OracleTransformerBlock
        ref.zero_grad(set_to_none=True)
        cls.offsets_in = iter(offsets) if offsets is not None else None
        cls.offsets_log = []
        cls.masks = iter(masks) if training else None
        cls.signs_in = iter(signs[0]) if signs is not None else None
        cls.kink_log = []
        patched = []
        if signs is not None:
            for name, mod in ref.named_modules():
                if isinstance(mod, torch.nn.LeakyReLU) and name in signs[1]:
                    it = iter(signs[1][name])

                    def fwd(z, _it=it, _slope=mod.negative_slope):
                        pat = next(_it).contiguous()   # (non-contiguous conditions: see oracle.blocks.leaky_relu_signed)
                        cls.kink_log.append(kink_stats(z.detach(), pat))
                        return torch.where(pat, z, _slope * z)
                    mod.forward = fwd
                    patched.append(mod)
        try:
            if backward:
                o = ref(x)
                l = segmentation_loss(o, target)
                l.backward()
            else:
                with torch.no_grad():
                    o = ref(x)
                l = segmentation_loss(o, target)
            log = cls.offsets_log
            run_ref.kinks = cls.kink_log
        finally:
            cls.offsets_in = cls.offsets_log = cls.masks = cls.signs_in = cls.kink_log = None
            for mod in patched:
                del mod.forward
        return ([t.detach().clone() for t in o], float(l.detach()), log,
                {k: p.grad.detach().clone() for k, p in ref.named_parameters() if p.grad is not None} if backward else {})

    res["ref_logits"], res["ref_loss"], res["ref_offsets"], res["ref_grads"] = run_ref(None)
    res["same_logits"], res["same_loss"], _, res["same_grads"] = run_ref(saved_log, (sign_log, plumb_signs))
    res["kinks"] = run_ref.kinks   # per LeakyReLU of the "same" run: (patterns that differ from the oracle's own, how close to 0 they are, elements)
    k3 = ((3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1))
    flips, total = 0, 0
    for oh, orf in zip(res["hip_offsets"], res["ref_offsets"]):
        dims = tuple(oh.shape[2:])
        i_h, m_h = oracle.deform_conv3d_sample_index(oh, dims, *k3)
        i_r, m_r = oracle.deform_conv3d_sample_index(orf, dims, *k3)
        flips += int(((i_h != i_r).any(-1) | (m_h != m_r)).sum())
        total += m_r.numel()
    res["flipped"], res["samples"] = flips, total
    return res


def summarize(res, top=8):
    """Numbers the tests assert on (and print): max |logit difference| per head, argmax agreement of the full-resolution head, loss differences,
    relative gradient errors (max-norm) of every parameter — against the oracle net on its own offsets and on identical cells."""
    from tests.parity import rel_err
    out = {"flipped": res["flipped"], "samples": res["samples"]}
    if "kinks" in res:
        out["kink_differ"] = sum(k for k, _, _ in res["kinks"])
        out["kink_worst_rel_z"] = max([r for _, r, _ in res["kinks"]] + [0.0])
        out["kink_elements"] = sum(n for _, _, n in res["kinks"])
    for tag in ("ref", "same"):
        lg = res[tag + "_logits"]
        out[tag + "_logit_abs"] = [float((a - b).abs().max()) for a, b in zip(res["hip_logits"], lg)]
        out[tag + "_logit_scale"] = [float(b.abs().max()) for b in lg]
        out[tag + "_argmax_agree"] = float((res["hip_logits"][0].argmax(1) == lg[0].argmax(1)).float().mean())
        out[tag + "_loss_abs"] = abs(res["hip_loss"] - res[tag + "_loss"])
        gr = res[tag + "_grads"]
        errs = {k: rel_err(res["hip_grads"][k], g) for k, g in gr.items() if k in res["hip_grads"] and g.abs().max() > 0}
        out[tag + "_grad_errs"] = errs
        out[tag + "_grad_worst"] = sorted(errs.items(), key=lambda kv: -kv[1])[:top]
    return out
