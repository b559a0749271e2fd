#!/usr/bin/env python
"""bench.py — 3-D D-LKA fwd+bwd volumes/s on MI355X (BASELINE.json metric).

One *step* = forward + backward through the 21 D-LKA attention blocks that one 64x128x128 Synapse patch traverses in
D_LKA_Former (6x(C=32,32^3) + 6x(64,16^3) + 6x(128,8^3) + 3x(256,4^3)), batch 2 per GPU, plus the
gradient all-reduce (N>1) and a plain SGD update of all block parameters.  Inputs are synthetic and already resident
in HBM when the timed region starts.  `value` = volumes (patches) per second over all ranks.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the driver contract), including
  "roofline":     dominant kernel, algorithmic flops|bytes per launch / HIP-event duration, vs the MI355X peak
  "cpu_baseline": the oracle block (ATen CPU convs + C deformable oracle) timed on the host cores, bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_F32_TFLOPS = 157.3     # fp32 vector == fp32-input MFMA peak
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA (not the roof of the bf16 run: its contractions other than the offset conv stay on fp32 MFMA)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2, help="volumes per GPU (BASELINE.json: b2)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step in a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--extras", action="store_true", help="also report the full-net training iteration, the 2-D block images/s and the sliding-window tiles/s "
                    "(SURVEY §8d secondary metrics; ~1 min)")
    ap.add_argument("--no-companion", action="store_true", help="skip the same step measured with the other activation dtype (N = 1 only)")
    ap.add_argument("--no-tblock", action="store_true", help="skip the second metric (wrapper-block stack through nn.Module/autograd)")
    ap.add_argument("--cpu-sample", default="stage", choices=["stage", "tiny"])
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------
# per-op HIP-event timing of the stage-0 block (C=32, 32^3, B): finds the dominant kernel and its roofline fraction
# ----------------------------------------------------------------------------------------------------------------
def op_table(B, C, N, dtype_bytes):
    """(name, launches per step over the 21 blocks is derived separately) algorithmic flops / bytes for ONE launch at
    stage (C, N^3, B).  Bytes follow SURVEY §8d's rule: unique input bytes + output bytes, intermediates on-chip = 0."""
    n = N ** 3
    E = B * C * n
    Off = B * 81 * n
    s = dtype_bytes
    t = {}
    t["pointwise_fwd"] = (2 * C * E, 2 * E * s)
    t["dw5_fwd"] = (2 * 125 * E, 2 * E * s)
    t["dw7_fwd"] = (2 * 343 * E, 2 * E * s)
    t["offset_conv_fwd"] = (2 * 27 * C * 81 * B * n, (E + Off) * s)
    t["deform_fwd"] = (2 * 27 * C * C * B * n + 27 * B * n * (15 * C + 30), (2 * E + Off) * s)
    t["pointwise_bwd_data"] = (2 * C * E, 2 * E * s)
    t["pointwise_bwd_weight"] = (2 * C * E, 2 * E * s)
    t["dw5_bwd_data"] = (2 * 125 * E, 2 * E * s)
    t["dw5_bwd_weight"] = (2 * 125 * E, 2 * E * s)
    t["dw7_bwd_data"] = (2 * 343 * E, 2 * E * s)
    t["dw7_bwd_weight"] = (2 * 343 * E, 2 * E * s)
    t["offset_conv_bwd_data"] = (2 * 27 * C * 81 * B * n, (E + Off) * s)
    t["offset_conv_bwd_weight"] = (2 * 27 * C * 81 * B * n, (E + Off) * s)
    # grad_input (brick scatter): Col = G W_tap^T on the matrix cores + 8 corners x C multiply-adds per (voxel, tap);
    # reads grad_out, offsets, writes grad_input.   grad_offset (gather): Col again + 8 x C dot products + 3 x 8 blends;
    # reads x, grad_out, offsets, writes grad_offset.
    t["deform_bwd_input"] = (2 * 27 * C * C * B * n + 27 * B * n * (16 * C), (2 * E + Off) * s)
    t["deform_bwd_offset"] = (2 * 27 * C * C * B * n + 27 * B * n * (16 * C + 48), (2 * E + 2 * Off) * s)
    t["deform_bwd_weight"] = (2 * 27 * C * C * B * n + 27 * B * n * (15 * C + 30), (2 * E + Off) * s)
    return t


def time_ops(B, C, N, dtype, iters=10, only=None):
    """HIP-event timing (torch.cuda.Event on the current stream == the stream the C-ABI launches on) of every
    kernel-level op of one token-layout block, through the channels-last entry points the block itself uses."""
    from ctypes import byref
    from deformablelka_amd import _lib as L
    lib = L.get_lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    dt = L.DLKA_F32 if dtype == torch.float32 else L.DLKA_BF16
    st = L.stream_ptr(torch.empty(1, device=dev))
    g = torch.Generator().manual_seed(0)
    bf16 = dtype == torch.bfloat16
    mk = lambda *s: torch.randn(*s, generator=g).to(dev, dtype)              # activations in the run's storage type
    mkf = lambda *s: torch.randn(*s, generator=g).to(dev, torch.float32)     # offsets, parameters: always fp32
    x, go = mk(B, N, N, N, C), mk(B, N, N, N, C)                     # channels-last activations
    off, goff = mkf(B, 81, N, N, N), mkf(B, 81, N, N, N)             # planar offsets
    out, out_off = torch.empty(B, N, N, N, C, dtype=torch.float32, device=dev), torch.empty_like(off)   # (fp32-sized: also serves as fp32 grad_x)
    w_pw, w5, w7 = mkf(C, C, 1, 1, 1), mkf(C, 1, 5, 5, 5), mkf(C, 1, 7, 7, 7)
    w_off, w_dc = mkf(81, C, 3, 3, 3) * 0.02, mkf(C, C, 3, 3, 3) * 0.03
    b_c, b_81 = mkf(C), mkf(81)
    if bf16 and only is None:   # the per-op entry points carry bf16 activations for the deformable conv only (the dominant ops)
        only = ("deform_fwd", "deform_bwd_input", "deform_bwd_offset", "deform_bwd_weight")
    gw_pw, gw5, gw7, gw_off, gw_dc = (torch.empty_like(t) for t in (w_pw, w5, w7, w_off, w_dc))

    def geom(cout, k, p, d, grp):
        return L.ConvGeom(B, C, N, N, N, cout, k, k, k, 1, 1, 1, p, p, p, d, d, d, grp, 1, 64)

    G = {"pw": geom(C, 1, 0, 1, 1), "dw5": geom(C, 5, 2, 1, C), "dw7": geom(C, 7, 9, 3, C), "off": geom(81, 3, 1, 1, 1),
         "dcn": geom(C, 3, 1, 1, 1)}
    wsb = max([lib.dlka_conv3d_cl_workspace(byref(v), dt, 1) for v in G.values()] +
              [lib.dlka_deform_conv3d_cl_workspace(byref(G["dcn"]), dt, 1)])
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    P = L.ptr
    N0 = None

    def conv_fwd(key, w, b, inp, o, planar=0):
        return lambda: lib.dlka_conv3d_forward_cl(P(inp), P(w), P(b), P(o), planar, P(ws), wsb, byref(G[key]), dt, st)

    def conv_bwd(key, w, inp, gout, gx, gw, planar=0):
        return lambda: lib.dlka_conv3d_backward_cl(P(inp), P(w), P(gout), planar, P(gx), P(gw), P(N0), P(ws), wsb, byref(G[key]), dt, st)

    ops = {
        "pointwise_fwd": conv_fwd("pw", w_pw, b_c, x, out),
        "dw5_fwd": conv_fwd("dw5", w5, b_c, x, out),
        "dw7_fwd": conv_fwd("dw7", w7, b_c, x, out),
        "offset_conv_fwd": conv_fwd("off", w_off, b_81, x, out_off, 1),
        "deform_fwd": lambda: lib.dlka_deform_conv3d_forward_cl(P(x), P(off), P(w_dc), P(b_c), P(out), P(ws), wsb, byref(G["dcn"]), dt, st),
        "pointwise_bwd_data": conv_bwd("pw", w_pw, x, go, out, None),
        "pointwise_bwd_weight": conv_bwd("pw", w_pw, x, go, None, gw_pw),
        "dw5_bwd_data": conv_bwd("dw5", w5, x, go, out, None),
        "dw5_bwd_weight": conv_bwd("dw5", w5, x, go, None, gw5),
        "dw7_bwd_data": conv_bwd("dw7", w7, x, go, out, None),
        "dw7_bwd_weight": conv_bwd("dw7", w7, x, go, None, gw7),
        "offset_conv_bwd_data": conv_bwd("off", w_off, x, goff, out, None, 1),
        "offset_conv_bwd_weight": conv_bwd("off", w_off, x, goff, None, gw_off, 1),
        "deform_bwd_input": lambda: lib.dlka_deform_conv3d_backward_cl(P(x), P(off), P(w_dc), P(go), P(out), P(N0), P(N0), P(N0), P(ws), wsb, byref(G["dcn"]), dt, st),
        "deform_bwd_offset": lambda: lib.dlka_deform_conv3d_backward_cl(P(x), P(off), P(w_dc), P(go), P(N0), P(out_off), P(N0), P(N0), P(ws), wsb, byref(G["dcn"]), dt, st),
        "deform_bwd_weight": lambda: lib.dlka_deform_conv3d_backward_cl(P(x), P(off), P(w_dc), P(go), P(N0), P(N0), P(gw_dc), P(N0), P(ws), wsb, byref(G["dcn"]), dt, st),
    }
    res = {}
    for name, fn in ops.items():
        if only is not None and name not in only:
            continue
        rc = fn()
        if rc != 0:
            raise RuntimeError(f"{name}: dlka status {rc}")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / iters  # ms per launch
    return res


# launches of each op per stage-block fwd+bwd (3 pointwise convs per block)
OP_COUNT = {"pointwise_fwd": 3, "pointwise_bwd_data": 3, "pointwise_bwd_weight": 3}
# the HIP kernel that carries each op (rocprofv3 --kernel-trace name) and what else the op launches
OP_KERNEL = {
    "deform_bwd_input": ("dlka::cl_deform_gx_fx2_kernel<34, 10, 10>", "+ cl_prep_weight (5 us) + zero_fill (5 us) + cl_deform_gx_gather_kernel (15 us)"),
    "deform_bwd_offset": ("dlka::cl_deform_goff2_kernel<1>", "+ cl_prep_weight (5 us); inside the block this kernel also stores the samples for the weight gradient"),
    "deform_fwd": ("dlka::cl_deform_fwd_kernel<1>", "+ cl_prep_weight (5 us)"),
    "deform_bwd_weight": ("dlka::cl_wgrad_deform_kernel<3>", "+ cl_wgrad_reduce_kernel (13 us); the operator call gathers for itself — inside the block "
                          "cl_wgrad_samp_kernel<3> contracts the samples cl_deform_goff2_kernel stored (59 us, profiles/r03n_f32_stage0_block_kernel_stats.csv)"),
    "offset_conv_fwd": ("dlka::cl_igemm_kernel<0, 1, 3, 3>", "+ cl_prep_weight (5 us)"),
    "offset_conv_bwd_data": ("dlka::cl_conv_wave_kernel<2, 0, 1, 2, 2>", "+ cl_prep_weight (5 us)"),
    "offset_conv_bwd_weight": ("dlka::cl_wgrad_dense_kernel<1, 3, 3, true, true>", "+ cl_wgrad_reduce_kernel (9 us)"),
    "dw7_fwd": ("dlka::cl_dwconv_rows2_kernel<7, 3, 8>", "+ cl_dw_prep_weight (5 us)"),
    "dw7_bwd_data": ("dlka::cl_dwconv_rows2_kernel<7, 3, 8>", "+ cl_dw_prep_weight (5 us)"),
    "dw7_bwd_weight": ("dlka::cl_dwconv_wgrad2_kernel<7, 3, 8>", "+ zero_fill + cl_dw_unprep"),
}


def roofline_report(B, dtype):
    dbytes = 4 if dtype == torch.float32 else 2
    C, N = 32, 32  # stage 0 carries ~70% of the step's FLOPs
    ms = time_ops(B, C, N, dtype)
    tab = op_table(B, C, N, dbytes)
    peak_tf = PEAK_F32_TFLOPS   # the deformable conv's contractions run on the fp32-input MFMA in both storage modes
    ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
    rows = []
    for name, t in ms.items():
        fl, by = tab[name]
        rows.append((t * OP_COUNT.get(name, 1), name, t, fl, by))
    rows.sort(reverse=True)
    log("per-op HIP-event timing, stage 0 (C=32, 32^3, B=%d), ms per launch:" % B)
    for tot, name, t, fl, by in rows:
        log(f"  {name:26s} {t:9.4f} ms  x{OP_COUNT.get(name, 1)}  {fl / t / 1e9:9.2f} TFLOP/s  {by / t / 1e6:9.1f} GB/s")
    _, name, t, fl, by = rows[0]
    ai = fl / by
    if ai > ridge:
        ach, peak, unit, bound = fl / (t * 1e-3) / 1e12, peak_tf, "TFLOP/s", "mfma"
    else:
        ach, peak, unit, bound = by / (t * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s", "hbm"
    traffic = traffic_source = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (scripts/pmc_traffic.sh)
    if os.path.exists(pmc):
        try:
            blob = json.load(open(pmc))
            traffic = blob.get(name, {}).get("traffic_bytes_per_launch")
            traffic_source = "profiles/pmc_traffic.json (%s): committed rocprofv3 --pmc passes, NOT measured by this run" % blob.get("_meta", {}).get("round", "r02f")
        except Exception:
            traffic = None
    kern, extra = OP_KERNEL.get(name, (name, ""))
    return {"bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit, "frac": round(ach / peak, 5),
            "traffic": traffic, "traffic_source": traffic_source, "kernel": kern, "op": name, "op_also_launches": extra,
            "kernel_ms": round(t, 4), "algorithmic_flops": fl, "algorithmic_bytes": by,
            "shape": f"C={C},{N}^3,B={B}", "per_op_ms": {k: round(v, 4) for k, v in ms.items()}}


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle block on the host cores, bounded sample
# ----------------------------------------------------------------------------------------------------------------
def cpu_baseline(sample, batch, budget_s=40.0):
    """the oracle block (ATen CPU convs = the reference's own CPU path for nn.Conv3d/GELU, C oracle for the deformable conv,
    autograd backward) on the host cores — B = `batch`, fp32, offsets ~1 voxel, every one of the four stage shapes timed (no
    extrapolation between stages; the 21-block time is 6*t0 + 6*t1 + 6*t2 + 3*t3 of MEASURED per-stage medians).  Protocol per stage:
    warm-up + timed repetitions, median; BOUNDED: the plan's 3 warm + 10 timed shrink to what fits the time budget and the counts
    actually run are reported.  All host threads, then one thread (per-core figure) on the stages that still fit the budget."""
    import statistics
    import oracle
    from oracle import blocks
    import deformablelka_amd as dk
    from deformablelka_amd.stack import _offset_std_for
    oracle.build()
    cores = torch.get_num_threads()
    if sample == "tiny":
        stages = [(32, (8, 8, 8), 6), (64, (4, 4, 4), 6)]
    else:
        from deformablelka_amd.stack import SYNAPSE_STAGES
        stages = SYNAPSE_STAGES
    # untimed warm-up (thread pools, oneDNN primitive caches)
    _m = dk.LKA_Attention3d_deform(8)
    _P = {k: v.detach().clone().requires_grad_(True) for k, v in _m.state_dict().items()}
    blocks.lka3d_attention_volume(torch.randn(1, 8, 6, 6, 6, requires_grad=True), _P).sum().backward()

    def make(C, dims):
        torch.manual_seed(0)
        m = dk.LKA_Attention3d_deform(C)
        blocks.randomize_offsets_(m, std=_offset_std_for(C))
        P = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
        return P, torch.randn(batch, C, *dims, requires_grad=True), torch.randn(batch, C, *dims)

    def once(P, x, gy):
        t0 = time.perf_counter()
        blocks.lka3d_attention_volume(x, P).backward(gy)
        return time.perf_counter() - t0

    t_start = time.perf_counter()
    per_stage, reps = [], []
    share = budget_s * 0.7 / len(stages)
    for C, dims, nblk in reversed(list(stages)):        # small stages first: they always get their full repetitions
        P, x, gy = make(C, dims)
        t_stage = time.perf_counter()
        warm = 0
        while warm < 3 and (warm == 0 or time.perf_counter() - t_stage < share * 0.3):
            once(P, x, gy)
            warm += 1
        ts = []
        while len(ts) < 10 and (len(ts) < 1 or time.perf_counter() - t_stage < share):
            ts.append(once(P, x, gy))
        per_stage.insert(0, statistics.median(ts))
        reps.insert(0, (warm, len(ts)))
    total = sum(t * n for t, (_, _, n) in zip(per_stage, stages))
    # one thread
    one = {}
    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "1"
    try:
        import ctypes
        try:
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(1)    # the C oracle's OpenMP team
        except OSError:
            pass
        for (C, dims, nblk), t_all in reversed(list(zip(stages, per_stage))):
            if time.perf_counter() - t_start + t_all * cores * 0.5 > budget_s:   # would not fit: say so instead of extrapolating
                one[f"C{C}"] = None
                continue
            P, x, gy = make(C, dims)
            once(P, x, gy)
            one[f"C{C}"] = round(once(P, x, gy), 4)
    finally:
        torch.set_num_threads(cores)
        try:
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
        except Exception:
            pass
    return {"value": round(batch / total, 5), "unit": "volumes/s", "cores": cores, "kind": "port",
            "sample": (f"oracle D-LKA block fwd+bwd at B={batch}, fp32, offsets ~1 voxel, one block of EACH of the 4 stage shapes timed "
                       f"(warm-up, timed repetitions per stage C=32/64/128/256: {reps}; median), 21-block time = 6*t0+6*t1+6*t2+3*t3; "
                       f"bounded to ~{budget_s:.0f} s (BASELINE.md §3 asks 3 warm + 10 timed)" if sample != "tiny" else "TINY shapes (debug only)"),
            "per_stage_block_s": [round(t, 4) for t in per_stage], "one_thread_block_s": one, "torch": torch.__version__,
            "wall_s": round(time.perf_counter() - t_start, 1)}


def step_work(batch, dbytes):
    """Algorithmic FLOPs and bytes of ONE step (fwd+bwd of the 21 blocks at B = batch), SURVEY.md §8d: bytes fwd+bwd = 43 E + 5 Off
    words per block (activations in the run's storage type, offsets always fp32), FLOPs = 3 x forward."""
    from deformablelka_amd.stack import SYNAPSE_STAGES
    fl = by = 0
    for C, (H, W, D), n in SYNAPSE_STAGES:
        N = H * W * D
        E, Off = batch * C * N, batch * 81 * N
        fwd = 6 * C * E + 936 * E + 2 * 27 * C * 81 * batch * N + 2 * 27 * C * C * batch * N + 27 * batch * N * (15 * C + 30)
        fl += n * 3 * fwd
        by += n * (43 * E * dbytes + 5 * Off * 4)
    return fl, by


def tblock_metric(batch, steps, warmup, dev):
    """Second reported metric: the same 21 blocks INSIDE their wrapper (TransformerBlock_3D_single_deform_LKA: LayerNorm, gamma residual,
    UnetResBlock, conv8 — SURVEY.md §8 rows a1/f1), fwd+bwd through the nn.Module / autograd path, chained per stage instance."""
    import deformablelka_amd as dk
    from deformablelka_amd.stack import SYNAPSE_STAGES, CHAIN, _offset_std_for
    torch.manual_seed(0)
    chains = []
    for C, (H, W, D), n in SYNAPSE_STAGES:
        for c0 in range(0, n, CHAIN):
            mods = []
            for _ in range(min(CHAIN, n - c0)):
                m = dk.TransformerBlock_3D_single_deform_LKA(H * W * D, C, C, 4, dropout_rate=0.1, pos_embed=True)
                with torch.no_grad():
                    m.epa_block.spatial_gating_unit.deform_conv.conv_offset.weight.normal_(0, _offset_std_for(C))
                m.keep_channels_last = True
                mods.append(m.to(dev))
            x = torch.randn(batch, H, W, D, C, device=dev).permute(0, 4, 1, 2, 3).requires_grad_(True)
            gy = torch.randn(batch, H, W, D, C, device=dev).permute(0, 4, 1, 2, 3)
            chains.append((mods, x, gy))

    def step():
        for mods, x, gy in chains:
            y = x
            for m in mods:
                y = m(y)
            y.backward(gy)

    for _ in range(max(warmup, 2)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"metric": "3D D-LKA transformer-block (wrapper + D-LKA) fwd+bwd volumes/sec (64x128x128)", "value": round(batch / dt, 3),
            "unit": "volumes/s", "ms_per_step": round(dt * 1e3, 3), "path": "nn.Module + autograd, eager (no hipGraph), training mode",
            "blocks": sum(len(c[0]) for c in chains)}


def fullnet_metric(batch, steps, dev, bf16=False):
    """Third metric (SURVEY §8d iii / §8f-2): the WHOLE D_LKA_Former (42.35 M parameters; its 21 D-LKA transformer blocks on this repo's kernels,
    the conv / norm plumbing around them as GEMM re-expressions, HIP 3^3 convs and torch norms), one trainer iteration per step — forward, deep-supervision loss, backward,
    clip_grad_norm_(12), SGD(momentum 0.99, nesterov) — on a synthetic 64x128x128 patch batch (d_lka_former_trainer_synapse.py:259-309)."""
    from deformablelka_amd import training
    from deformablelka_amd.stack import _offset_std_for
    torch.manual_seed(0)
    net = training.initialize_network(1, 14, (64, 128, 128), device=dev)
    with torch.no_grad():
        for blk in net.dlka_blocks():
            w = blk.epa_block.spatial_gating_unit.deform_conv.conv_offset.weight
            w.normal_(0, _offset_std_for(w.shape[1]))
    opt = training.initialize_optimizer(net, initial_lr=1e-6)
    x = torch.randn(batch, 1, 64, 128, 128, device=dev)
    tgt = torch.randint(0, 14, (batch, 64, 128, 128), device=dev)
    net.train()
    for _ in range(2):
        training.run_iteration(net, opt, x, tgt, bf16_autocast=bf16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = training.run_iteration(net, opt, x, tgt, bf16_autocast=bf16)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if not bool(torch.isfinite(loss)):
        raise RuntimeError("full-net loss is not finite")
    return {"metric": "3D D-LKA Former full-net training iteration volumes/sec (64x128x128)", "value": round(batch / dt, 3), "unit": "volumes/s",
            "ms_per_step": round(dt * 1e3, 2), "params": sum(p.numel() for p in net.parameters()), "loss": round(float(loss), 4),
            "path": "nn.Module + autograd, eager; D-LKA blocks = HIP kernels; plumbing convs = GEMM re-expressions (rocBLAS) / HIP 3^3 convs, norms = torch" + ("; bf16 autocast policy" if bf16 else "")}


def lka2d_metric(steps, dev):
    """Secondary metric of SURVEY §8d: the 2-D D-LKA attention block fwd+bwd at B=24 on the three decoder shapes of the 224^2 net
    (2D/networks/MaxViT_deform_LKA.py:643-679), two blocks each — images/s through deformable_LKA_Attention."""
    import deformablelka_amd as dk
    from deformablelka_amd.init_utils import randomize_offset_nets
    shapes = [(384, 14), (192, 28), (96, 56)]
    torch.manual_seed(0)
    mods, xs, gys = [], [], []
    for C, n in shapes:
        for _ in range(2):
            m = dk.deformable_LKA_Attention(C).to(dev)
            randomize_offset_nets(m, 0.02)
            mods.append(m)
            xs.append(torch.randn(24, C, n, n, device=dev, requires_grad=True))
            gys.append(torch.randn(24, C, n, n, device=dev))

    def step():
        for m, x, gy in zip(mods, xs, gys):
            m(x).backward(gy)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    per = []
    for m, x, gy in zip(mods[::2], xs[::2], gys[::2]):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            m(x).backward(gy)
        e1.record()
        torch.cuda.synchronize()
        per.append(round(e0.elapsed_time(e1) / steps, 3))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"metric": "2D D-LKA attention blocks fwd+bwd images/sec (224x224 net, B=24: 2x(384,14^2)+2x(192,28^2)+2x(96,56^2))", "value": round(24 / dt, 2),
            "unit": "images/s", "ms_per_step": round(dt * 1e3, 2), "ms_per_block_fwd_bwd": dict(zip(["384x14^2", "192x28^2", "96x56^2"], per))}


def inference_metric(dev):
    """BASELINE.json config 5: pancreas-style sliding-window inference, 96^3 tiles, stride 16, on a synthetic (240, 240, 160) volume =
    10 x 10 x 5 = 500 tiles (test_util.py:73-75), the volume, score map and counts resident in HBM; tiles/s, forward only."""
    from deformablelka_amd import inference, training
    torch.manual_seed(0)
    net = training.initialize_network(1, 2, (96, 96, 96), device=dev, patch_size=(2, 2, 2)).eval()
    net.do_ds = False
    vol = torch.randn(240, 240, 160, device=dev)
    small = vol[:112, :112, :96].contiguous()          # warm-up: 2 x 2 x 1 tiles
    inference.predict_single_case(net, small, 16, 16, (96, 96, 96), num_classes=2, tile_batch=4)
    torch.cuda.synchronize()
    n = inference.num_tiles(vol.shape, (96, 96, 96), 16, 16)
    t0 = time.perf_counter()
    lab, score = inference.predict_single_case(net, vol, 16, 16, (96, 96, 96), num_classes=2, tile_batch=4)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert lab.shape == vol.shape and bool(torch.isfinite(score).all())
    return {"metric": "3D D-LKA Former sliding-window inference tiles/sec (96^3 tiles, stride 16, 240x240x160 volume resident in HBM)",
            "value": round(n / dt, 2), "unit": "tiles/s", "tiles": n, "seconds_per_volume": round(dt, 2), "tile_batch": 4}


def companion_metric(batch, steps, warmup, dev, dtype, lr):
    """The same stack step (fwd + bwd of the 21 blocks + SGD update, hipGraph replay) with the OTHER activation storage type — reported next to
    the headline so that one default run shows both: fp32 (the reference's arithmetic, 1e-4 parity) and bf16 activations (north_star's target
    dtype; parity against the bf16-storage oracle)."""
    from deformablelka_amd.stack import DLKABlockStack
    from deformablelka_amd import dp
    st = DLKABlockStack(batch, device=dev, dtype=dtype, seed=1234, data_seed=4321)
    st.forward_backward()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st.forward_backward()

    def step():
        dp.step_single(st, lr, 1, None, g.replay)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    h = st.health()
    if not h["finite"]:
        raise RuntimeError(f"non-finite parameters or gradients: {h}")
    return {"dtype": "bf16" if dtype == torch.bfloat16 else "f32", "value": round(batch * steps / el, 3), "unit": "volumes/s",
            "ms_per_step": round(el / steps * 1e3, 4), "steps": steps, "offset_std_voxels_by_stage": h["offset_std"]}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm; ranks talk over xGMI
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from deformablelka_amd.stack import DLKABlockStack
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    # replicas start from the same parameters (seed); every rank gets its own shard of synthetic volumes (data_seed)
    stack = DLKABlockStack(args.batch, device=dev, dtype=dtype, seed=1234, data_seed=4321 + rank)
    # Synthetic grad_outputs (N(0,1), no loss behind them) make the block gradients huge; a training-sized step would blow
    # the parameters up within a few iterations (offsets -> inf/NaN, every sample dropped, kernels get FASTER: observed,
    # profiles/r01i).  The SGD update is executed in full but with a step small enough that the data distribution the
    # kernels see (offset std ~ 1 voxel) is the same in the last timed step as in the first; checked after the run.
    lr = 1e-12

    def compute():
        stack.forward_backward()

    # N > 1: the step is cut where the backward pass has produced most of the gradient bytes (the C = 256 / 128 blocks come first in the
    # backward order and hold 84 % of them): their all-reduce runs on RCCL's stream while the second half of the backward pass computes.
    overlap = world > 1 or os.environ.get("DLKA_BENCH_FORCE_SPLIT") is not None
    if os.environ.get("DLKA_BENCH_NO_OVERLAP") is not None:
        overlap = False
    split = stack.split_index() if overlap else 0
    if split <= 0 or split >= len(stack.blocks):
        overlap = False
    cut = stack.grad_offset_of(split) if overlap else 0

    def compute_a():
        stack.forward()
        stack.backward(split, None)

    def compute_b():
        stack.backward(0, split)

    from deformablelka_amd import dp

    graph = None
    graph_a = graph_b = None
    # eager warm-up (also first-touch of every kernel), then capture
    compute()
    torch.cuda.synchronize()
    if not args.no_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                compute()
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(graph):
                compute()
        except Exception as e:  # capture is an optimisation, not a requirement
            log("hipGraph capture failed, running eagerly:", repr(e))
            graph = None
            torch.cuda.synchronize()
        if not dp.all_ranks_agree(graph is not None, dist, world, dev):   # every rank replays a graph, or none does
            graph = None

    def prepare_overlap():
        nonlocal graph_a, graph_b
        if args.no_graph or graph is None:
            return True            # eager halves need no preparation
        try:
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga):
                compute_a()
            with torch.cuda.graph(gb):
                compute_b()
            graph_a, graph_b = ga, gb
            return True
        except Exception as e:
            log("split capture failed:", repr(e))
            torch.cuda.synchronize()
            return False

    def run_a():
        (graph_a.replay() if graph_a is not None else compute_a())

    def run_b():
        (graph_b.replay() if graph_b is not None else compute_b())

    def trial_overlap():   # the two compute halves WITHOUT collectives
        try:
            run_a()
            run_b()
            torch.cuda.synchronize()
            return True
        except Exception as e:
            log("overlapped trial failed:", repr(e))
            torch.cuda.synchronize()
            return False

    # the ADVICE-r1 finding: every local decision is reduced over the ranks before anyone acts on it (deformablelka_amd/dp.py,
    # tests/test_dist_gloo.py::test_all_ranks_take_the_same_allreduce_schedule)
    schedule = dp.choose_schedule(overlap, prepare_overlap, trial_overlap, dist, world, dev)
    if schedule != "overlap":
        graph_a = graph_b = None

    def step_simple():
        dp.step_single(stack, lr, world, dist, graph.replay if graph is not None else compute)   # one flat all-reduce + SGD update

    def step_overlapped():
        dp.step_overlap(stack, lr, world, dist, run_a, run_b, cut)

    step = step_overlapped if schedule == "overlap" else step_simple

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = args.batch * world * args.steps / elapsed
    health = stack.health()   # finite parameters / gradients and the offset statistics of the LAST executed step
    if not health["finite"]:
        raise SystemExit(f"bench invalid: non-finite parameters or gradients after the timed steps: {health}")

    if rank == 0:
        out = {
            "metric": "3D D-LKA fwd+bwd volumes/sec (64x128x128, b2)", "value": round(value, 3), "unit": "volumes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "3D D-LKA Former Synapse 64x128x128 patch: fwd+bwd of its 21 D-LKA attention blocks "
                                   "(6x(32,32^3)+6x(64,16^3)+6x(128,8^3)+3x(256,4^3)) + grad all-reduce + SGD update",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "block_params": stack.num_params(), "hipgraph": graph is not None,
                       "allreduce_overlap": bool(step is step_overlapped), "allreduce_split_block": int(split),
                       "offset_std_voxels_stage0": health["offset_std"][0], "offset_std_voxels_by_stage": health["offset_std"]},
        }
        if not args.no_roofline:
            try:
                out["roofline"] = roofline_report(args.batch, dtype)
                fl, by = step_work(args.batch, 4 if dtype == torch.float32 else 2)
                t_step = ms_per_step * 1e-3
                out["roofline"]["step"] = {   # the WHOLE step against both roofs (north_star: achieved-HBM-bandwidth fraction)
                    "algorithmic_flops": fl, "algorithmic_bytes": by,
                    "hbm_frac": round(by / t_step / (PEAK_HBM_GBS * 1e9), 5), "achieved_GBps": round(by / t_step / 1e9, 1),
                    "fp32_frac": round(fl / t_step / (PEAK_F32_TFLOPS * 1e12), 5), "achieved_TFLOPs": round(fl / t_step / 1e12, 2),
                    "floor_ms_hbm": round(by / (PEAK_HBM_GBS * 1e9) * 1e3, 3), "floor_ms_fp32": round(fl / (PEAK_F32_TFLOPS * 1e12) * 1e3, 3)}
            except Exception as e:
                log("roofline timing failed:", repr(e))
                out["roofline"] = None
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.batch)
            except Exception as e:
                log("cpu baseline failed:", repr(e))
                out["cpu_baseline"] = None
        if dtype == torch.bfloat16:
            out["config"]["bf16"] = ("bf16 STORAGE of every activation tensor (x, y, saved, intermediate gradients); fp32 parameters, offsets, "
                                     "grad_offset and accumulation; offset-predict conv on single bf16 MFMA products, the other contractions on fp32 MFMA")
        if world == 1 and not args.no_companion:
            try:
                out["other_dtype"] = companion_metric(args.batch, args.steps, args.warmup, dev, torch.bfloat16 if dtype == torch.float32 else torch.float32, lr)
            except Exception as e:
                log("companion dtype measurement failed:", repr(e))
                out["other_dtype"] = None
            torch.cuda.empty_cache()
        if not args.no_tblock and world == 1 and dtype == torch.float32:
            try:
                out["tblock"] = tblock_metric(args.batch, max(3, args.steps // 2), 2, dev)
            except Exception as e:
                log("tblock metric failed:", repr(e))
                out["tblock"] = None
        if args.extras and world == 1:
            for key, fn in (("fullnet", lambda: fullnet_metric(args.batch, 5, dev, bf16=(dtype == torch.bfloat16))),
                            ("lka2d", lambda: lka2d_metric(5, dev)), ("inference", lambda: inference_metric(dev))):
                if dtype == torch.bfloat16 and key != "fullnet":
                    continue
                try:
                    out[key] = fn()
                except Exception as e:
                    log(f"{key} metric failed:", repr(e))
                    out[key] = None
                torch.cuda.empty_cache()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
