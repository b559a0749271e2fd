// Channels-last weight gradients on the matrix cores (fp32-input MFMA, exact fp32).
//
//     gW[co][ci][tap] = sum_m G[m][co] * A(m, tap, ci)            m = (b, voxel)
//   AMODE 0  A = in[b][voxel + tap offset][ci]  (zero padded)     1x1x1 projections, offset-predict conv
//   AMODE 1  A = trilinear sample                                  deformable conv; the reference recomputes the whole
//                                                                  im2col buffer for this (deform_conv_cuda.cu:254-261),
//                                                                  here a 32-row sample tile per tap lives in LDS only, or the
//                                                                  samples come stored from the grad_offset kernel (samp)
//   GMODE 0  G channels-last [M][Cout];  GMODE 1  G planar [B][Cout][N] (the offset tensor keeps the reference layout)
//
// One wave (64-thread workgroup) owns one 32(co) x 32(ci) output tile for TPW taps and walks a chunk of rows, 32 at a
// time: MFMA A operand = G^T (lane i = co), B operand = A(m,tap,ci) (lane j = ci), k = 32 rows per step pair.
// Partial sums per row-chunk go to a workspace and are folded by cl_wgrad_reduce_kernel (no same-address atomics).
#include <stdlib.h>

#include "deform_sample.h"
#include "cl_args.h"
#include "cl_gather.h"
#include "dlka_kernels.h"

namespace dlka {

// ---------------------------------------------------------------------------------------------------------------------
// Deformable weight gradient, second generation: the sample tile of each tap is gathered in the line-friendly layout of
// cl_gather.h (8 whole 128-byte rows per load instruction), interpolated, and written to a wave-private LDS tile from
// which the MFMA B operand is read; the corner loads of tap t+1 are in flight under the MFMAs of tap t.
//     gW[co][ci][tap] = sum_m G[m][co] * S(m, tap, ci)
// One wave = one 32(co) x 32(ci) tile x TPW taps over a chunk of rows.
// ---------------------------------------------------------------------------------------------------------------------
template <int TPW, typename T = float>   // T: storage of `in` and `g` (both channels-last)
__global__ __launch_bounds__(64) void cl_wgrad_deform_kernel(WgradArgs p)
{
    constexpr unsigned XB = sizeof(T);
    constexpr int SROW = 36;
    __shared__ __attribute__((aligned(16))) float Ssm[2][32 * SROW];
    __shared__ __attribute__((aligned(16))) float Dt[32 * GATHER_DESC_WORDS];
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    using GG = GatherGeom<T>;
    const int gr = lane >> GG::PSHIFT, gp = lane & ((1 << GG::PSHIFT) - 1);
    int chunk = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_total) {   // XCD-swizzled 1-D grid: an XCD owns a contiguous range of row chunks (with all their tiles and tap groups)
        const int r = xcd_item(blockIdx.x, p.xcd_total);
        if (r < 0) return;
        chunk = r / (p.xcd_ny * p.xcd_nz); by = (r / p.xcd_nz) % p.xcd_ny; bz = r % p.xcd_nz;
    }
    const int ot = by / p.CT, ct = by % p.CT;
    const int tap0 = bz * TPW;
    const int ntap = min(TPW, p.K - tap0);
    const int co = ot * 32 + i;
    const bool want_bias = p.bpart && ct == 0 && bz == 0;
    const BufRsrc rin = make_rsrc(p.in, (size_t)p.M * p.Cin * XB), rg = make_rsrc(p.g, (size_t)p.M * p.Cout * XB);
    const int HW = p.H * p.W, rowbytes = p.Cin * XB;
    const unsigned cbyte = (unsigned)(ct * 32 + GG::PE * gp) * XB;

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    const int m_lo = chunk * p.rows_per_chunk;
    const int m_hi = min(p.M, m_lo + p.rows_per_chunk);
    GatherPiece<T> xr[GG::NG][8];
    RowLook rd[GG::NG];
    for (int mbase = m_lo; mbase < m_hi; mbase += 32) {
        // A operand: G[m = mbase + 16h + s][co]
        float ga[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int mm = mbase + 16 * h + s;
            ga[s] = act_buf_load1<T>(rg, (mm < m_hi && co < p.Cout) ? (unsigned)(mm * p.Cout + co) * XB : DLKA_OOB);
        }
        // describing lane: row m = mbase + i
        const int m = mbase + i;
        const bool row_ok = m < m_hi;
        const int b = row_ok ? m / p.N : 0, v = row_ok ? m - b * p.N : 0;
        const int w0 = v % p.W, h0 = (v / p.W) % p.H, d0 = v / (p.W * p.H);
        auto describe_issue = [&](int t) {
            const int tap = tap0 + t;
            int ti_, tj_, tk_;
            tap_decode(tap, p.kw, p.kh, ti_, tj_, tk_);
            const int od = ti_ * p.dd - p.pd, oh = tj_ * p.dh - p.ph, ow = tk_ * p.dw - p.pw;
            wave_sync();
            if (h == 0) {
                RowDesc r;
                r.base = 0; r.okm = 0; r.ld = r.lh = r.lw = 0.f;
                if (row_ok) r = gather_describe(p.off + ((long)b * 3 * p.K + 3 * tap) * p.N + v, p.N, b, d0 + od, h0 + oh, w0 + ow, p.D, p.H, p.W);
                gather_publish(Dt, i, r, rowbytes);
            }
            wave_sync();
#pragma unroll
            for (int g = 0; g < GG::NG; ++g) {
                rd[g] = gather_lookup(Dt, GG::RPI * g + gr);
#pragma unroll
                for (int q = 0; q < 8; ++q) xr[g][q] = gather_load<T>(rin, gather_offset(rd[g], q, HW, p.W, rowbytes, cbyte));
            }
        };
        describe_issue(0);
        if (want_bias) {
#pragma unroll
            for (int s = 0; s < 16; ++s) bsum += ga[s];
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            if (t >= ntap) break;   // uniform
            float *S = Ssm[t & 1];
            // interpolate tap t and put its tile into LDS (the tile read two taps ago is free: LDS ops retire in order)
#pragma unroll
            for (int g = 0; g < GG::NG; ++g) {
                float wq[8];
                gather_weights(rd[g], wq);
#pragma unroll
                for (int v = 0; v < GG::PE / 4; ++v) {
                    f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const f32x4 x4 = xr[g][q].get(v);
                    s4[0] = fmaf(wq[q], x4[0], s4[0]); s4[1] = fmaf(wq[q], x4[1], s4[1]);
                        s4[2] = fmaf(wq[q], x4[2], s4[2]); s4[3] = fmaf(wq[q], x4[3], s4[3]);
                    }
                    *reinterpret_cast<f32x4 *>(S + (GG::RPI * g + gr) * SROW + GG::PE * gp + 4 * v) = s4;
                }
            }
            if (t + 1 < ntap) describe_issue(t + 1);   // its corner loads fly under the MFMAs below
            else wave_sync();
            const float *srow = S + (16 * h) * SROW + i;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc[t] = mfma_32x32x2(ga[s], srow[s * SROW], acc[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tap = tap0 + t;
        if (tap >= p.K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            p.part[(((long)chunk * p.K + tap) * p.CoutP + ot * 32 + row) * p.Cin + ct * 32 + i] = acc[t][r];
        }
    }
    if (want_bias) {
        bsum += __shfl_xor(bsum, 32);
        if (h == 0) p.bpart[(long)chunk * p.CoutP + co] = bsum;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Deformable weight gradient from STORED samples: cl_deform_goff2_kernel (which holds the 8 corners of every (row, tap) anyway)
// has written S[tap][m][ci] (WgradArgs::samp; fp32, or bf16 with bf16 activations), so the contraction  gW[co][ci][tap] = sum_m G[m][co] * S[tap][m][ci]  is a dense
// stream: per 32-row step a lane loads its 16 G elements and 16 S elements per tap (each wave load = two whole 128-byte rows), the next
// step's operands are in flight under the current step's 16 * TPW MFMAs.  Same tile / chunk / partial layout as cl_wgrad_deform_kernel,
// and the same arithmetic: S is produced by the same fma chain, the MFMA order over the rows is the same.
// ---------------------------------------------------------------------------------------------------------------------
// S16 (T = float only, round 6): the samples are IEEE halves (WgradArgs::samp_f16) — half the bytes this HBM-bound stream reads; grad_out, the products (fp32-input MFMA on the
// widened sample) and the accumulation stay fp32.
template <int TPW, typename T = float, bool S16 = false, bool B16M = false>   // T: storage of the channels-last `g`; B16M (S16 only): the contraction on the bf16 matrix cores
__global__ __launch_bounds__(64, 2) void cl_wgrad_samp_kernel(WgradArgs p)
{
    static_assert(!B16M || S16, "B16M is a variant of the half-sample kernel");
    static_assert(!S16 || sizeof(T) == 4, "half samples belong to the fp32 path");
    constexpr unsigned XB = sizeof(T);
    constexpr unsigned SB = S16 ? 2u : XB;   // bytes of a stored sample
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    int chunk = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_total) {
        const int r = xcd_item(blockIdx.x, p.xcd_total);
        if (r < 0) return;
        chunk = r / (p.xcd_ny * p.xcd_nz); by = (r / p.xcd_nz) % p.xcd_ny; bz = r % p.xcd_nz;
    }
    const int ot = by / p.CT, ct = by % p.CT;
    const int tap0 = bz * TPW;
    const int co = ot * 32 + i, ci = ct * 32 + i;
    const bool want_bias = p.bpart && ct == 0 && bz == 0;
    const BufRsrc rg = make_rsrc(p.g, (size_t)p.M * p.Cout * XB), rs = make_rsrc(p.samp, (size_t)p.K * p.M * p.Cin * SB);

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    const int m_lo = chunk * p.rows_per_chunk;
    const int m_hi = min(p.M, m_lo + p.rows_per_chunk);
    // register ring of RAW loaded words: any conversion (bf16 half -> float) happens at compute time — a conversion next to its load makes the
    // compiler wait for that load on the spot, which serialises the prefetch (measured: 96 instead of 60 us with 2-byte loads converted at once)
    float ga[2][16], sv[2][TPW][16];
    // one per-lane offset per operand (rows 16h.., column co / ci); the row step and the tap go through the wave-uniform offset
    // (bf16 grad_out / samples: a dword holding the column pair (c & ~1, c | 1); the lane keeps its half)
    constexpr bool G16 = sizeof(T) == 2;
    const unsigned vg = co < p.Cout ? (unsigned)(16 * h * p.Cout + (G16 ? (co & ~1) : co)) * XB : DLKA_OOB;
    const unsigned vs = (unsigned)(16 * h * p.Cin + ((G16 || S16) ? (ci & ~1) : ci)) * SB;   // (the samples have the storage type of grad_out, or are halves: S16)
    const unsigned gsh = (co & 1) ? 0u : 16u, ssh = (ci & 1) ? 0u : 16u;           // half selection: (word << sh) & 0xffff0000
    unsigned tapbit[TPW];   // 0, or the out-of-range bit for taps past K (uniform)
#pragma unroll
    for (int t = 0; t < TPW; ++t) tapbit[t] = tap0 + t < p.K ? 0u : DLKA_OOB;
    auto load_step = [&](int buf, int mbase) {
        const int lim = m_hi - mbase - 16 * h;   // rows of this half-wave inside the chunk (ragged only at m_hi == M)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const unsigned rowbit = s < lim ? 0u : DLKA_OOB;   // offsets are < 2^31: OR-ing the top bit sends the load out of range -> 0
            ga[buf][s] = buf_load_f32_s(rg, vg | rowbit, (unsigned)((mbase + s) * p.Cout) * XB);
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                sv[buf][t][s] = buf_load_f32_s(rs, vs | rowbit | tapbit[t], (unsigned)(((tap0 + t) * p.M + mbase + s) * p.Cin) * SB);
        }
    };
    auto half = [&](float w, unsigned sh) { return G16 ? __uint_as_float((__float_as_uint(w) << sh) & 0xffff0000u) : w; };
    // S16: the dword holds the halves of columns (ci & ~1, ci | 1); out-of-range loads return 0 = +0.0 in either format
    auto samp_val = [&](float w) { return S16 ? f16_value((unsigned short)(__float_as_uint(w) >> ((ci & 1) ? 16 : 0))) : half(w, ssh); };
    // bf16 storage (round 6): grad_out and the samples ARE bf16 — the halves the lane keeps of sixteen rows, packed pairwise, are the two K = 16 operands of
    // v_mfma_f32_32x32x16_bf16 (lane (i, h) supplies k = 8 h .. 8 h + 7 <-> rows 16 h + 8 kb + e of k-block kb: the same assignment on both operands): two instructions of 32
    // cycles per tap and 32-row step instead of sixteen fp32-input ones of 64, and no widening.  Same products (bf16 x bf16 is exact in fp32), fp32 accumulation; the sum over
    // the rows of a step is formed in the matrix core's order instead of row by row (rounding only).
    auto pack8 = [&](const float *w16, int kb, unsigned parity) {   // rows 8 kb .. 8 kb + 7 of this lane's column -> 8 bf16
        float o[4];
        const unsigned sh = parity ? 16u : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned w0 = __float_as_uint(w16[8 * kb + 2 * e]), w1 = __float_as_uint(w16[8 * kb + 2 * e + 1]);
            o[e] = __uint_as_float(((w0 >> sh) & 0xffffu) | ((w1 >> sh) << 16));
        }
        return bf16x8_from_words(o);
    };
    auto compute = [&](int buf) {
        if (G16) {
            if (want_bias) {
#pragma unroll
                for (int s = 0; s < 16; ++s) bsum += half(ga[buf][s], gsh);
            }
            const bf16x8 a0 = pack8(ga[buf], 0, (unsigned)(co & 1)), a1 = pack8(ga[buf], 1, (unsigned)(co & 1));
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                acc[t] = mfma_32x32x16_bf16(a0, pack8(sv[buf][t], 0, (unsigned)(ci & 1)), acc[t]);
                acc[t] = mfma_32x32x16_bf16(a1, pack8(sv[buf][t], 1, (unsigned)(ci & 1)), acc[t]);
            }
            return;
        }
        float g[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) g[s] = half(ga[buf][s], gsh);
        if (want_bias) {
#pragma unroll
            for (int s = 0; s < 16; ++s) bsum += g[s];
        }
        if (B16M) {   // fp32 grad_out x half samples on the bf16 matrix cores: both as two bf16 terms (a half's 11 significant bits are EXACTLY hi + lo)
            bf16x8 gh[2], gl[2];
            split_bf16x8(g, gh[0], gl[0]);
            split_bf16x8(g + 8, gh[1], gl[1]);
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                float sf[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) sf[s] = samp_val(sv[buf][t][s]);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    bf16x8 sh, sl;
                    split_bf16x8(sf + 8 * kb, sh, sl);
                    acc[t] = mfma_32x32x16_bf16(gl[kb], sh, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(gh[kb], sl, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(gh[kb], sh, acc[t]);
                }
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int s = 0; s < 16; ++s) acc[t] = mfma_32x32x2(g[s], samp_val(sv[buf][t][s]), acc[t]);
    };
    if (m_lo < m_hi) load_step(0, m_lo);
    for (int mbase = m_lo; mbase < m_hi; mbase += 64) {   // two steps per trip: the register buffers are addressed statically
        if (mbase + 32 < m_hi) load_step(1, mbase + 32);
        compute(0);
        if (mbase + 32 < m_hi) {
            if (mbase + 64 < m_hi) load_step(0, mbase + 64);
            compute(1);
        }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tap = tap0 + t;
        if (tap >= p.K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            p.part[(((long)chunk * p.K + tap) * p.CoutP + ot * 32 + row) * p.Cin + ct * 32 + i] = acc[t][r];
        }
    }
    if (want_bias) {
        bsum += __shfl_xor(bsum, 32);
        if (h == 0) p.bpart[(long)chunk * p.CoutP + co] = bsum;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense (plain-neighbour) weight gradient, second generation.  One wave owns COT co-tiles x TPW taps x one 32-channel
// ci-tile: the B operand (input rows of a tap) is loaded once and feeds COT MFMA chains, the A operand (grad_out rows of
// a co-tile) once and feeds TPW chains -> (COT + TPW) * 16 loads per COT * TPW * 16 MFMAs per 32-row step (3x3: 96 / 144
// instead of the first version's 80 / 64).  The next step's operands are in flight while the current step's MFMAs issue.
// ---------------------------------------------------------------------------------------------------------------------
// SPLIT: bf16 x3-split contraction (dlka_intrin.h) instead of the exact fp32-input MFMA: 54 x 32 cycles per 32-row step instead
// of 144 x 64 for the 3 x 3 blocking.
// T: storage of `in` and of a channels-last `g` (GMODE 0); a planar `g` (GMODE 1: grad_offset) is always fp32.  A bf16 `in` is its own high
// term, so the split contraction drops the b_lo product.
template <int GMODE, int COT, int TPW, bool N16, bool SPLIT, typename T, bool PAD = false>   // N16: N % 16 == 0 (the 16 rows of a half-wave never straddle two volumes)
__device__ __forceinline__ void wgrad_dense_body(const WgradArgs &p, const int bz_in)
{
    constexpr unsigned XB = sizeof(T), GB = GMODE == 0 ? sizeof(T) : 4u;
    constexpr bool B16 = sizeof(T) == 2;
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    int chunk = blockIdx.x, by = blockIdx.y, bz = bz_in;
    if (p.xcd_total) {   // XCD-swizzled 1-D grid (see cl_wgrad_deform_kernel)
        const int r = xcd_item(blockIdx.x, p.xcd_total);
        if (r < 0) return;
        chunk = r / (p.xcd_ny * p.xcd_nz); by = (r / p.xcd_nz) % p.xcd_ny; bz = r % p.xcd_nz;
    }
    const int OTG = cdiv(p.CoutP / 32, COT);                  // co-tile groups
    const int otg = by / p.CT, ct = by % p.CT;
    const int tap0 = bz * TPW;
    const int ci = ct * 32 + i;
    const bool want_bias = p.bpart && ct == 0 && bz == 0;
    (void)OTG;

    f32x16 acc[COT][TPW];
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;
    float bsum[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) bsum[c] = 0.f;

    int od[TPW], oh[TPW], ow[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tap = tap0 + t;
        od[t] = (tap / (p.kw * p.kh)) * p.dd - p.pd;
        oh[t] = ((tap / p.kw) % p.kh) * p.dh - p.ph;
        ow[t] = (tap % p.kw) * p.dw - p.pw;
    }
    const int m_lo = chunk * p.rows_per_chunk;
    const int m_hi = min(p.M, m_lo + p.rows_per_chunk);

    float ga_n[COT][16], bv_n[TPW][16];
    // TPW == kw == 3 with the fast row addressing below: the wave's three taps are the three w-neighbours of one (kd, kh), so their 3 x 16 rows are 18 DISTINCT
    // rows (w_ - 1 .. w_ + 16) — loaded once into xr_n, tap t's operand row s is xr[s + t].  (Ablation, profiles/r06_notes.md: 43 of this kernel's 79 us at
    // 32^3 were its operand fetch, 48 of the 60 loads per step these rows.)
    constexpr bool WIN3 = TPW == 3 && N16 && !PAD;
    float xr_n[WIN3 ? 18 : 1];
    const bool win3 = WIN3 && p.K > 1 && p.w16 && p.kw == 3 && p.dw == 1 && p.pw == 1 && p.no_win3 == 0;
    const int gcp = (GMODE == 1 && p.g_cpad) ? p.g_cpad : p.Cout;   // channel planes per batch of a planar g
    const BufRsrc rg = make_rsrc(p.g, (size_t)p.B * p.N * gcp * GB);
    const BufRsrc rx = PAD ? make_rsrc(p.pad, (size_t)p.B * p.DP * p.HP * p.WP * p.Cin * XB) : make_rsrc(p.in, (size_t)p.M * p.Cin * XB);
    // PAD (round 5): the B rows come from the zero-padded copy.  Row (b, d, h, w) of tap (i, j, k) is element ((b DP + d + i dd) HP + h + j dh) WP + w + k dw of it —
    // the VOXEL part is a per-row register, the TAP part a wave-uniform scalar that rides in the load's soffset: no coordinate test, no select, no address add per
    // (tap, row).  The general path below spends ~12 vector instructions on each of them, 576 of a step's ~930 at the 2-D net's 7 x 7 dilation-3 offset net — and
    // with one wave per SIMD (370 - 500 registers) the kernel is bound by its instruction count (ISA mix; 2.3 us per 32-row step for 24 MFMAs).
    unsigned soff[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tap = min(tap0 + t, p.K - 1);   // (a tap past K: any in-range address — its tile is never stored)
        const int ti = tap / (p.kw * p.kh), tj = (tap / p.kw) % p.kh, tk = tap % p.kw;
        soff[t] = PAD ? (unsigned)(((ti * p.dd * p.HP + tj * p.dh) * p.WP + tk * p.dw) * p.Cin) * XB : 0u;
    }
    // operands of the 32-row step starting at mbase: rows m = mbase + 16h + s.  Every load is an unconditional buffer
    // load; rows beyond the chunk, channels beyond Cout and zero-padded neighbours read offset DLKA_OOB -> 0.
    auto load_step = [&](int mbase) {
        const int mrow0 = mbase + 16 * h;
        const int b0 = mrow0 / p.N, v0 = mrow0 - b0 * p.N;
#pragma unroll
        for (int c = 0; c < COT; ++c) {
            const int co = (otg * COT + c) * 32 + i;
            if (GMODE == 0) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int m = mrow0 + s;
                    ga_n[c][s] = act_buf_load1<T>(rg, (m < m_hi && co < p.Cout) ? (unsigned)(m * p.Cout + co) * XB : DLKA_OOB);
                }
            } else if (N16) {   // 16 consecutive voxels of one plane, 64-byte aligned: four 16-byte loads
                const unsigned off = (mrow0 < m_hi && co < p.Cout) ? (unsigned)((b0 * gcp + co) * p.N + v0) * 4u : DLKA_OOB;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 t4 = buf_load_f32x4(rg, off + 16u * e);
                    ga_n[c][4 * e] = t4[0]; ga_n[c][4 * e + 1] = t4[1]; ga_n[c][4 * e + 2] = t4[2]; ga_n[c][4 * e + 3] = t4[3];
                }
            } else if ((p.N & 3) == 0) {   // N % 4 == 0 (round 5: the 2-D net's 14 x 14 stage, N = 196): the half-wave's 16 rows start at a multiple of 16, so every
                // run of 4 rows is 16-byte aligned inside ONE plane — four 16-byte loads as above, only that a run may belong to the next volume.  (The
                // per-row dword loads this replaces were 32 of the step's 80 load instructions, each touching 64 cache lines.)
                int b = b0, v = v0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 t4 = buf_load_f32x4(rg, (mrow0 + 4 * e < m_hi && co < p.Cout) ? (unsigned)((b * gcp + co) * p.N + v) * 4u : DLKA_OOB);
                    ga_n[c][4 * e] = t4[0]; ga_n[c][4 * e + 1] = t4[1]; ga_n[c][4 * e + 2] = t4[2]; ga_n[c][4 * e + 3] = t4[3];
                    v += 4;
                    if (v >= p.N) { v = 0; ++b; }
                }
            } else {   // any N: (b, v) walked from the half-wave's first row (one division per step, not one per row)
                int b = b0, v = v0;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    ga_n[c][s] = buf_load_f32(rg, (mrow0 + s < m_hi && co < p.Cout) ? (unsigned)((b * gcp + co) * p.N + v) * 4u : DLKA_OOB);
                    ++v;
                    if (v == p.N) { v = 0; ++b; }
                }
            }
        }
        if (PAD) {
            const int w_0 = v0 % p.W, hh_0 = (v0 / p.W) % p.H, d_0 = v0 / (p.W * p.H);
            const unsigned base0 = (unsigned)((((b0 * p.DP + d_0) * p.HP + hh_0) * p.WP + w_0) * p.Cin + ci) * XB;
            const unsigned rs = (unsigned)p.Cin * XB;
            const unsigned jw = (unsigned)(p.WP - p.W) * rs, jh = (unsigned)((p.HP - p.H) * p.WP) * rs, jd = (unsigned)((p.DP - p.D) * p.HP * p.WP) * rs;
            if (p.W >= 16) {   // uniform.  At most ONE w-row ends inside the half-wave's 16 rows: rows from `sw` on skip the halo once (and with it, when that row also
                // ends a plane / a volume, the plane's / volume's halo) — one compare and two adds per row, no divergent control flow
                const int sw = p.W - w_0;
                const bool eh = hh_0 == p.H - 1, ed = d_0 == p.D - 1;
                const unsigned J = jw + (eh ? jh + (ed ? jd : 0u) : 0u);
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const unsigned b_ = base0 + (unsigned)s * rs + (s >= sw ? J : 0u);
                    const unsigned vo = (mrow0 + s < m_hi) ? b_ : DLKA_OOB;
#pragma unroll
                    for (int t = 0; t < TPW; ++t) bv_n[t][s] = act_buf_load1_s<T>(rx, vo, soff[t]);
                }
            } else {   // narrow volumes: the general walk, written with selects (the `if` form compiles to a divergent branch per row)
                unsigned base = base0;
                int w_ = w_0, hh = hh_0, d_ = d_0;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const unsigned vo = (mrow0 + s < m_hi) ? base : DLKA_OOB;
#pragma unroll
                    for (int t = 0; t < TPW; ++t) bv_n[t][s] = act_buf_load1_s<T>(rx, vo, soff[t]);
                    ++w_;
                    const bool cw = w_ == p.W;
                    w_ = cw ? 0 : w_;
                    hh += cw ? 1 : 0;
                    const bool ch = hh == p.H;
                    hh = ch ? 0 : hh;
                    d_ += ch ? 1 : 0;
                    const bool cd = d_ == p.D;
                    d_ = cd ? 0 : d_;
                    base += rs + (cw ? jw : 0u) + (ch ? jh : 0u) + (cd ? jd : 0u);
                }
            }
        } else if (p.K == 1) {   // pointwise: the B rows are the A rows
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int m = mrow0 + s;
                bv_n[0][s] = act_buf_load1<T>(rx, (m < m_hi) ? (unsigned)(m * p.Cin + ci) * XB : DLKA_OOB);
            }
        } else if (N16 && p.w16 && p.kw == 3 && p.dw == 1) {
            // Fast addressing (W % 16 == 0, 3-wide taps): the 16 voxels of a half-wave are one aligned run of a W-row, so (d, h) — and with
            // them the whole tap's validity in d and h — are common to the 16 rows, only w = w_ + s moves, and only s = 0 / s = 15 can
            // step over the row's ends.  One select per tap, two edge selects and one add per row instead of ~12 VALU instructions per
            // (tap, row).  (The general decode below was half of this kernel's VALU time,
            // and the kernel is VALU-bound: matrix cores 17 % busy, profiles/archive/r01u_pmc_offc.txt.)
            const int w_ = v0 % p.W, hh = (v0 / p.W) % p.H, d_ = v0 / (p.W * p.H);
            const unsigned base = (mrow0 < m_hi) ? (unsigned)((b0 * p.N + v0) * p.Cin + ci) * XB : DLKA_OOB;
            const unsigned rs = (unsigned)p.Cin * XB;
            if (WIN3 && win3) {   // uniform
                const int zd = d_ + od[1], zh = hh + oh[1];   // (the three taps share kd, kh; tap 1 is the centre column)
                const bool ok = (tap0 + 2 < p.K) & (base != DLKA_OOB) & ((unsigned)zd < (unsigned)p.D) & ((unsigned)zh < (unsigned)p.H);
                const int doff = ((od[1] * p.H + oh[1]) * p.W) * p.Cin * (int)XB;
                const unsigned tc = ok ? base + (unsigned)doff : DLKA_OOB;
                xr_n[0] = act_buf_load1<T>(rx, (tc != DLKA_OOB && w_ >= 1) ? tc - rs : DLKA_OOB);
#pragma unroll
                for (int s = 0; s < 16; ++s) xr_n[1 + s] = act_buf_load1<T>(rx, tc == DLKA_OOB ? DLKA_OOB : tc + (unsigned)s * rs);
                xr_n[WIN3 ? 17 : 0] = act_buf_load1<T>(rx, (tc != DLKA_OOB && w_ + 16 < p.W) ? tc + 16u * rs : DLKA_OOB);
            } else
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int zd = d_ + od[t], zh = hh + oh[t];
                const bool ok = (tap0 + t < p.K) & (base != DLKA_OOB) & ((unsigned)zd < (unsigned)p.D) & ((unsigned)zh < (unsigned)p.H);
                const int doff = ((od[t] * p.H + oh[t]) * p.W + ow[t]) * p.Cin * (int)XB;
                const unsigned tb = ok ? base + (unsigned)doff : DLKA_OOB;
                const unsigned tb0 = (w_ + ow[t] >= 0) ? tb : DLKA_OOB, tb15 = (w_ + 15 + ow[t] < p.W) ? tb : DLKA_OOB;
                // (the row stride is added in the vector offset, not the scalar one: the hardware range-checks the vector offset alone, and
                //  tb itself can lie one element before the buffer when the run starts the tensor and the tap looks left)
                bv_n[t][0] = act_buf_load1<T>(rx, tb0);
#pragma unroll
                for (int s = 1; s < 15; ++s) bv_n[t][s] = act_buf_load1<T>(rx, tb + (unsigned)s * rs);
                bv_n[t][15] = act_buf_load1<T>(rx, tb15 + 15u * rs);
            }
        } else {
            int crd[16];
            unsigned rowoff[16];
            if (N16) {
                // 16 consecutive voxels of one volume: decode the first (three runtime divisions, ~60 VALU instructions),
                // walk the rest with carries — the per-row div/mod chain was most of this kernel's VALU time
                int w_ = v0 % p.W, hh = (v0 / p.W) % p.H, d_ = v0 / (p.W * p.H);
                const unsigned base = (unsigned)((b0 * p.N + v0) * p.Cin + ci) * XB;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    crd[s] = (mrow0 + s < m_hi) ? ((d_ << 20) | (hh << 10) | w_) : -1;
                    rowoff[s] = base + (unsigned)(s * p.Cin) * XB;
                    ++w_;
                    if (w_ == p.W) { w_ = 0; ++hh; if (hh == p.H) { hh = 0; ++d_; } }
                }
            } else {
                // any N (round 5; the 2-D net's 14 x 14 stage is N = 196): the same walk — rows are consecutive in memory across volumes too (x is [M][Cin]),
                // only the coordinates start over at a volume's end.  The per-row div / mod chain this replaces made that stage's offset-net weight
                // gradient 5x slower per FLOP than the 28 x 28 and 56 x 56 stages' (712 us against ~140 at their rate, BENCH_r04 lka2d table).
                int w_ = v0 % p.W, hh = (v0 / p.W) % p.H, d_ = v0 / (p.W * p.H);
                const unsigned base = (unsigned)(mrow0 * p.Cin + ci) * XB;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    crd[s] = (mrow0 + s < m_hi) ? ((d_ << 20) | (hh << 10) | w_) : -1;
                    rowoff[s] = base + (unsigned)(s * p.Cin) * XB;
                    ++w_;
                    if (w_ == p.W) { w_ = 0; ++hh; if (hh == p.H) { hh = 0; ++d_; if (d_ == p.D) d_ = 0; } }
                }
            }
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const bool tap_ok = tap0 + t < p.K;  // uniform
                const int doff = ((od[t] * p.H + oh[t]) * p.W + ow[t]) * p.Cin * (int)XB;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int c_ = crd[s];
                    const int zd = (c_ >> 20) + od[t], zh = ((c_ >> 10) & 1023) + oh[t], zw = (c_ & 1023) + ow[t];
                    // bitwise, not short-circuit: the compiler turns && chains into a branch per element
                    const bool ok = tap_ok & (c_ >= 0) & ((unsigned)zd < (unsigned)p.D) & ((unsigned)zh < (unsigned)p.H) & ((unsigned)zw < (unsigned)p.W);
                    bv_n[t][s] = act_buf_load1<T>(rx, ok ? rowoff[s] + (unsigned)doff : DLKA_OOB);
                }
            }
        }
    };

#ifndef DLKA_ABLG   // -DDLKA_ABLG=bits: TIMING-ONLY ablations of the split dense weight gradient (wrong results): 1 no split arithmetic, 2 one MFMA per
#define DLKA_ABLG 0  // (co-tile, tap) and k-half instead of three, 4 no operand fetch in the loop
#endif
    constexpr int ABLG = (SPLIT && GMODE == 1 && COT == 3 && TPW == 3) ? DLKA_ABLG : 0;
    if (m_lo < m_hi) load_step(m_lo);
    for (int mbase = m_lo; mbase < m_hi; mbase += 32) {
        float ga[COT][16], bv[TPW][16];
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int s = 0; s < 16; ++s) ga[c][s] = ga_n[c][s];
        if (WIN3 && win3) {   // uniform
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int s = 0; s < 16; ++s) bv[t][s] = xr_n[WIN3 ? s + t : 0];
        } else {
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int s = 0; s < 16; ++s) bv[t][s] = bv_n[t][s];
        }
        if (mbase + 32 < m_hi && !(ABLG & 4)) load_step(mbase + 32);   // in flight under the MFMAs below
        const bool gpk = GMODE == 1 && SPLIT && p.g_cpad;   // uniform: g arrives as pack_split2() words
        if (want_bias) {
#pragma unroll
            for (int c = 0; c < COT; ++c)
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if (gpk) {
                        const unsigned u = __builtin_bit_cast(unsigned, ga[c][s]);
                        bsum[c] += __builtin_bit_cast(float, u & 0xffff0000u) + __builtin_bit_cast(float, u << 16);
                    } else bsum[c] += ga[c][s];
                }
        }
        if (SPLIT) {
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {   // k = rows 16h + 8mf + e
                bf16x8 ahi[COT], alo[COT], bhi[TPW], blo[TPW];
#pragma unroll
                for (int c = 0; c < COT; ++c) {
                    if (gpk) unpack_split2x8(ga[c] + 8 * mf, ahi[c], alo[c]);
                    else if (ABLG & 1) { ahi[c] = bf16x8_from_words(ga[c] + 8 * mf); alo[c] = bf16x8_from_words(ga[c] + 8 * mf + 4); }
                    else split_bf16x8(ga[c] + 8 * mf, ahi[c], alo[c]);
                }
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    if (ABLG & 1) { bhi[t] = bf16x8_from_words(bv[t] + 8 * mf); blo[t] = bf16x8_from_words(bv[t] + 8 * mf + 4); }
                    else split_bf16x8(bv[t] + 8 * mf, bhi[t], blo[t]);
                }
#pragma unroll
                for (int c = 0; c < COT; ++c)
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        acc[c][t] = mfma_32x32x16_bf16(alo[c], bhi[t], acc[c][t]);
                        if (ABLG & 2) continue;
                        if (!B16) acc[c][t] = mfma_32x32x16_bf16(ahi[c], blo[t], acc[c][t]);   // a bf16 `in` has no low term
                        acc[c][t] = mfma_32x32x16_bf16(ahi[c], bhi[t], acc[c][t]);
                    }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int c = 0; c < COT; ++c)
#pragma unroll
                    for (int t = 0; t < TPW; ++t) acc[c][t] = mfma_32x32x2(ga[c][s], bv[t][s], acc[c][t]);
        }
    }
    // ---- partial tiles out: D row = co_local, col = ci_local ----
#pragma unroll
    for (int c = 0; c < COT; ++c) {
        const int ot = otg * COT + c;
        if (ot * 32 >= p.CoutP) continue;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tap = tap0 + t;
            if (tap >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                p.part[(((long)chunk * p.K + tap) * p.CoutP + ot * 32 + row) * p.Cin + ci] = acc[c][t][r];
            }
        }
        if (want_bias) {
            float bs = bsum[c];
            bs += __shfl_xor(bs, 32);   // the two halves hold the same co for different rows
            if (h == 0) p.bpart[(long)chunk * p.CoutP + ot * 32 + i] = bs;
        }
    }
}

template <int GMODE, int COT, int TPW, bool N16, bool SPLIT = false, typename T = float>
__global__ __launch_bounds__(64) void cl_wgrad_dense_kernel(WgradArgs p)
{
    wgrad_dense_body<GMODE, COT, TPW, N16, SPLIT, T>(p, blockIdx.z);
}

// the same from the zero-padded copy of the input (WgradArgs::pad): planar fp32 grad_out, split contraction, three taps per wave
template <int COT, bool N16, typename T>
__global__ __launch_bounds__(64) void cl_wgrad_dense_pad_kernel(WgradArgs p)
{
    wgrad_dense_body<1, COT, 3, N16, true, T, true>(p, blockIdx.z);
}

// out[b][d + lo_d][h + lo_h][w + lo_w][c] = in[b][d][h][w][c], zeros around.  One workgroup per OUTPUT w-row (b, dp, hp): the row's coordinates are decoded once
// (scalar), the threads stream its WP * C * sizeof(T) bytes in 16-byte pieces.  (The first version decoded every piece with three 64-bit divisions: 192 us for the
// 19 MB of the (384, 14^2, B = 24) copy, profiles/r08_notes.md.)
template <typename T>
__global__ __launch_bounds__(256) void cl_pad_copy_kernel(const T *__restrict__ in, T *__restrict__ out, int B, int D, int H, int W, int C, int DP, int HP, int WP,
                                                          int lo_d, int lo_h, int lo_w)
{
    constexpr int PE = 16 / sizeof(T);   // elements per 16 bytes
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    const int pieces = C / PE, rowp = WP * pieces;
    const int nrows = B * DP * HP;
    for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
        const int hp = r % HP, t = r / HP;
        const int dp = t % DP, b = t / DP;
        const int hh = hp - lo_h, d = dp - lo_d;
        const bool row_in = (unsigned)hh < (unsigned)H && (unsigned)d < (unsigned)D;
        const T *src = in + ((((long)b * D + (row_in ? d : 0)) * H + (row_in ? hh : 0)) * W) * C;
        u4_t *dst = reinterpret_cast<u4_t *>(out + (long)r * WP * C);
        const int lo = lo_w * pieces, hi = (lo_w + W) * pieces;   // pieces [lo, hi) of the row are the input's
        for (int q = threadIdx.x; q < rowp; q += 256) {
            u4_t val = {0u, 0u, 0u, 0u};
            if (row_in && q >= lo && q < hi) val = *reinterpret_cast<const u4_t *>(src + (long)(q - lo) * PE);
            dst[q] = val;
        }
    }
}

// The three pointwise weight gradients of a D-LKA block (proj_2, conv1, proj_1: same geometry, different operands) in ONE
// launch, blockIdx.z = job: every dependent kernel node costs ~4.5 us inside the graph, and each of these is a ~1 us kernel.
struct WgradArgs3 { WgradArgs a[3]; };
template <int COT, bool N16, typename T = float>
__global__ __launch_bounds__(64) void cl_wgrad_pw3_kernel(WgradArgs3 b)
{
    wgrad_dense_body<0, COT, 1, N16, false, T>(b.a[blockIdx.z], 0);
}

// gW[co][ci][tap] (reference layout, storage type T) = sum_chunk part[chunk][tap][co][ci];  gb[co] = sum_chunk bpart[chunk][co]
// A workgroup folds 32 consecutive outputs: thread (e = tid & 31, cl = tid >> 5) sums chunks cl, cl+8, ... (coalesced
// 128-byte reads per chunk), the 8 partial sums meet in LDS.  (The first version gave one thread all 128 chunks of an
// output: 40 us for the 1024 outputs of a pointwise conv, pure dependent-load latency; profiles/archive/r01e.)
template <typename T>
__global__ __launch_bounds__(256) void cl_wgrad_reduce_kernel(const float *__restrict__ part, const float *__restrict__ bpart, T *__restrict__ gw, T *__restrict__ gb,
                                                              int chunks, int K, int CoutP, int Cout, int Cin)
{
    __shared__ float red[8][33];
    const long n = (long)K * Cout * Cin, ntot = n + (gb ? Cout : 0);
    const long stride = (long)K * CoutP * Cin;
    const int el = threadIdx.x & 31, cl = threadIdx.x >> 5;
    for (long base = (long)blockIdx.x * 32; base < ntot; base += (long)gridDim.x * 32) {
        const long e = base + el;
        float a0 = 0.f, a1 = 0.f;
        int ci = 0, co = 0, tap = 0;
        if (e < n) {
            ci = (int)(e % Cin); co = (int)((e / Cin) % Cout); tap = (int)(e / Cin / Cout);
            const float *src = part + ((long)tap * CoutP + co) * Cin + ci;
            int c = cl;
            for (; c + 8 < chunks; c += 16) { a0 += src[(long)c * stride]; a1 += src[(long)(c + 8) * stride]; }
            if (c < chunks) a0 += src[(long)c * stride];
        } else if (e < ntot) {
            co = (int)(e - n);
            for (int c = cl; c < chunks; c += 8) a0 += bpart[(long)c * CoutP + co];
        }
        red[cl][el] = a0 + a1;
        __syncthreads();
        if (cl == 0 && e < ntot) {
            const float t = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
            if (e < n) stf(gw + ((long)co * Cin + ci) * K + tap, t);
            else stf(gb + co, t);
        }
        __syncthreads();
    }
}

// Work decomposition.  Waves = chunks x (co-tile groups x ci-tiles) x tap groups; aim at ~2 waves per SIMD over the chip
// so that small outputs (a 32x32 pointwise gradient is ONE tile) still get their parallelism from the row dimension.
struct WgradPlan { int chunks, tpw, cot; };

static WgradPlan wgrad_plan(int M, int K, int Cout, int Cin, int amode)
{
    WgradPlan pl;
    const int OT = round_up(Cout, 32) / 32, CT = Cin / 32;
    if (amode == 1) {
        pl.tpw = 3; pl.cot = 1;   // measured best (4 and 7 taps per wave: more registers, fewer waves)
    }
    else if (K == 1) { pl.tpw = 1; pl.cot = (OT % 2 == 0) ? 2 : 1; }
    else {
        static int tpw_env = -1;
        if (tpw_env < 0) tpw_env = 3;
        pl.tpw = tpw_env; pl.cot = (OT % 3 == 0) ? 3 : ((OT % 2 == 0) ? 2 : 1);
        static const int cot_cap = [] { const char *e = getenv("DLKA_WGRAD_COT"); return e ? atoi(e) : 0; }();   // (A/B knob, read once: fewer co-tiles per wave = fewer registers)
        if (cot_cap > 0 && pl.cot > cot_cap) pl.cot = cot_cap;
    }
    const int groups = cdiv(OT, pl.cot) * CT * cdiv(K, pl.tpw);
    const int tiles = cdiv(M, 32);
    // register-heavy variants run one wave per SIMD (1024 slots): fill them once rather than 2.004 times
    const int slots = (amode == 0 && K > 1 && pl.cot * pl.tpw >= 6) ? 1024 : 2048;   // <=2 waves/SIMD for the rest
    int chunks = slots / groups;
    if (K == 1) {   // pointwise: every partial is re-read by the fold; past a few hundred the fold costs more than the extra waves buy
        static int pw_cap = -1;
        if (pw_cap < 0) pw_cap = 512;
        if (chunks > pw_cap) chunks = pw_cap;
    }
    if (chunks > tiles) chunks = tiles;
    if (chunks > 1024) chunks = 1024;
    if (chunks < 1) chunks = 1;
    pl.chunks = chunks;
    return pl;
}

int cl_wgrad_pick_chunks(int M, int K, int Cout, int Cin, int amode) { return wgrad_plan(M, K, Cout, Cin, amode).chunks; }

size_t cl_wgrad_part_floats_mode(int M, int K, int Cout, int Cin, int amode)
{
    return (size_t)wgrad_plan(M, K, Cout, Cin, amode).chunks * ((size_t)K * round_up(Cout, 32) * Cin + round_up(Cout, 32));
}

size_t cl_wgrad_part_floats(int M, int K, int Cout, int Cin)
{
    const int c0 = wgrad_plan(M, K, Cout, Cin, 0).chunks, c1 = wgrad_plan(M, K, Cout, Cin, 1).chunks;
    return (size_t)(c0 > c1 ? c0 : c1) * ((size_t)K * round_up(Cout, 32) * Cin + round_up(Cout, 32));
}

// bytes of the zero-padded copy the padded dense kernels read (WgradArgs::pad): [B][D + (kd-1) dd][H + (kh-1) dh][W + (kw-1) dw][Cin] activation elements
size_t cl_wgrad_pad_bytes(int B, int D, int H, int W, int Cin, int kd, int kh, int kw, int dd, int dh, int dw, int act_bf16)
{
    return (size_t)B * (D + (kd - 1) * dd) * (H + (kh - 1) * dh) * (W + (kw - 1) * dw) * Cin * (act_bf16 ? 2 : 4);
}

template <typename T>
int launch_cl_wgrad(int amode, int gmode, WgradArgs a, T *gw, T *gb, hipStream_t st, FinalizeJob *defer)
{
    const WgradPlan pl = wgrad_plan(a.M, a.K, a.Cout, a.Cin, amode);
    const int tiles = cdiv(a.M, 32);
    a.rows_per_chunk = cdiv(tiles, pl.chunks) * 32;
    const int nchunks = cdiv(a.M, a.rows_per_chunk);
    a.CoutP = round_up(a.Cout, 32);
    a.CT = a.Cin / 32;
    constexpr bool slow_addr = false;
    a.w16 = (!slow_addr && (a.W & 15) == 0) ? 1 : 0;
    { const char *e = getenv("DLKA_WGRAD_WIN3"); a.no_win3 = (e && e[0] == '0') ? 1 : 0; }   // (read per launch: a parity test toggles it)
    const int OT = a.CoutP / 32;
    if ((long)a.M * a.Cin * 4 >= (1l << 31) || (long)a.M * a.Cout * 4 >= (1l << 31)) return DLKA_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    a.bpart = gb ? a.part + (size_t)nchunks * a.K * a.CoutP * a.Cin : nullptr;
    dim3 block(64);
    if (amode == 1) {
        if (gmode != 0 || a.K == 1) return DLKA_ERR_UNSUPPORTED;
        dim3 grid(nchunks, OT * a.CT, cdiv(a.K, pl.tpw));
        static const bool no_xcd = getenv("DLKA_NO_XCD_SWIZZLE") != nullptr;   // A/B switch
        if (!no_xcd && nchunks >= xcd_min_blocks()) {
            a.xcd_ny = grid.y; a.xcd_nz = grid.z; a.xcd_total = (int)(grid.x * grid.y * grid.z);
            grid = dim3(xcd_grid(a.xcd_total), 1, 1);
        }
        if (a.samp) {   // samples stored by the grad_offset kernel: dense stream, no gather
            if (pl.tpw != 3 || (long)a.K * a.M * a.Cin * (a.act_bf16 ? 2 : 4) >= (1l << 31)) return DLKA_ERR_UNSUPPORTED;   // 32-bit buffer offsets below DLKA_OOB
            if (a.act_bf16) { auto k = cl_wgrad_samp_kernel<3, bf16_t>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
            else if (a.samp_f16 && a.samp_b16mfma) { auto k = cl_wgrad_samp_kernel<3, float, true, true>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
            else if (a.samp_f16) { auto k = cl_wgrad_samp_kernel<3, float, true>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
            else { auto k = cl_wgrad_samp_kernel<3>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
        }
        else if (a.act_bf16) {
            if (pl.tpw != 3) return DLKA_ERR_UNSUPPORTED;
            auto k = cl_wgrad_deform_kernel<3, bf16_t>; DLKA_LAUNCH(k, grid, block, 0, st, a);
        }
        else if (pl.tpw == 3) { auto k = cl_wgrad_deform_kernel<3>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
        else return DLKA_ERR_UNSUPPORTED;   // (4 and 7 taps per wave measured slower: more registers, fewer waves)
    } else {
        dim3 grid(nchunks, cdiv(OT, pl.cot) * a.CT, cdiv(a.K, pl.tpw));
        static const bool no_xcd2 = getenv("DLKA_NO_XCD_SWIZZLE") != nullptr;
        if (!no_xcd2 && a.K > 1 && nchunks >= xcd_min_blocks()) {
            a.xcd_ny = grid.y; a.xcd_nz = grid.z; a.xcd_total = (int)(grid.x * grid.y * grid.z);
            grid = dim3(xcd_grid(a.xcd_total), 1, 1);
        }
        static const bool exact = getenv("DLKA_EXACT_FP32") != nullptr;
        const bool split = !exact && a.K > 1;   // MFMA-bound contractions: bf16 x3 split (see cl_igemm.hip)
        if (a.g_cpad && !((a.N & 15) == 0 && split && gmode == 1)) return DLKA_ERR_UNSUPPORTED;   // packed g: split + N16 variant only
#define DLKA_WG(GM, CO, TP)                                                                                      \
    {                                                                                                            \
        if ((a.N & 15) == 0 && split) { auto k = cl_wgrad_dense_kernel<GM, CO, TP, true, true>; DLKA_LAUNCH(k, grid, block, 0, st, a); }  \
        else if ((a.N & 15) == 0) { auto k = cl_wgrad_dense_kernel<GM, CO, TP, true>; DLKA_LAUNCH(k, grid, block, 0, st, a); }  \
        else { auto k = cl_wgrad_dense_kernel<GM, CO, TP, false>; DLKA_LAUNCH(k, grid, block, 0, st, a); }              \
    }
        if (a.pad && a.K > 1 && gmode == 1 && split && !a.g_cpad && pl.tpw == 3 && (a.Cin * (a.act_bf16 ? 2 : 4)) % 16 == 0 &&
            cl_wgrad_pad_bytes(a.B, a.D, a.H, a.W, a.Cin, a.kd, a.kh, a.kw, a.dd, a.dh, a.dw, a.act_bf16) < ((size_t)1 << 31)) {
            // round 5: the zero-padded copy of the input first (one streaming pass), then the kernels that need no coordinate test
            a.DP = a.D + (a.kd - 1) * a.dd; a.HP = a.H + (a.kh - 1) * a.dh; a.WP = a.W + (a.kw - 1) * a.dw;
            const int prow = a.B * a.DP * a.HP;
            const unsigned pgrid = (unsigned)(prow > 16384 ? 16384 : prow);
            if (a.act_bf16) {
                auto k = cl_pad_copy_kernel<bf16_t>;
                DLKA_LAUNCH(k, dim3(pgrid), dim3(256), 0, st, reinterpret_cast<const bf16_t *>(a.in), reinterpret_cast<bf16_t *>(const_cast<float *>(a.pad)), a.B, a.D, a.H, a.W,
                            a.Cin, a.DP, a.HP, a.WP, a.pd, a.ph, a.pw);
            } else {
                auto k = cl_pad_copy_kernel<float>;
                DLKA_LAUNCH(k, dim3(pgrid), dim3(256), 0, st, a.in, const_cast<float *>(a.pad), a.B, a.D, a.H, a.W, a.Cin, a.DP, a.HP, a.WP, a.pd, a.ph, a.pw);
            }
            DLKA_CHECK_LAUNCH();
            const bool n16 = (a.N & 15) == 0;
#define DLKA_WGP(CO, T_)                                                                                                   \
    {                                                                                                                      \
        if (n16) { auto k = cl_wgrad_dense_pad_kernel<CO, true, T_>; DLKA_LAUNCH(k, grid, block, 0, st, a); }              \
        else { auto k = cl_wgrad_dense_pad_kernel<CO, false, T_>; DLKA_LAUNCH(k, grid, block, 0, st, a); }                 \
    }
            if (a.act_bf16) { if (pl.cot == 3) DLKA_WGP(3, bf16_t) else if (pl.cot == 2) DLKA_WGP(2, bf16_t) else DLKA_WGP(1, bf16_t) }
            else { if (pl.cot == 3) DLKA_WGP(3, float) else if (pl.cot == 2) DLKA_WGP(2, float) else DLKA_WGP(1, float) }
#undef DLKA_WGP
        } else if (a.act_bf16) {   // DLKA_BF16 token path: only the offset-predict conv's weight gradient comes through here (planar fp32 g, bf16 in)
            if (a.K == 1 || gmode != 1 || !split || a.g_cpad || pl.tpw != 3) return DLKA_ERR_UNSUPPORTED;
#define DLKA_WGB(CO)                                                                                                                           \
    {                                                                                                                                          \
        if ((a.N & 15) == 0) { auto k = cl_wgrad_dense_kernel<1, CO, 3, true, true, bf16_t>; DLKA_LAUNCH(k, grid, block, 0, st, a); }    \
        else { auto k = cl_wgrad_dense_kernel<1, CO, 3, false, true, bf16_t>; DLKA_LAUNCH(k, grid, block, 0, st, a); }                   \
    }
            if (pl.cot == 3) DLKA_WGB(3) else if (pl.cot == 2) DLKA_WGB(2) else DLKA_WGB(1)
#undef DLKA_WGB
        } else if (a.K == 1) {
            if (gmode != 0) return DLKA_ERR_UNSUPPORTED;
            if (pl.cot == 2) DLKA_WG(0, 2, 1) else DLKA_WG(0, 1, 1)
        } else if (gmode == 1) {
            if (pl.cot == 3 && pl.tpw == 1) DLKA_WG(1, 3, 1) else if (pl.cot == 3 && pl.tpw == 2) DLKA_WG(1, 3, 2)
            else if (pl.cot == 3) DLKA_WG(1, 3, 3) else if (pl.cot == 2) DLKA_WG(1, 2, 3) else DLKA_WG(1, 1, 3)
        } else {
            if (pl.cot == 3) DLKA_WG(0, 3, 3) else if (pl.cot == 2) DLKA_WG(0, 2, 3) else DLKA_WG(0, 1, 3)
        }
#undef DLKA_WG
    }
    DLKA_CHECK_LAUNCH();
    const long n = (long)a.K * a.Cout * a.Cin + (gb ? a.Cout : 0);
    if (defer) {   // the caller folds the partials of several gradients in one launch (launch_cl_wgrad_finalize)
        defer->part = a.part; defer->bpart = a.bpart; defer->gw = (float *)gw; defer->gb = (float *)gb;
        defer->chunks = nchunks; defer->K = a.K; defer->CoutP = a.CoutP; defer->Cout = a.Cout; defer->Cin = a.Cin; defer->kind = 0; defer->n = n;
        return DLKA_OK;
    }
    long blocks = cdivl(n, 32);
    if (blocks > 4096) blocks = 4096;
    auto rk = cl_wgrad_reduce_kernel<T>;
    DLKA_LAUNCH(rk, dim3((unsigned)blocks), dim3(256), 0, st, (const float *)a.part, (const float *)a.bpart, gw, gb, nchunks, a.K, a.CoutP, a.Cout, a.Cin);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// three pointwise (K = 1) weight gradients with identical geometry; always deferred to the caller's fused finalisation
int launch_cl_wgrad_pw3(const WgradArgs *jobs, float *const *gw, float *const *gb, hipStream_t st, FinalizeJob *defer)
{
    WgradArgs3 b;
    const WgradArgs &a0 = jobs[0];
    if (a0.K != 1) return DLKA_ERR_UNSUPPORTED;
    const WgradPlan pl = wgrad_plan(a0.M, 1, a0.Cout, a0.Cin, 0);
    const int tiles = cdiv(a0.M, 32);
    const int rpc = cdiv(tiles, pl.chunks) * 32;
    const int nchunks = cdiv(a0.M, rpc);
    const int CoutP = round_up(a0.Cout, 32), CT = a0.Cin / 32, OT = CoutP / 32;
    if ((long)a0.M * a0.Cin * 4 >= (1l << 31) || (long)a0.M * a0.Cout * 4 >= (1l << 31)) return DLKA_ERR_UNSUPPORTED;
    for (int k = 0; k < 3; ++k) {
        b.a[k] = jobs[k];
        WgradArgs &a = b.a[k];
        a.rows_per_chunk = rpc; a.CoutP = CoutP; a.CT = CT;
        a.bpart = gb[k] ? a.part + (size_t)nchunks * CoutP * a.Cin : nullptr;
        FinalizeJob &d = defer[k];
        memset(&d, 0, sizeof(d));
        d.part = a.part; d.bpart = a.bpart; d.gw = gw[k]; d.gb = gb[k];
        d.chunks = nchunks; d.K = 1; d.CoutP = CoutP; d.Cout = a.Cout; d.Cin = a.Cin; d.kind = 0; d.n = (long)a.Cout * a.Cin + (gb[k] ? a.Cout : 0);
    }
    dim3 grid(nchunks, cdiv(OT, pl.cot) * CT, 3), block(64);
    const bool n16 = (a0.N & 15) == 0;
    if (a0.act_bf16) {
        if (pl.cot == 2 && n16) { auto k = cl_wgrad_pw3_kernel<2, true, bf16_t>; DLKA_LAUNCH(k, grid, block, 0, st, b); }
        else if (pl.cot == 2) { auto k = cl_wgrad_pw3_kernel<2, false, bf16_t>; DLKA_LAUNCH(k, grid, block, 0, st, b); }
        else if (n16) { auto k = cl_wgrad_pw3_kernel<1, true, bf16_t>; DLKA_LAUNCH(k, grid, block, 0, st, b); }
        else { auto k = cl_wgrad_pw3_kernel<1, false, bf16_t>; DLKA_LAUNCH(k, grid, block, 0, st, b); }
    }
    else if (pl.cot == 2 && n16) { auto k = cl_wgrad_pw3_kernel<2, true>; DLKA_LAUNCH(k, grid, block, 0, st, b); }
    else if (pl.cot == 2) { auto k = cl_wgrad_pw3_kernel<2, false>; DLKA_LAUNCH(k, grid, block, 0, st, b); }
    else if (n16) { auto k = cl_wgrad_pw3_kernel<1, true>; DLKA_LAUNCH(k, grid, block, 0, st, b); }
    else { auto k = cl_wgrad_pw3_kernel<1, false>; DLKA_LAUNCH(k, grid, block, 0, st, b); }
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

template int launch_cl_wgrad<float>(int, int, WgradArgs, float *, float *, hipStream_t, FinalizeJob *);

// All weight-gradient finalisations of one D-LKA block in ONE launch: the five partial-sum folds (3 pointwise, offset
// conv, deformable conv) and the two depthwise [tap][c] -> [c][tap] re-layouts were seven ~5-10 us launches per block.
// (blk: workgroup index inside the job)
__device__ __forceinline__ void wgrad_finalize_body(const FinalizeJob &jb, const long blk)
{
    __shared__ float red[8][33];
    const long e = blk * 32 + (threadIdx.x & 31);
    const int el = threadIdx.x & 31, cl = threadIdx.x >> 5;
    if (jb.kind == 1) {   // depthwise staging [K + 1][C] (row K = bias sums): gw[c][tap] = gwp[tap][c], gb[c] = gwp[K][c]
        if (cl == 0 && e < jb.n) {
            const long nw = (long)jb.K * jb.Cin;
            if (e < nw) {
                const int tap = (int)(e % jb.K), c = (int)(e / jb.K);
                jb.gw[e] = jb.part[(long)tap * jb.Cin + c];
            } else {
                jb.gb[e - nw] = jb.part[nw + (e - nw)];
            }
        }
        return;
    }
    const long n = (long)jb.K * jb.Cout * jb.Cin;
    const long stride = (long)jb.K * jb.CoutP * jb.Cin;
    if (jb.tr) {
        // Few partials, many outputs (the small stages: 27 x 256 x 256 weights, 3 partials): a workgroup owns (co, 32 ci, all K taps).
        // Reads stay 128-byte rows of the [tap][co][ci] partial tiles; the K x 32 results are re-laid through LDS and leave as ONE
        // contiguous run of 32*K floats of gW[co][ci][tap] — the element-per-lane version wrote 4-byte pieces K*4 bytes apart
        // (67 us for the C = 256 block, profiles/archive/r01n).
        __shared__ float tile[32 * 28];   // [ci][tap], K <= 27 (+1 padding)
        const int cblocks = jb.Cin / 32;
        const long wblocks = (long)jb.Cout * cblocks;
        if (blk < wblocks) {
            const int co = (int)(blk / cblocks), ci0 = (int)(blk % cblocks) * 32;
            for (int tp = cl; tp < jb.K; tp += 8) {
                const float *src = jb.part + ((long)tp * jb.CoutP + co) * jb.Cin + ci0 + el;
                float s0 = 0.f, s1 = 0.f;
                int c = 0;
                for (; c + 1 < jb.chunks; c += 2) { s0 += src[(long)c * stride]; s1 += src[(long)(c + 1) * stride]; }
                if (c < jb.chunks) s0 += src[(long)c * stride];
                tile[el * 28 + tp] = s0 + s1;
            }
            __syncthreads();
            float *dst = jb.gw + ((long)co * jb.Cin + ci0) * jb.K;
            for (int l = threadIdx.x; l < 32 * jb.K; l += 256) dst[l] = tile[(l / jb.K) * 28 + l % jb.K];
        } else if (jb.gb) {
            const int co = (int)(blk - wblocks) * 32 + el;
            float s0 = 0.f;
            if (co < jb.Cout)
                for (int c = cl; c < jb.chunks; c += 8) s0 += jb.bpart[(long)c * jb.CoutP + co];
            red[cl][el] = s0;
            __syncthreads();
            if (cl == 0 && co < jb.Cout)
                jb.gb[co] = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
        }
        return;
    }
    float a0 = 0.f, a1 = 0.f;
    int ci = 0, co = 0, tap = 0;
    if (e < n) {
        ci = (int)(e % jb.Cin); co = (int)((e / jb.Cin) % jb.Cout); tap = (int)(e / jb.Cin / jb.Cout);
        const float *src = jb.part + ((long)tap * jb.CoutP + co) * jb.Cin + ci;
        int c = cl;
        // 8 independent loads in flight per work-item: with 2048 partials per output (pointwise convs at 32^3) the fold is a
        // chain of dependent-latency loads otherwise (133 us for the stage-0 batch with two chains, profiles/archive/r01n)
        float a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
        for (; c + 56 < jb.chunks; c += 64) {
            a0 += src[(long)c * stride]; a1 += src[(long)(c + 8) * stride]; a2 += src[(long)(c + 16) * stride]; a3 += src[(long)(c + 24) * stride];
            a4 += src[(long)(c + 32) * stride]; a5 += src[(long)(c + 40) * stride]; a6 += src[(long)(c + 48) * stride]; a7 += src[(long)(c + 56) * stride];
        }
        for (; c < jb.chunks; c += 8) a0 += src[(long)c * stride];
        a0 = ((a0 + a2) + (a4 + a6)); a1 = ((a1 + a3) + (a5 + a7));
    } else if (e < jb.n) {
        co = (int)(e - n);
        for (int c = cl; c < jb.chunks; c += 8) a0 += jb.bpart[(long)c * jb.CoutP + co];
    }
    red[cl][el] = a0 + a1;
    __syncthreads();
    if (cl == 0 && e < jb.n) {
        const float t = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
        if (e < n) jb.gw[((long)co * jb.Cin + ci) * jb.K + tap] = t;
        else jb.gb[co] = t;
    }
}

__global__ __launch_bounds__(256) void cl_wgrad_finalize_kernel(FinalizeBatch b)
{
    int ji = 0;
    while (ji + 1 < b.njobs && (long)blockIdx.x >= b.j[ji + 1].block0) ++ji;
    wgrad_finalize_body(b.j[ji], (long)blockIdx.x - b.j[ji].block0);
}

// The finalisations of MANY blocks (a whole backward pass) in one launch: the job table lives in device memory (built once: partial-sum areas and
// gradient buffers never move), jobs[k].block0 ascending; this launch covers jobs [job_lo, job_hi) and workgroup 0 is jobs[job_lo].block0.
__global__ __launch_bounds__(256) void cl_wgrad_finalize_table_kernel(const FinalizeJob *__restrict__ jobs, int job_lo, int job_hi)
{
    const long blk = (long)blockIdx.x + jobs[job_lo].block0;
    int lo = job_lo, hi = job_hi - 1;   // last job whose block0 <= blk
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= blk) lo = mid; else hi = mid - 1;
    }
    const FinalizeJob jb = jobs[lo];
    wgrad_finalize_body(jb, blk - jb.block0);
}

// workgroups a job needs; sets the job's fold variant
long cl_wgrad_finalize_plan_job(FinalizeJob &j)
{
    constexpr bool no_tr = false;
    j.tr = (!no_tr && j.kind == 0 && j.K > 1 && j.K <= 27 && j.chunks <= 16 && j.Cin % 32 == 0 && (long)j.Cout * j.Cin >= 1024) ? 1 : 0;
    return j.tr ? (long)j.Cout * (j.Cin / 32) + (j.gb ? cdiv(j.Cout, 32) : 0) : cdivl(j.n, 32);
}

int launch_cl_wgrad_finalize_table(const FinalizeJob *jobs_device, int job_lo, int job_hi, long nblocks, hipStream_t st)
{
    if (job_hi <= job_lo || nblocks <= 0) return DLKA_OK;
    if (nblocks > 0x7fffffffL) return DLKA_ERR_UNSUPPORTED;
    DLKA_LAUNCH(cl_wgrad_finalize_table_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, jobs_device, job_lo, job_hi);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_wgrad_finalize(FinalizeBatch &b, hipStream_t st)
{
    if (b.njobs <= 0) return DLKA_OK;
    long blk = 0;
    for (int k = 0; k < b.njobs; ++k) {
        FinalizeJob &j = b.j[k];
        j.block0 = blk;
        blk += cl_wgrad_finalize_plan_job(j);
    }
    b.nblocks = blk;
    if (blk > 0x7fffffffL) return DLKA_ERR_UNSUPPORTED;
    DLKA_LAUNCH(cl_wgrad_finalize_kernel, dim3((unsigned)blk), dim3(256), 0, st, b);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// ---------------------------------------------------------------------------------------------
// column sums of a channels-last matrix:  gb[n] = sum_m G[m][n]     (bias gradients)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cl_colsum_kernel(const float *__restrict__ g, float *__restrict__ gb, int M, int Cout, int rows_per_block)
{
    __shared__ float red[256];
    const int cols = Cout < 256 ? Cout : 256;       // columns handled per pass
    const int rpp = 256 / cols;                     // rows in flight per pass
    const int c_in = threadIdx.x % cols, r_in = threadIdx.x / cols;
    const int m_lo = blockIdx.x * rows_per_block, m_hi = min(M, m_lo + rows_per_block);
    for (int cb = 0; cb < Cout; cb += cols) {
        const int c = cb + c_in;
        float a = 0.f;
        if (c < Cout && r_in < rpp)
            for (int m = m_lo + r_in; m < m_hi; m += rpp) a += g[(long)m * Cout + c];
        red[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < cols && c < Cout) {
            float t = 0.f;
            for (int r = 0; r < rpp; ++r) t += red[r * cols + threadIdx.x];
            atomicAdd(gb + c, t);
        }
        __syncthreads();
    }
}

int launch_cl_colsum(const float *g, float *gb32, int M, int Cout, hipStream_t st)
{
    if (launch_zero(gb32, (size_t)Cout * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
    int blocks = cdiv(M, 256);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    const int rpb = cdiv(M, blocks);
    DLKA_LAUNCH(cl_colsum_kernel, dim3(cdiv(M, rpb)), dim3(256), 0, st, g, gb32, M, Cout, rpb);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

}  // namespace dlka
