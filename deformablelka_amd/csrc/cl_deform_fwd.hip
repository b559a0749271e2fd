// Channels-last forward of the 3-D deformable convolution (D3D semantics, groups = deformable_groups = 1) on the matrix
// cores, second generation:
//     out[m][n] = bias[n] + sum_tap sum_c  S(m, tap, c) * Wp[tap][c][n],     S = trilinear sample (deform_im2col_cuda.cuh:26-72)
// The reference writes S for all (tap, c) to a 27*C x B*N column buffer in HBM and multiplies it with at::addmm
// (deform_conv_cuda.cu:95-119).  Here S never leaves the chip: per (tap, 32-channel chunk) a wave gathers the 32 x 32
// sample tile in the line-friendly layout of cl_gather.h (lane = (row of 8, 16-byte piece of 8): every load instruction
// covers 8 whole 128-byte rows — 3.5x the gather rate of the first version's "lane = row" loads), transposes it through
// a wave-private LDS tile into the MFMA A layout, and feeds v_mfma_f32_32x32x2_f32.  The gather loads of unit u+1 are in
// flight under the MFMAs of unit u; the 4 waves of a workgroup share the 32 x NP weight chunk through LDS.
#include <stdlib.h>

#include "cl_args.h"
#include "cl_gather.h"
#include "dlka_kernels.h"

namespace dlka {

template <typename T, int NT>   // T: activation storage of `in` / `out` (float | bf16_t); split partial sums (gridDim.y > 1) always land in fp32
__global__ __launch_bounds__(256) void cl_deform_fwd_kernel(IgemmArgs p)
{
    constexpr unsigned SB = sizeof(T);
    constexpr int NPB = NT * 32;
    constexpr int BV = NT;
    constexpr int SROW = 36;   // padded sample-tile row (floats): 16-byte aligned, conflict-free b128 rows
    __shared__ __attribute__((aligned(16))) float Bs[2][32 * NPB];
    __shared__ __attribute__((aligned(16))) float Ssm[4][32 * SROW];
    __shared__ __attribute__((aligned(16))) float Dsm[4][32 * GATHER_DESCW_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;      // MFMA roles
    using GG = GatherGeom<T>;
    const int gr = lane >> GG::PSHIFT, gp = lane & ((1 << GG::PSHIFT) - 1);   // gather roles: row gr of each group of RPI, 16-byte piece gp
    const int bx = DLKA_XCD_BX(p.xcd_nx);
    if (bx < 0) return;
    const int mbase = (bx * 4 + wave) * 32;
    const int m = mbase + i;
    const bool row_ok = m < p.M;
    const int b = row_ok ? m / p.N : 0;
    const int v = row_ok ? m - b * p.N : 0;
    const int w0 = v % p.W, h0 = (v / p.W) % p.H, d0 = v / (p.W * p.H);
    const int n0 = blockIdx.z * NPB;
    const int HW = p.H * p.W, rowbytes = p.Cin * SB;
    const BufRsrc rin = make_rsrc(p.in, (size_t)p.M * p.Cin * SB);
    float *S = Ssm[wave], *Dt = Dsm[wave];

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nchunk = p.CinP / 32;
    const int unit_lo = blockIdx.y * p.units_per_split;
    const int unit_hi = min(p.K * nchunk, unit_lo + p.units_per_split);

    f32x4 breg[BV];
    GatherPiece<T> xr[GG::NG][8];   // gathered corner pieces of the next unit, in flight
    int cur_tap = -1;

#define DLKA_LOAD_B(unit_)                                                                         \
    {                                                                                              \
        int ck_;                                                                                   \
        const int tap_ = divmod_fast((unit_), nchunk, ck_);                                        \
        const float *src_ = p.wp + ((long)tap_ * p.CinP + ck_ * 32) * p.NP + n0;                   \
        _Pragma("unroll") for (int e = 0; e < BV; ++e) {                                           \
            const int idx_ = tid + e * 256;                                                        \
            const int rr_ = idx_ / (NPB / 4), c4_ = idx_ - rr_ * (NPB / 4);                        \
            breg[e] = reinterpret_cast<const f32x4 *>(src_ + (long)rr_ * p.NP)[c4_];              \
        }                                                                                          \
    }
    // The offsets of a tap are requested one tap AHEAD of their description (onx): loaded inside the description, every tap began with two
    // dependent L2 round trips — offsets, then the corner rows they address — of which the one-unit prefetch distance hides one at best.
    float onx[3] = {0.f, 0.f, 0.f};
    const int tap_first = unit_lo / nchunk, tap_last = (unit_hi - 1) / nchunk;
    auto load_offsets = [&](int tap) {
        if (h == 0 && row_ok && tap <= tap_last) {
            const float *op = p.off + ((long)b * 3 * p.K + 3 * tap) * p.N + v;
            onx[0] = op[0]; onx[1] = op[p.N]; onx[2] = op[2 * (long)p.N];
        }
    };
    if (unit_lo < unit_hi) load_offsets(tap_first);
    // describe (when the tap changes) and issue the 32 corner loads of one unit
    auto issue = [&](int unit) {
        int ck;
        const int tap = divmod_fast(unit, nchunk, ck);
        if (tap != cur_tap) {   // uniform
            cur_tap = tap;
            int ti, tj, tk;
            tap_decode(tap, p.kw, p.kh, ti, tj, tk);
            wave_sync();        // every lane has consumed the previous table
            if (h == 0) {
                RowDesc r;
                r.base = 0; r.okm = 0; r.ld = r.lh = r.lw = 0.f;
                if (row_ok)
                    r = gather_describe3(onx[0], onx[1], onx[2], p.N, b, d0 + ti * p.dd - p.pd, h0 + tj * p.dh - p.ph, w0 + tk * p.dw - p.pw, p.D, p.H, p.W);
                gather_publish_w(Dt, i, r, rowbytes);
            }
            load_offsets(tap + 1);   // in flight until the next tap is described
            wave_sync();   // (the table keeps this tap's rows until the next tap is described: descriptions and weights are read where they are used)
        }
        const unsigned cbyte = (unsigned)(ck * 32 + GG::PE * gp) * SB;
#pragma unroll
        for (int g = 0; g < GG::NG; ++g) {
            const RowLook r = gather_lookup_d(Dt, GG::RPI * g + gr);
#pragma unroll
            for (int q = 0; q < 8; ++q) xr[g][q] = gather_load<T>(rin, gather_offset(r, q, HW, p.W, rowbytes, cbyte));
        }
    };
    // interpolate, transpose through the wave-private tile, return the MFMA A values of this lane
    auto finish = [&](float *a) {
        wave_sync();   // previous tile consumed
#pragma unroll
        for (int g = 0; g < GG::NG; ++g) {
            float wq[8];   // formed once per row by the publishing lane
            gather_lookup_weights(Dt, GG::RPI * g + gr, wq);
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
        wave_sync();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(S + i * SROW + 16 * h + 4 * e);
            a[4 * e] = t[0]; a[4 * e + 1] = t[1]; a[4 * e + 2] = t[2]; a[4 * e + 3] = t[3];
        }
    };

    if (unit_lo < unit_hi) {
        DLKA_LOAD_B(unit_lo)
        issue(unit_lo);
    }
    int buf = 0;
    for (int unit = unit_lo; unit < unit_hi; ++unit, buf ^= 1) {
        float a_cur[16];
#pragma unroll
        for (int e = 0; e < BV; ++e) reinterpret_cast<f32x4 *>(Bs[buf])[tid + e * 256] = breg[e];
        finish(a_cur);      // consumes xr (the loads issued one iteration ago)
        __syncthreads();    // Bs[buf] staged; Bs[buf^1] (read two iterations ago) is free again
        if (unit + 1 < unit_hi) {
            DLKA_LOAD_B(unit + 1)
            issue(unit + 1);
        }
        const float *brow = Bs[buf] + (16 * h) * NPB + i;
#pragma unroll
        for (int st = 0; st < 16; ++st) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma_32x32x2(a_cur[st], brow[st * NPB + t * 32], acc[t]);
        }
    }
#undef DLKA_LOAD_B

    // ---- epilogue: D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
    const bool split = gridDim.y > 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + t * 32 + i;
        if (n >= p.Cout) continue;
        const float bv = (p.bias && blockIdx.y == 0) ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = mbase + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mr >= p.M) continue;
            const float val = acc[t][r] + bv;
            if (split) p.out[((long)blockIdx.y * p.M + mr) * p.Cout + n] = val;   // this tap range's slab (fp32 [splits][M][Cout]): summed in slab order by cl_slab_reduce_kernel
            else act_store1(reinterpret_cast<T *>(p.out), (long)mr * p.Cout + n, val);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 16-row waves (round 3).  The kernel above is latency-bound at 2 waves per SIMD (profiles/archive/r03p_pmc_issue_wait_stage0_f32.txt: an instruction of
// any kind issues in 37 % of the resident wave-cycles): 128 VGPRs of corner pieces in flight per 32-row wave, and a 32-row tile count that gives
// the chip exactly two waves per SIMD at stage 0 (65 536 rows / 32 = 2048 waves on 1024 SIMDs).  Here a wave owns 16 rows: half the corner pieces
// (64 VGPRs), v_mfma_f32_16x16x4_f32 (same FLOP rate as 32x32x2, MI355X_MICROARCH.md), twice the waves — 4 per SIMD under the 128-register budget.
// Same arithmetic per output element: the k order within a 32-channel chunk is permuted identically on both operands (lane group g holds
// channels 8g .. 8g+7), the fmaf order of the interpolation is unchanged.  One 32-column tile per workgroup (blockIdx.z walks wider outputs).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int NTC>   // NTC: 32-column tiles per workgroup (1 | 2)
__global__ __launch_bounds__(512, 4) void cl_deform_fwd16_kernel(IgemmArgs p)
{
    constexpr unsigned SB = sizeof(T);
    constexpr int WAVES = 8;
    constexpr int SROW = 36;   // padded sample-tile row (floats)
    constexpr int TGRP = 4;    // taps described at once: lane (row i, sub-tap g4) — all 64 lanes work (a per-tap description keeps 16 of them busy)
    using GG = GatherGeom<T>;
    constexpr int NG = GG::NG / 2;   // row groups of a 16-row tile: fp32 2 x 8 rows, bf16 1 x 16 rows
    constexpr int NPB = 32 * NTC;
    __shared__ __attribute__((aligned(16))) float Bs[2][32 * NPB];
    __shared__ __attribute__((aligned(16))) float Ssm[WAVES][16 * SROW];
    __shared__ __attribute__((aligned(16))) float Dsm[WAVES][TGRP * 16 * GATHER_DESCW_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g4 = lane >> 4;                                  // MFMA roles: row / column index, k group; description role: row, sub-tap
    const int gr = lane >> GG::PSHIFT, gp = lane & ((1 << GG::PSHIFT) - 1);   // gather roles: row gr of each group of RPI, 16-byte piece gp
    const int bx = DLKA_XCD_BX(p.xcd_nx);
    if (bx < 0) return;
    const int mbase = (bx * WAVES + wave) * 16;
    const int m = mbase + i;
    const bool row_ok = m < p.M;
    const int b = row_ok ? m / p.N : 0;
    const int v = row_ok ? m - b * p.N : 0;
    const int w0 = v % p.W, h0 = (v / p.W) % p.H, d0 = v / (p.W * p.H);
    const int n0 = blockIdx.z * NPB;
    const int HW = p.H * p.W, rowbytes = p.Cin * SB;
    const BufRsrc rin = make_rsrc(p.in, (size_t)p.M * p.Cin * SB);
    float *S = Ssm[wave], *Dt = Dsm[wave];

    f32x4 acc[2 * NTC];
#pragma unroll
    for (int t = 0; t < 2 * NTC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunk = p.CinP / 32;
    const int unit_lo = blockIdx.y * p.units_per_split;
    const int unit_hi = min(p.K * nchunk, unit_lo + p.units_per_split);

    f32x4 breg = {0.f, 0.f, 0.f, 0.f};
    const bool bload = tid < 256 * NTC;   // 32 x NPB weight chunk: one 16-byte piece per thread (NTC = 1: the first four waves)
    GatherPiece<T> xr[NG][8];   // gathered corner pieces of the next unit, in flight
    int cur_grp = -1, cur_tap = 0;
    auto load_b = [&](int unit) {
        if (bload) {
            int ck;
            const int tap = divmod_fast(unit, nchunk, ck);
            const float *src = p.wp + ((long)tap * p.CinP + ck * 32) * p.NP + n0;
            breg = reinterpret_cast<const f32x4 *>(src + (long)(tid / (NPB / 4)) * p.NP)[tid % (NPB / 4)];
        }
    };
    // s_memtime stamps (round 3, scripts/fwd_stamps.py history in profiles/r04_notes.md) showed a THIRD of every (tile, tap) step going into the
    // description of the tap — offsets, guard, floor, corner mask, weights — executed as full-wave instructions for 16 useful lanes, on the critical
    // path of a kernel that is bound by its per-step instruction chain.  Four taps are therefore described at once, lane = (row, sub-tap): the same
    // instructions, every fourth tap.  The offsets of a group are requested a whole group ahead.
    float onx[3] = {0.f, 0.f, 0.f};
    const int tap_last = (unit_hi - 1) / nchunk;
    auto load_offsets = [&](int grp) {
        const int tap = TGRP * grp + g4;
        if (row_ok && tap <= tap_last && tap < p.K) {
            const float *op = p.off + ((long)b * 3 * p.K + 3 * tap) * p.N + v;
            onx[0] = op[0]; onx[1] = op[p.N]; onx[2] = op[2 * (long)p.N];
        }
    };
    if (unit_lo < unit_hi) load_offsets((unit_lo / nchunk) / TGRP);
    auto issue = [&](int unit) {
        int ck;
        const int tap = divmod_fast(unit, nchunk, ck);
        const int grp = tap / TGRP;
        cur_tap = tap;
        if (grp != cur_grp) {   // uniform
            cur_grp = grp;
            const int mytap = TGRP * grp + g4;
            int ti, tj, tk;
            tap_decode(mytap < p.K ? mytap : 0, p.kw, p.kh, ti, tj, tk);
            wave_sync();        // every lane has consumed the previous table
            RowDesc r;
            r.base = 0; r.okm = 0; r.ld = r.lh = r.lw = 0.f;
            r.zd = r.zh = r.zw = 0;
            if (row_ok && mytap <= tap_last && mytap < p.K)
                r = gather_describe3(onx[0], onx[1], onx[2], p.N, b, d0 + ti * p.dd - p.pd, h0 + tj * p.dh - p.ph, w0 + tk * p.dw - p.pw, p.D, p.H, p.W);
            gather_publish_w(Dt, g4 * 16 + i, r, rowbytes);
            load_offsets(grp + 1);
            wave_sync();
        }
        const float *tab = Dt + (tap - TGRP * grp) * 16 * GATHER_DESCW_WORDS;
        const unsigned cbyte = (unsigned)(ck * 32 + GG::PE * gp) * SB;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const RowLook r = gather_lookup_d(tab, GG::RPI * g + gr);
#pragma unroll
            for (int q = 0; q < 8; ++q) xr[g][q] = gather_load<T>(rin, gather_offset(r, q, HW, p.W, rowbytes, cbyte));
        }
    };
    // interpolate, transpose through the wave-private tile, return this lane's A values: channels 8 g4 .. 8 g4 + 7 of row i
    // (tap_of_tile: the tap the pieces in xr belong to — its weights are still in the table: a group is only re-described by a LATER issue())
    auto finish = [&](float *a, int tap_of_tile) {
        const float *tab = Dt + (tap_of_tile & (TGRP - 1)) * 16 * GATHER_DESCW_WORDS;
        wave_sync();   // previous tile consumed
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float wq[8];
            gather_lookup_weights(tab, GG::RPI * g + gr, wq);
#pragma unroll
            for (int vv = 0; vv < GG::PE / 4; ++vv) {
                f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x4 x4 = xr[g][q].get(vv);
                    s4[0] = fmaf(wq[q], x4[0], s4[0]); s4[1] = fmaf(wq[q], x4[1], s4[1]);
                    s4[2] = fmaf(wq[q], x4[2], s4[2]); s4[3] = fmaf(wq[q], x4[3], s4[3]);
                }
                *reinterpret_cast<f32x4 *>(S + (GG::RPI * g + gr) * SROW + GG::PE * gp + 4 * vv) = s4;
            }
        }
        wave_sync();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(S + i * SROW + 8 * g4 + 4 * e);
            a[4 * e] = t[0]; a[4 * e + 1] = t[1]; a[4 * e + 2] = t[2]; a[4 * e + 3] = t[3];
        }
    };

    if (unit_lo < unit_hi) {
        load_b(unit_lo);
        issue(unit_lo);
    }
    int buf = 0;
    for (int unit = unit_lo; unit < unit_hi; ++unit, buf ^= 1) {
        float a_cur[8];
        if (bload) reinterpret_cast<f32x4 *>(Bs[buf])[tid] = breg;
        finish(a_cur, cur_tap);      // consumes xr (the loads issued one iteration ago, for tap cur_tap)
        __syncthreads();    // Bs[buf] staged; Bs[buf^1] (read two iterations ago) is free again
        if (unit + 1 < unit_hi) {
            load_b(unit + 1);
            issue(unit + 1);
        }
        // B[k = 8 g4 + s][n = 32 c + 2 i + t]: one 8-byte LDS read per step and 32-column tile feeds two 16-column MFMAs (columns 2i and 2i + 1)
        const float *brow = Bs[buf] + (8 * g4) * NPB + 2 * i;
#pragma unroll
        for (int st = 0; st < 8; ++st) {
#pragma unroll
            for (int c = 0; c < NTC; ++c) {
                const float b0 = brow[st * NPB + 32 * c], b1 = brow[st * NPB + 32 * c + 1];
                acc[2 * c] = mfma_16x16x4(a_cur[st], b0, acc[2 * c]);
                acc[2 * c + 1] = mfma_16x16x4(a_cur[st], b1, acc[2 * c + 1]);
            }
        }
    }

    // ---- epilogue: D layout col = lane & 15 -> output columns n0 + 2 i + t, row = 4 * (lane >> 4) + r ----
    const bool split = gridDim.y > 1;
#pragma unroll
    for (int t = 0; t < 2 * NTC; ++t) {
        const int n = n0 + 32 * (t >> 1) + 2 * i + (t & 1);
        if (n >= p.Cout) continue;
        const float bv = (p.bias && blockIdx.y == 0) ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mr = mbase + 4 * g4 + r;
            if (mr >= p.M) continue;
            const float val = acc[t][r] + bv;
            if (split) p.out[((long)blockIdx.y * p.M + mr) * p.Cout + n] = val;   // this tap range's slab (fp32 [splits][M][Cout]): summed in slab order by cl_slab_reduce_kernel
            else act_store1(reinterpret_cast<T *>(p.out), (long)mr * p.Cout + n, val);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// DLKA_BF16 with the contraction on the bf16 matrix cores (round 4).  The kernels above feed v_mfma_f32_32x32x2_f32 / 16x16x4_f32 also when the
// activations are bf16: 16 (resp. 8) MFMAs of 64 cycles per (32-row tile, tap, 32-channel chunk), i.e. the fp32-input rate — 1/16 of what a bf16
// operand is entitled to.  Here
//   * the interpolated sample tile goes into the wave's LDS tile AS bf16 (one 8-byte store per 4 channels: half the LDS bytes) and comes back as the
//     MFMA A operand in ONE 16-byte read per k-group (lane (i, h), group mf: channels 16 h + 8 mf .. + 7 — the k order is permuted identically on
//     both operands);
//   * the weights are prepared as two-term bf16 records (cl_igemm.hip prep_store, mode | 8: w = hi + lo to 2^-17; [part][mf][h][n][8]) and read by
//     each lane straight from L2 with 16-byte loads — the B operand needs no LDS tile, hence no workgroup barrier per unit; the next unit's records
//     are requested before the current unit's MFMAs;
//   * per unit and 32-column tile: 4 v_mfma_f32_32x32x16_bf16 (hi and lo term for two k-groups) = 128 matrix-pipe cycles instead of 1024.
// Accumulation, bias and epilogue as above (fp32).  Same sampling rule, same fmaf order of the interpolation; the sample is rounded to bf16 once (it
// is a bf16-STORED activation's interpolation — the rounding is inside the 2e-2 contract of the bf16 path, measured with the parity tests).
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void cl_deform_fwd_b16_kernel(IgemmArgs p)
{
    using T = bf16_t;
    constexpr unsigned SB = 2;
    constexpr int SROWW = 20;   // sample-tile row in 32-bit words: 32 bf16 = 16 words + 4 of padding (80 bytes: 16-byte aligned rows)
    __shared__ __attribute__((aligned(16))) unsigned Ssm[4][32 * SROWW];
    __shared__ __attribute__((aligned(16))) float Dsm[4][32 * GATHER_DESCW_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;      // MFMA roles
    using GG = GatherGeom<T>;
    const int gr = lane >> GG::PSHIFT, gp = lane & ((1 << GG::PSHIFT) - 1);   // gather roles: row gr of each group of RPI, 16-byte piece gp
    const int bx = DLKA_XCD_BX(p.xcd_nx);
    if (bx < 0) return;
    const int mbase = (bx * 4 + wave) * 32;
    const int m = mbase + i;
    const bool row_ok = m < p.M;
    const int b = row_ok ? m / p.N : 0;
    const int v = row_ok ? m - b * p.N : 0;
    const int w0 = v % p.W, h0 = (v / p.W) % p.H, d0 = v / (p.W * p.H);
    const int n0 = blockIdx.z * NT * 32;
    const int HW = p.H * p.W, rowbytes = p.Cin * SB;
    const BufRsrc rin = make_rsrc(p.in, (size_t)p.M * p.Cin * SB);
    unsigned *S = Ssm[wave];
    float *Dt = Dsm[wave];

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nchunk = p.CinP / 32;
    const int unit_lo = blockIdx.y * p.units_per_split;
    const int unit_hi = min(p.K * nchunk, unit_lo + p.units_per_split);

    // weight records of a unit (tap, chunk): 32 * NP floats = [part (hi, lo)][mf][h][NP][8 bf16]
    f32x4 breg[NT][4];   // [t][part * 2 + mf]: this lane's records of the NEXT unit, in flight
    auto load_b = [&](int unit) {
        const float *src = p.wp + (long)unit * 32 * p.NP;   // (unit = tap * nchunk + chunk: the prepared layout's own unit order)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) breg[t][q] = *reinterpret_cast<const f32x4 *>(src + ((long)(q * 2 + h) * p.NP + n0 + 32 * t + i) * 4);
    };
    GatherPiece<T> xr[GG::NG][8];   // gathered corner pieces of the next unit, in flight
    int cur_tap = -1;
    float onx[3] = {0.f, 0.f, 0.f};
    const int tap_first = unit_lo / nchunk, tap_last = (unit_hi - 1) / nchunk;
    auto load_offsets = [&](int tap) {
        if (h == 0 && row_ok && tap <= tap_last) {
            const float *op = p.off + ((long)b * 3 * p.K + 3 * tap) * p.N + v;
            onx[0] = op[0]; onx[1] = op[p.N]; onx[2] = op[2 * (long)p.N];
        }
    };
    if (unit_lo < unit_hi) load_offsets(tap_first);
    auto issue = [&](int unit) {
        int ck;
        const int tap = divmod_fast(unit, nchunk, ck);
        if (tap != cur_tap) {   // uniform
            cur_tap = tap;
            int ti, tj, tk;
            tap_decode(tap, p.kw, p.kh, ti, tj, tk);
            wave_sync();        // every lane has consumed the previous table
            if (h == 0) {
                RowDesc r;
                r.base = 0; r.okm = 0; r.ld = r.lh = r.lw = 0.f;
                if (row_ok)
                    r = gather_describe3(onx[0], onx[1], onx[2], p.N, b, d0 + ti * p.dd - p.pd, h0 + tj * p.dh - p.ph, w0 + tk * p.dw - p.pw, p.D, p.H, p.W);
                gather_publish_w(Dt, i, r, rowbytes);
            }
            load_offsets(tap + 1);
            wave_sync();
        }
        const unsigned cbyte = (unsigned)(ck * 32 + GG::PE * gp) * SB;
#pragma unroll
        for (int g = 0; g < GG::NG; ++g) {
            const RowLook r = gather_lookup_d(Dt, GG::RPI * g + gr);
#pragma unroll
            for (int q = 0; q < 8; ++q) xr[g][q] = gather_load<T>(rin, gather_offset(r, q, HW, p.W, rowbytes, cbyte));
        }
    };
    // interpolate (fp32), round once, store the tile as bf16, read this lane's two A operands back
    auto finish = [&](bf16x8 a[2]) {
        wave_sync();   // previous tile consumed
#pragma unroll
        for (int g = 0; g < GG::NG; ++g) {
            float wq[8];
            gather_lookup_weights(Dt, GG::RPI * g + gr, wq);
#pragma unroll
            for (int vv = 0; vv < GG::PE / 4; ++vv) {
                f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x4 x4 = xr[g][q].get(vv);
                    s4[0] = fmaf(wq[q], x4[0], s4[0]); s4[1] = fmaf(wq[q], x4[1], s4[1]);
                    s4[2] = fmaf(wq[q], x4[2], s4[2]); s4[3] = fmaf(wq[q], x4[3], s4[3]);
                }
                const unsigned long long pk = (unsigned long long)((unsigned)bf16_bits(s4[0]) | ((unsigned)bf16_bits(s4[1]) << 16)) |
                                              ((unsigned long long)((unsigned)bf16_bits(s4[2]) | ((unsigned)bf16_bits(s4[3]) << 16)) << 32);
                *reinterpret_cast<unsigned long long *>(S + (GG::RPI * g + gr) * SROWW + (GG::PE * gp + 4 * vv) / 2) = pk;   // (8-byte aligned: even word index)
            }
        }
        wave_sync();
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(S + i * SROWW + 8 * h + 4 * mf);   // channels 16 h + 8 mf .. + 7
            const float w4[4] = {t[0], t[1], t[2], t[3]};
            a[mf] = bf16x8_from_words(w4);
        }
    };

    if (unit_lo < unit_hi) {
        load_b(unit_lo);
        issue(unit_lo);
    }
    for (int unit = unit_lo; unit < unit_hi; ++unit) {
        bf16x8 a_cur[2];
        finish(a_cur);      // consumes xr (the loads issued one iteration ago)
        bf16x8 bcur[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float w4[4] = {breg[t][q][0], breg[t][q][1], breg[t][q][2], breg[t][q][3]};
                bcur[t][q] = bf16x8_from_words(w4);
            }
        if (unit + 1 < unit_hi) {
            load_b(unit + 1);
            issue(unit + 1);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                acc[t] = mfma_32x32x16_bf16(a_cur[mf], bcur[t][mf], acc[t]);       // hi term
                acc[t] = mfma_32x32x16_bf16(a_cur[mf], bcur[t][2 + mf], acc[t]);   // lo term
            }
    }

    // ---- epilogue: D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
    const bool split = gridDim.y > 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + t * 32 + i;
        if (n >= p.Cout) continue;
        const float bv = (p.bias && blockIdx.y == 0) ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = mbase + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mr >= p.M) continue;
            const float val = acc[t][r] + bv;
            if (split) p.out[((long)blockIdx.y * p.M + mr) * p.Cout + n] = val;   // this tap range's slab (fp32 [splits][M][Cout]): summed in slab order by cl_slab_reduce_kernel
            else act_store1(reinterpret_cast<T *>(p.out), (long)mr * p.Cout + n, val);
        }
    }
}

int launch_cl_deform_fwd(IgemmArgs a, int splits, hipStream_t st)
{
    if ((long)a.M * a.Cin * 4 >= (1l << 31)) return DLKA_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    // splits > 1 (small volumes): the tap ranges' partial tiles go to SLABS, a.out = fp32 [splits][M][Cout], which launch_cl_slab_reduce sums in slab order into the
    // output (rounds 1 - 5: fp32 atomics on a zero-filled output — the order of arrival decided the last bit; deform_conv_cuda.cu:95-123 is deterministic)
    a.units_per_split = cdiv(a.K * (a.CinP / 32), splits);
    splits = cdiv(a.K * (a.CinP / 32), a.units_per_split);
    // (splits > 1: a.out is the slab buffer fp32 [splits][M][Cout] — every element of every slab is written, nothing is zero-filled; round 6)
    const int NT_total = a.NP / 32;
    const int mblocks = cdiv(a.M, 128);
    // all column tiles in one workgroup (the gather is not repeated) up to 4; wider outputs split over gridDim.z
    int NT = NT_total;
    if (NT_total == 8) NT = 4;
    else if (NT_total == 3 || NT_total > 4) return DLKA_ERR_UNSUPPORTED;
    // small volumes: fewer column tiles per workgroup (-> more workgroups) beats not repeating the gather
    // (C=256 / 4^3: 57.8 -> 36.0 us with NT = 1; C=128 / 8^3: 47.6 -> 41.0 us with NT = 2; profiles/archive/r01s_dfwd_tuning.txt)
    if (a.M <= 256) NT = 1;
    else if (a.M <= 2048 && NT_total == 4) NT = 2;
    constexpr int nt_env = 0;
    if (nt_env == 1 || nt_env == 2 || nt_env == 4) { if (NT_total % nt_env == 0 && nt_env <= NT_total) NT = nt_env; }
    if (a.act_bf16 && a.split_bf16 == 2) {   // weights in two-term bf16 records: the contraction runs on the bf16 matrix cores (all stages)
        dim3 gridb(mblocks, splits, NT_total / NT), blockb(256);
        a.xcd_nx = 0;
        if (xcd_swizzle_enabled() && mblocks >= (unsigned)xcd_min_blocks()) { a.xcd_nx = mblocks; gridb.x = xcd_grid(mblocks); }
        switch (NT) {
            case 1: { auto k = cl_deform_fwd_b16_kernel<1>; DLKA_LAUNCH(k, gridb, blockb, 0, st, a); } break;
            case 2: { auto k = cl_deform_fwd_b16_kernel<2>; DLKA_LAUNCH(k, gridb, blockb, 0, st, a); } break;
            case 4: { auto k = cl_deform_fwd_b16_kernel<4>; DLKA_LAUNCH(k, gridb, blockb, 0, st, a); } break;
            default: return DLKA_ERR_UNSUPPORTED;
        }
        DLKA_CHECK_LAUNCH();
        return DLKA_OK;
    }
    // 16-row waves where the 32-row tiling leaves the chip at two waves per SIMD and the output is one 32-column tile (stage 0: C = 32)
    {
        // Measured (profiles/r04_notes.md): 81 vs 94 us at C = 32 / 32^3 (fp32; 75 vs 86 bf16); no gain at the smaller stages with the
        // two-tile variant (47.0 vs 48.1 / 36.4 vs 34.1 / 25.6 vs 25.5 us at stages 1 - 3), so it is used for one 32-column tile and large M.
        // DLKA_FWD16_MIN_ROWS lowers the row threshold so that small test shapes take this kernel too (not cached: tests toggle it).
        const char *e16 = getenv("DLKA_FWD16_MIN_ROWS");
        const bool want16 = e16 ? a.M >= atoi(e16) : (NT_total == 1 && a.M >= 16384);
        if (want16 && (NT_total == 1 || NT_total % 2 == 0) && NT_total <= 8 && a.NP % 32 == 0) {
            const int ntc = NT_total == 1 ? 1 : 2;
            const int mb16 = cdiv(a.M, 128);
            dim3 grid16(mb16, splits, NT_total / ntc), block16(512);
            a.xcd_nx = 0;
            if (xcd_swizzle_enabled() && mb16 >= (unsigned)xcd_min_blocks()) { a.xcd_nx = mb16; grid16.x = xcd_grid(mb16); }
            if (a.act_bf16 && ntc == 1) { auto k = cl_deform_fwd16_kernel<bf16_t, 1>; DLKA_LAUNCH(k, grid16, block16, 0, st, a); }
            else if (a.act_bf16) { auto k = cl_deform_fwd16_kernel<bf16_t, 2>; DLKA_LAUNCH(k, grid16, block16, 0, st, a); }
            else if (ntc == 1) { auto k = cl_deform_fwd16_kernel<float, 1>; DLKA_LAUNCH(k, grid16, block16, 0, st, a); }
            else { auto k = cl_deform_fwd16_kernel<float, 2>; DLKA_LAUNCH(k, grid16, block16, 0, st, a); }
            DLKA_CHECK_LAUNCH();
            return DLKA_OK;
        }
    }
    dim3 grid(mblocks, splits, NT_total / NT), block(256);
    a.xcd_nx = 0;
    if (xcd_swizzle_enabled() && mblocks >= (unsigned)xcd_min_blocks()) { a.xcd_nx = mblocks; grid.x = xcd_grid(mblocks); }
    if (a.act_bf16) {
        switch (NT) {
            case 1: { auto k = cl_deform_fwd_kernel<bf16_t, 1>; DLKA_LAUNCH(k, grid, block, 0, st, a); } break;
            case 2: { auto k = cl_deform_fwd_kernel<bf16_t, 2>; DLKA_LAUNCH(k, grid, block, 0, st, a); } break;
            case 4: { auto k = cl_deform_fwd_kernel<bf16_t, 4>; DLKA_LAUNCH(k, grid, block, 0, st, a); } break;
            default: return DLKA_ERR_UNSUPPORTED;
        }
    } else {
        switch (NT) {
            case 1: { auto k = cl_deform_fwd_kernel<float, 1>; DLKA_LAUNCH(k, grid, block, 0, st, a); } break;
            case 2: { auto k = cl_deform_fwd_kernel<float, 2>; DLKA_LAUNCH(k, grid, block, 0, st, a); } break;
            case 4: { auto k = cl_deform_fwd_kernel<float, 4>; DLKA_LAUNCH(k, grid, block, 0, st, a); } break;
            default: return DLKA_ERR_UNSUPPORTED;
        }
    }
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// out[e] = sum_s slab[s][e], s = 0 .. S-1 IN THAT ORDER (the bias rode in slab 0): the deterministic meeting point of the tap-split deformable forward.
template <typename T>
__global__ __launch_bounds__(256) void cl_slab_reduce_kernel(const float *__restrict__ slab, int S, long n, T *__restrict__ out)
{
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= n) return;
    if (e + 3 < n) {
        f32x4 acc = *reinterpret_cast<const f32x4 *>(slab + e);
        for (int s = 1; s < S; ++s) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(slab + (long)s * n + e);
            acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) act_store1(out, e + k, acc[k]);
    } else {
        for (long j = e; j < n; ++j) {
            float acc = slab[j];
            for (int s = 1; s < S; ++s) acc += slab[(long)s * n + j];
            act_store1(out, j, acc);
        }
    }
}

int cl_deform_fwd_actual_splits(int K, int CinP, int splits)
{
    const int units = K * (CinP / 32), ups = cdiv(units, splits < 1 ? 1 : splits);
    return cdiv(units, ups);
}

int launch_cl_slab_reduce(const float *slab, int S, long n, void *out, int out_bf16, hipStream_t st)
{
    if (n % 4) return DLKA_ERR_UNSUPPORTED;   // (16-byte slab rows: M * Cout with Cout % 32 == 0)
    const unsigned grid = (unsigned)cdiv(n / 4, 256);
    if (out_bf16) { auto k = cl_slab_reduce_kernel<bf16_t>; DLKA_LAUNCH(k, dim3(grid), dim3(256), 0, st, slab, S, n, reinterpret_cast<bf16_t *>(out)); }
    else { auto k = cl_slab_reduce_kernel<float>; DLKA_LAUNCH(k, dim3(grid), dim3(256), 0, st, slab, S, n, reinterpret_cast<float *>(out)); }
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

}  // namespace dlka
