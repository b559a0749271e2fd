// Small-volume dense convolutions (K > 1 taps, split-operand contraction on the bf16 matrix cores) with the contraction split over the WAVES OF ONE
// WORKGROUP — the deterministic replacement of the tap split over gridDim.y whose partial sums met in global fp32 atomics (round 6).
//
//     out[m][n] = bias[n] + sum_tap sum_c A(m, tap, c) * Wp[tap][c][n]   (+ aux[m][n], epilogue 3)
//
// At the 16^3 / 8^3 / 4^3 stages of the D-LKA net (8192 / 1024 / 128 rows at B = 2) a row tiling alone leaves the chip empty, so rounds 1 - 5 split the
// (tap, 32-channel chunk) units over blockIdx.y and let the partial tiles meet in `atomicAdd` on a zero-filled output: the summation ORDER then depends on
// which workgroup arrives first — the offset-predict conv's output differed in the last bit from run to run, and a sample within rounding of a cell boundary
// changed cell (the reference's forward is im2col + addmm, deterministic: deform_conv_cuda.cu:95-123) — and the atomics were 20 - 40 % of these kernels' time
// (profiles/r09e_noatomic_ablation.txt).  Here a workgroup owns ONE 32-row x 32-column output tile; its KW waves (4 or 8) each contract a contiguous range of
// the units exactly as a wave of cl_conv_wave_kernel does (A rows by unconditional buffer loads, B records straight from the L2-resident prepared weights,
// two-stage register ring, no LDS and no barrier in the loop) and leave their 32 x 32 partial tile in LDS; after ONE barrier the tile is summed over the waves
// in wave order — a fixed order — and written once: no zero fill, no atomics, no fp32 staging buffer for bf16 storage.  The partial sums of different waves are
// added in the same order on every run and on every launch geometry of the same shape, so the result is bitwise reproducible.
#include <stdlib.h>

#include <atomic>

#include "cl_arow.h"
#include "dlka_kernels.h"

namespace dlka {

std::atomic<long> g_conv_kw_launches{0};   // dlka_conv_kw_launch_count (include/dlka.h): diagnostics; cl_deform_fwd.hip's workgroup-split launches count here too

// T: storage of the channels-last tensors (AMODE 0 `in`; OMODE 0 `out` / `aux`): float | bf16_t.  Planar tensors (AMODE 2 `in`, OMODE 1 `out`) are fp32.
// NT: 32-column tiles per workgroup (the A rows are fetched once for all of them; small volumes take 1 for the sake of the workgroup count).
template <int AMODE, int OMODE, int SPLIT, int KW, int NT, typename T>
__global__ __launch_bounds__(64 * KW) void cl_conv_kw_kernel(IgemmArgs p)
{
    constexpr bool A16 = AMODE == 0 && sizeof(T) == 2;
    constexpr int UF = SPLIT == 3 ? 48 : 32;      // floats of prepared weights per unit and column
    constexpr int NB = 2 * SPLIT * NT;            // B records (16 bytes) per lane and unit: [(part * 2 + mf) * NT + t]
    constexpr int RS = 65;                        // row stride of a wave's partial tile in LDS (floats): register r of lane l at r * RS + l
    // LF ("line-friendly" A fetch, channels-last fp32 rows): a "lane = row" load (ARow) touches 64 cache lines per instruction — ~256 address-unit cycles per unit and wave,
    // which is what bounded this kernel with one workgroup per CU (profiles/r10_notes.md).  Here a load instruction covers 8 WHOLE 128-byte rows (lane = (row of 8, 16-byte
    // piece), as the gather kernels of cl_gather.h do); the pieces go through a wave-private LDS tile [32 rows][36] into the MFMA A layout (lane (i, h): channels 16 h ..
    // 16 h + 15 of row i).  The tile shares its LDS with the wave's partial tile of the final reduction (used only behind the loop).
    constexpr bool LF = AMODE == 0 && sizeof(T) == 4;
    constexpr int AROW = 36;
    constexpr int WSM = (LF && 32 * AROW > 16 * RS) ? 32 * AROW : 16 * RS;
    __shared__ __attribute__((aligned(16))) float Red[KW][WSM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int mbase = blockIdx.x * 32;
    const int m = mbase + i;
    const bool row_ok = m < p.M;
    const int b = row_ok ? m / p.N : 0;
    const int v = row_ok ? m - b * p.N : 0;
    const int w0 = v % p.W, h0 = (v / p.W) % p.H, d0 = v / (p.W * p.H);
    const int n0 = blockIdx.z * (32 * NT);

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nchunk = p.CinP / 32;
    const int units = p.K * nchunk;
    const int ups = (units + KW - 1) / KW;        // units per wave (the last waves may get fewer, or none)
    const int unit_lo = wave * ups;
    const int unit_hi = min(units, unit_lo + ups);
    const BufRsrc rin = make_rsrc(p.in, AMODE == 2 ? (size_t)p.B * p.CinReal * p.N * 4 : (size_t)p.M * p.Cin * sizeof(T));
    const BufRsrc rw = make_rsrc(p.wp, (size_t)units * UF * p.NP * 4);
    const unsigned unit_bytes = (unsigned)(UF * p.NP) * 4u, seg_bytes = (unsigned)p.NP * 16u;
    const unsigned blane = (unsigned)(h * p.NP + n0 + i) * 16u;   // this lane's record inside segment (part, mf)

    ARow<AMODE, T> arow;
    float abuf[2][16];
    f32x4 bbuf[2][NB];
    // LF: this lane's four rows 8 e + rg and its piece pc; their byte offsets for the current tap (formed by "lane = row" and handed round by shuffles)
    const int rg = lane >> 3, pc = lane & 7;
    unsigned lf_ro[4] = {DLKA_OOB, DLKA_OOB, DLKA_OOB, DLKA_OOB};
    int lf_tap = -1;
    auto issue = [&](int unit, float *ad, f32x4 *bd) {
        int ck;
        const int tap = divmod_fast(unit, nchunk, ck);
        if (LF) {
            if (tap != lf_tap) {   // wave-uniform
                lf_tap = tap;
                int ti, tj, tk;
                tap_decode(tap, p.kw, p.kh, ti, tj, tk);
                const int zd = d0 + ti * p.dd - p.pd, zh = h0 + tj * p.dh - p.ph, zw = w0 + tk * p.dw - p.pw;
                const bool ok = row_ok & ((unsigned)zd < (unsigned)p.D) & ((unsigned)zh < (unsigned)p.H) & ((unsigned)zw < (unsigned)p.W);
                const unsigned mine = !ok ? DLKA_OOB : (unsigned)((b * p.N + (zd * p.H + zh) * p.W + zw) * p.Cin) * 4u;   // row i = lane & 31 (both half-waves agree)
#pragma unroll
                for (int e = 0; e < 4; ++e) lf_ro[e] = (unsigned)__shfl((int)mine, 8 * e + rg);
            }
            const unsigned cb = (unsigned)(ck * 32 + 4 * pc) * 4u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // the raw pieces ride in the ring as 16 floats: [4 e .. 4 e + 3] = the 16 bytes of row 8 e + rg, piece pc
                const f32x4 t = buf_load_f32x4(rin, lf_ro[e] == DLKA_OOB ? DLKA_OOB : lf_ro[e] + cb);
                ad[4 * e] = t[0]; ad[4 * e + 1] = t[1]; ad[4 * e + 2] = t[2]; ad[4 * e + 3] = t[3];
            }
        } else
        arow.fetch(p, rin, tap, ck, h, row_ok, b, v, d0, h0, w0, ad);
        const unsigned ub = (unsigned)unit * unit_bytes + blane;
#pragma unroll
        for (int part = 0; part < SPLIT; ++part)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int t = 0; t < NT; ++t) bd[(part * 2 + mf) * NT + t] = buf_load_f32x4(rw, ub + (unsigned)((part * 2 + mf) * 2) * seg_bytes + (unsigned)t * 512u);
    };
    auto compute = [&](const float *a_raw, const f32x4 *b_cur) {
        float a_lf[16];
        if (LF) {   // pieces -> wave-private tile -> this lane's 16 A values (LDS operations of one wave execute in order: a fence for the compiler is all it takes)
            float *At = Red[wave];
            wave_sync();
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x4 *>(At + (8 * e + rg) * AROW + 4 * pc) = f32x4{a_raw[4 * e], a_raw[4 * e + 1], a_raw[4 * e + 2], a_raw[4 * e + 3]};
            wave_sync();
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(At + i * AROW + 16 * h + 4 * e);
                a_lf[4 * e] = t[0]; a_lf[4 * e + 1] = t[1]; a_lf[4 * e + 2] = t[2]; a_lf[4 * e + 3] = t[3];
            }
        }
        const float *a_cur = LF ? a_lf : a_raw;
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            if (SPLIT == 3) {   // three-term operands, the six products above 2^-24: fp32-equivalent (cl_igemm.hip) — the FORWARD offset conv, whose output feeds floor()
                bf16x8 ahi, amid, alo;
                split3_bf16x8(a_cur + 8 * mf, ahi, amid, alo);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 bhi = __builtin_bit_cast(bf16x8, b_cur[(0 * 2 + mf) * NT + t]), bmid = __builtin_bit_cast(bf16x8, b_cur[(1 * 2 + mf) * NT + t]),
                                 blo = __builtin_bit_cast(bf16x8, b_cur[(2 * 2 + mf) * NT + t]);
                    acc[t] = mfma_32x32x16_bf16(alo, bhi, acc[t]);   // small terms first
                    acc[t] = mfma_32x32x16_bf16(ahi, blo, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(amid, bmid, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(amid, bhi, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(ahi, bmid, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(ahi, bhi, acc[t]);
                }
            } else {
                bf16x8 ahi, alo;
                if (A16) ahi = alo = bf16x8_from_words(a_cur + 4 * mf);   // raw bf16 rows (ARow): their own high term, no low term
                else if (AMODE == 2 && p.a_packed) unpack_split2x8(a_cur + 8 * mf, ahi, alo);
                else split_bf16x8(a_cur + 8 * mf, ahi, alo);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 bhi = __builtin_bit_cast(bf16x8, b_cur[(0 * 2 + mf) * NT + t]), blo = __builtin_bit_cast(bf16x8, b_cur[(1 * 2 + mf) * NT + t]);
                    if (!A16) acc[t] = mfma_32x32x16_bf16(alo, bhi, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(ahi, blo, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(ahi, bhi, acc[t]);
                }
            }
        }
    };
    // two-stage register ring: while unit u computes, unit u + 1 is in flight
    if (unit_lo < unit_hi) issue(unit_lo, abuf[0], bbuf[0]);
    for (int unit = unit_lo; unit < unit_hi; unit += 2) {
        if (unit + 1 < unit_hi) issue(unit + 1, abuf[1], bbuf[1]);
        compute(abuf[0], bbuf[0]);
        if (unit + 1 < unit_hi) {
            if (unit + 2 < unit_hi) issue(unit + 2, abuf[0], bbuf[0]);
            compute(abuf[1], bbuf[1]);
        }
    }

    // ---- the waves' partial tiles meet in LDS and are summed in WAVE ORDER (D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) ----
    float *mine = Red[wave];
    wave_sync();   // (LF: the wave's A tile, in the same LDS, has been read)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t) __syncthreads();   // the previous tile has been read
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[r * RS + lane] = acc[t][r];
        __syncthreads();
        // one output element per thread and pass: OMODE 0 (channels-last) walks the columns fastest, OMODE 1 (planar) the rows (= voxels): 128 contiguous bytes per store either way
        for (int e = tid; e < 32 * 32; e += 64 * KW) {
            const int row = OMODE == 1 ? (e & 31) : (e >> 5), col = OMODE == 1 ? (e >> 5) : (e & 31);
            const int hh = (row >> 2) & 1, rr = (row & 3) + 4 * (row >> 3);
            float val = 0.f;
#pragma unroll
            for (int w = 0; w < KW; ++w) val += Red[w][rr * RS + col + 32 * hh];
            const int mr = mbase + row, n = n0 + 32 * t + col;
            if (mr >= p.M || n >= p.Cout) continue;
            if (p.bias) val += p.bias[n];
            if (OMODE == 1) {
                const int bb = mr / p.N, vv = mr - bb * p.N;
                p.out[((long)bb * p.Cout + n) * p.N + vv] = val;
            } else {
                const long o = (long)mr * p.Cout + n;
                if (p.epi == 3) val += (sizeof(T) == 2 && p.aux_f32) ? p.aux[o] : act_load1(reinterpret_cast<const T *>(p.aux), o);
                act_store1(reinterpret_cast<T *>(p.out), o, val);
            }
        }
    }
}

// Would the K split of this contraction run inside the workgroup (cl_conv_kw_kernel)?  The C-ABI sequencing code asks BEFORE it decides on zero fills and fp32
// staging buffers: a contraction this returns true for is launched with splits = 1 and needs neither.
// volume: a 3-D volume (D > 1).  The 2-D nets' offset convs (D = 1: 25 / 49 taps, up to 384 columns, 4 704 / 18 816 rows at B = 24) keep the tap split: with this kernel the 2-D
// block measured 1328 against 1385 images/s (profiles/r10_notes.md) — many column tiles re-fetching "lane = row" A operands.
bool cl_conv_kw_applies(int amode, int omode, int split_bf16, int K, int epi, int NP, bool act_bf16, bool volume)
{
    static const bool off = [] { const char *e = getenv("DLKA_CONV_KW"); return e && e[0] == '0'; }();   // (A/B: 0 = the tap split over gridDim.y with fp32 atomics, rounds 1 - 5)
    if (off || !volume || K <= 1 || (epi != 0 && epi != 3) || NP % 32) return false;
    if (act_bf16) return split_bf16 == 2 && ((amode == 0 && omode == 1) || (amode == 2 && omode == 0));   // the two 27-tap convs of the bf16 token path
    if (split_bf16 == 3) return amode == 0 && (omode == 0 || omode == 1);
    if (split_bf16 == 2) return (amode == 0 && (omode == 0 || omode == 1)) || (amode == 2 && omode == 0);
    return false;   // DLKA_EXACT_FP32: cl_igemm_kernel's tap split (atomics) stays
}

int launch_cl_conv_kw(int amode, int omode, const IgemmArgs &a, hipStream_t st)
{
    if (!cl_conv_kw_applies(amode, omode, a.split_bf16, a.K, a.epi, a.NP, a.act_bf16 != 0, a.D > 1)) return DLKA_ERR_UNSUPPORTED;
    if ((long)a.K * (a.CinP / 32) * (a.split_bf16 == 3 ? 48 : 32) * a.NP * 4 >= (1l << 31)) return DLKA_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    const int nt_total = a.NP / 32, row_tiles = cdiv(a.M, 32);
    // column tiles per workgroup: as many as still leave >= 2048 waves of four-wave workgroups (the A rows are fetched once per workgroup), else one
    int nt = 1;
    for (int c = 4; c > 1; --c)
        if (nt_total % c == 0 && (long)row_tiles * (nt_total / c) * 4 >= 2048) { nt = c; break; }
    const int tiles = row_tiles * (nt_total / nt);
    const int kw = (nt > 1 || tiles * 4 >= 1024) ? 4 : 8;   // waves per tile: four where that fills the chip (16^3 and up); the 8^3 / 4^3 stages take eight
    dim3 grid(row_tiles, 1, nt_total / nt), block(64 * kw);
#define DLKA_KW1(AM, OM, SP, TT, KWV, NTV) { auto k = cl_conv_kw_kernel<AM, OM, SP, KWV, NTV, TT>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
#define DLKA_KW(AM, OM, SP, TT)                      \
    {                                                \
        if (kw == 8) DLKA_KW1(AM, OM, SP, TT, 8, 1)  \
        else if (nt == 1) DLKA_KW1(AM, OM, SP, TT, 4, 1) \
        else if (nt == 2) DLKA_KW1(AM, OM, SP, TT, 4, 2) \
        else if (nt == 3) DLKA_KW1(AM, OM, SP, TT, 4, 3) \
        else DLKA_KW1(AM, OM, SP, TT, 4, 4)          \
    }
    if (a.act_bf16) {
        if (amode == 0) DLKA_KW(0, 1, 2, bf16_t)
        else DLKA_KW(2, 0, 2, bf16_t)
    } else if (a.split_bf16 == 3) {
        if (omode == 1) DLKA_KW(0, 1, 3, float)
        else DLKA_KW(0, 0, 3, float)
    } else {
        if (amode == 0 && omode == 0) DLKA_KW(0, 0, 2, float)
        else if (amode == 0) DLKA_KW(0, 1, 2, float)
        else DLKA_KW(2, 0, 2, float)
    }
#undef DLKA_KW
#undef DLKA_KW1
    DLKA_CHECK_LAUNCH();
    g_conv_kw_launches.fetch_add(1, std::memory_order_relaxed);
    return DLKA_OK;
}

}  // namespace dlka

extern "C" long dlka_conv_kw_launch_count(void) { return dlka::g_conv_kw_launches.load(std::memory_order_relaxed); }
