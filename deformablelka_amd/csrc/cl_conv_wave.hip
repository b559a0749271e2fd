// Wave-granular implicit GEMM for the K > 1 dense convolutions whose contraction runs on the bf16 matrix cores with split
// operands (offset-predict conv forward / data gradient, the 3^3 convs of UnetResBlock):
//     out[m][n] = bias[n] + sum_tap sum_c A(m, tap, c) * Wp[tap][c][n]
//
// cl_igemm_kernel shares each 32 x NP weight chunk among the 4 waves of a workgroup through LDS: one barrier per (tap, chunk)
// unit.  With split operands a unit is only 6..36 MFMAs of 32 cycles, and the measured profile of those launches was per-unit
// latency, not arithmetic (offset-conv data gradient at 32^3: matrix cores 15 % busy, 30 % of wave cycles waiting, 87 us;
// profiles/archive/r01u_pmc_offc.txt).  Here every wave runs alone: its B operand comes straight from the L2-resident prepared weights —
// the split layout stores, per (unit, part, mf, h), NP records of 8 bf16, i.e. exactly one 16-byte load per MFMA B operand —
// and the A rows and B records of unit u+1 are in flight while unit u computes.  No LDS in the main loop, no barrier anywhere.
#include <stdlib.h>

#include "cl_arow.h"
#include "dlka_kernels.h"

namespace dlka {

// T: storage of a channels-last `in` (AMODE 0).  bf16 rows are their own high term: the two-term contraction drops the a_lo product.
// (bf16 storage is wired for the planar-output forward only — the offset-predict conv; outputs of the other modes stay fp32.)
template <int AMODE, int OMODE, int NT, int SPLIT, int DEPTH, typename T = float>
__global__ __launch_bounds__(256) void cl_conv_wave_kernel(IgemmArgs p)
{
    constexpr bool A16 = AMODE == 0 && sizeof(T) == 2;
    constexpr int NPB = NT * 32;
    constexpr int UF = SPLIT == 3 ? 48 : 32;      // floats of prepared weights per unit and column
    constexpr int NB = 2 * SPLIT * NT;            // B records (16 bytes) per lane and unit
    __shared__ float Tsm[OMODE == 1 ? 4 * 32 * 33 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int bx = DLKA_XCD_BX(p.xcd_nx);
    if (bx < 0) return;
    const int mbase = (bx * 4 + wave) * 32;
    const int m = mbase + i;
    const bool row_ok = m < p.M;
    const int b = row_ok ? m / p.N : 0;
    const int v = row_ok ? m - b * p.N : 0;
    const int w0 = v % p.W, h0 = (v / p.W) % p.H, d0 = v / (p.W * p.H);
    const int n0 = blockIdx.z * NPB;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nchunk = p.CinP / 32;
    const int unit_lo = blockIdx.y * p.units_per_split;
    const int unit_hi = min(p.K * nchunk, unit_lo + p.units_per_split);
    const BufRsrc rin = make_rsrc(p.in, AMODE == 2 ? (size_t)p.B * p.CinReal * p.N * 4 : (size_t)p.M * p.Cin * sizeof(T));
    const BufRsrc rw = make_rsrc(p.wp, (size_t)p.K * nchunk * UF * p.NP * 4);
    const unsigned unit_bytes = (unsigned)(UF * p.NP) * 4u, seg_bytes = (unsigned)p.NP * 16u;
    const unsigned blane = (unsigned)(h * p.NP + n0 + i) * 16u;   // this lane's record inside segment (part, mf)

    ARow<AMODE, T> arow;
    float abuf[DEPTH][16];
    f32x4 bbuf[DEPTH][NB];   // [(part * 2 + mf) * NT + t]
    // A rows and B records of one unit into a register set (all loads unconditional buffer loads)
#ifndef DLKA_ABLW   // -DDLKA_ABLW=bits: TIMING-ONLY ablations of the planar-input two-term kernel (wrong results): 1 no split arithmetic, 2 one MFMA per
#define DLKA_ABLW 0  // k-half instead of three, 4 no A fetch in the loop, 8 no weight-record fetch in the loop
#endif
    constexpr int ABL = (AMODE == 2 && SPLIT == 2 && NT == 1) ? DLKA_ABLW : 0;
    bool first_issue = true;
    auto issue = [&](int unit, float *ad, f32x4 *bd) {
        int ck, tap;
#ifdef DLKA_CK_OUTER
        if (AMODE == 2) { ck = unit / p.K; tap = unit - ck * p.K; }   // planar input: plane chunks OUTER, taps inner (see the launcher)
        else
#endif
        tap = divmod_fast(unit, nchunk, ck);
        if (!(ABL & 4) || first_issue) arow.fetch(p, rin, tap, ck, h, row_ok, b, v, d0, h0, w0, ad);
        const unsigned ub = (unsigned)(tap * nchunk + ck) * unit_bytes + blane;
        if (!(ABL & 8) || first_issue) {
#pragma unroll
        for (int part = 0; part < SPLIT; ++part)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    bd[(part * 2 + mf) * NT + t] = buf_load_f32x4(rw, ub + (unsigned)((part * 2 + mf) * 2) * seg_bytes + (unsigned)t * 512u);
        }
        if (ABL) first_issue = false;
    };
    auto compute = [&](const float *a_cur, const f32x4 *b_cur) {
#ifdef DLKA_IGLP
        __builtin_amdgcn_iglp_opt(DLKA_IGLP);   // (experiment: scripts/build_variant.sh NAME -DDLKA_IGLP=0|1)
#endif

#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            if (SPLIT == 3) {
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
                else if (ABL & 1) { ahi = bf16x8_from_words(a_cur + 8 * mf); alo = bf16x8_from_words(a_cur + 8 * mf + 4); }
                else split_bf16x8(a_cur + 8 * mf, ahi, alo);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 bhi = __builtin_bit_cast(bf16x8, b_cur[(0 * 2 + mf) * NT + t]), blo = __builtin_bit_cast(bf16x8, b_cur[(1 * 2 + mf) * NT + t]);
                    if (!A16) acc[t] = mfma_32x32x16_bf16(alo, bhi, acc[t]);
                    if (ABL & 2) continue;
                    acc[t] = mfma_32x32x16_bf16(ahi, blo, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(ahi, bhi, acc[t]);
                }
            }
        }
    };
    // DEPTH-stage register ring: while unit u computes, units u+1 .. u+DEPTH-1 are in flight.  (One unit is only a few hundred cycles of
    // MFMA + split arithmetic against ~1-2 us of L2 latency; with a one-deep prefetch and 2 waves per SIMD the kernel sat at half speed.)
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s)
        if (unit_lo + s < unit_hi) issue(unit_lo + s, abuf[s], bbuf[s]);
    for (int unit = unit_lo; unit < unit_hi; unit += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            if (unit + s < unit_hi) {   // uniform
#ifdef DLKA_CW_UNCOND   // (experiment: unconditional ring loads — behind the last unit they re-read it — instead of a conditional load merged by register copies)
                issue(min(unit + s + DEPTH - 1, unit_hi - 1), abuf[(s + DEPTH - 1) % DEPTH], bbuf[(s + DEPTH - 1) % DEPTH]);
#else
                if (unit + s + DEPTH - 1 < unit_hi) issue(unit + s + DEPTH - 1, abuf[(s + DEPTH - 1) % DEPTH], bbuf[(s + DEPTH - 1) % DEPTH]);
#endif
                compute(abuf[s], bbuf[s]);
            }
        }
    }

    // ---- epilogue (same contract as cl_igemm_kernel): D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
    const bool split = gridDim.y > 1;
    if (OMODE == 1) {   // planar output [B][Cout][N]: transpose each tile through a wave-private LDS tile so that lanes run over voxels
        float *Tt = Tsm + wave * (32 * 33);
        const bool rok = row_ok;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            wave_sync();
#pragma unroll
            for (int r = 0; r < 16; ++r) Tt[((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + i] = acc[t][r];
            wave_sync();
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) {
                const int col = 2 * cc + h, n = n0 + t * 32 + col;
                if (n >= p.Cout) continue;   // uniform per half-wave
                float val = Tt[i * 33 + col];
                if (p.bias && blockIdx.y == 0) val += p.bias[n];
                if (!rok) continue;
                float *dst = p.out + ((long)b * p.Cout + n) * p.N + v;
                if (split) atomicAdd(dst, val);
                else *dst = val;
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + t * 32 + i;
        if (n >= p.Cout) continue;
        const float bv = (p.bias && blockIdx.y == 0) ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = mbase + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mr >= p.M) continue;
            float val = acc[t][r] + bv;
            const long o = (long)mr * p.Cout + n;
            if (split) {
                if (p.epi == 3 && blockIdx.y == 0) val += p.aux[o];
                atomicAdd(p.out + o, val);
            } else if (p.epi == 3) {
                p.out[o] = val + p.aux[o];
            } else {
                p.out[o] = val;
            }
        }
    }
}

// Split-operand convs with K > 1, epilogues 0 and 3.  Returns DLKA_ERR_UNSUPPORTED for anything else (the caller then uses cl_igemm_kernel).
int launch_cl_conv_wave(int amode, int omode, const IgemmArgs &a, int splits, hipStream_t st)
{
    constexpr bool off = false;
    if (off || a.K <= 1 || (a.split_bf16 != 2 && a.split_bf16 != 3) || (a.epi != 0 && a.epi != 3)) return DLKA_ERR_UNSUPPORTED;
    if (amode != 0 && amode != 2) return DLKA_ERR_UNSUPPORTED;
    if ((long)a.K * (a.CinP / 32) * (a.split_bf16 == 3 ? 48 : 32) * a.NP * 4 >= (1l << 31)) return DLKA_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    const int NT_total = a.NP / 32;
    // column tiles per wave / prefetch depth.  DLKA_CONV_WAVE_CFG=<NT><DEPTH> overrides for tuning (e.g. 32 = NT 3, depth 2).
    constexpr int cfg = 0;
    int NT = 1, DEPTH = 2;
    if (cfg) { NT = cfg / 10; DEPTH = cfg % 10; }
    else {
        // Measured (profiles/archive/r01w_conv_wave_cfg.txt, us, this kernel vs cl_igemm_kernel): offset conv forward (three-term, 96 columns) 128 vs 81 at
        // C=32/32^3 — one column tile per wave re-reads the A rows three times through the "lane = row" loads, which cost one L1 access per row —
        // but 46 vs 52 at C=64/16^3 and 28 vs 39 at C=128/8^3; data gradient (two-term) 88 vs 104 at C=32/32^3, equal elsewhere.  A deeper
        // register ring (3 stages) changed nothing: these launches are not waiting on latency.
        if (a.split_bf16 == 3 && a.M > 16384) return DLKA_ERR_UNSUPPORTED;
        if (a.split_bf16 == 2 && NT_total != 1 && !a.act_bf16) return DLKA_ERR_UNSUPPORTED;
        // bf16 activations: the offset-predict conv forward only, where the fp32 path takes this kernel too (its three-term variant)
        if (a.act_bf16 && !(amode == 0 && omode == 1 && a.split_bf16 == 2 && a.M <= 16384)) return DLKA_ERR_UNSUPPORTED;
    }
    if (a.act_bf16 && (NT != 1 || DEPTH != 2)) return DLKA_ERR_UNSUPPORTED;
    if (NT < 1 || NT > 3 || NT_total % NT || (DEPTH != 2 && DEPTH != 3)) return DLKA_ERR_UNSUPPORTED;
    dim3 grid(cdiv(a.M, 128), splits, NT_total / NT), block(256);
    IgemmArgs ax = a;
    ax.xcd_nx = 0;
    if (xcd_swizzle_enabled() && grid.x >= (unsigned)xcd_min_blocks()) { ax.xcd_nx = (int)grid.x; grid.x = xcd_grid(ax.xcd_nx); }
#define DLKA_CW(AM, OM, NTV, SP, DP)                                        \
    {                                                                       \
        auto k = cl_conv_wave_kernel<AM, OM, NTV, SP, DP>;                  \
        DLKA_LAUNCH(k, grid, block, 0, st, ax);                      \
    }
#define DLKA_CW_NT(AM, OM, SP)                                              \
    {                                                                       \
        if (NT == 1 && DEPTH == 3) DLKA_CW(AM, OM, 1, SP, 3)                \
        else if (NT == 1) DLKA_CW(AM, OM, 1, SP, 2)                         \
        else if (NT == 2 && DEPTH == 3) DLKA_CW(AM, OM, 2, SP, 3)           \
        else if (NT == 2) DLKA_CW(AM, OM, 2, SP, 2)                         \
        else if (DEPTH == 2) DLKA_CW(AM, OM, 3, SP, 2)                      \
        else return DLKA_ERR_UNSUPPORTED;                                   \
    }
    if (a.act_bf16) {
        auto k = cl_conv_wave_kernel<0, 1, 1, 2, 2, bf16_t>;
        DLKA_LAUNCH(k, grid, block, 0, st, ax);
    } else if (a.split_bf16 == 3) {
        if (amode == 0 && omode == 1) DLKA_CW_NT(0, 1, 3)
        else if (amode == 0 && omode == 0) DLKA_CW_NT(0, 0, 3)
        else return DLKA_ERR_UNSUPPORTED;
    } else {
        if (amode == 0 && omode == 0) DLKA_CW_NT(0, 0, 2)
        else if (amode == 0 && omode == 1) DLKA_CW_NT(0, 1, 2)
        else if (amode == 2 && omode == 0) DLKA_CW_NT(2, 0, 2)
        else return DLKA_ERR_UNSUPPORTED;
    }
#undef DLKA_CW_NT
#undef DLKA_CW
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

}  // namespace dlka
