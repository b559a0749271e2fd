// Channels-last backward of the 3-D deformable convolution w.r.t. offsets and input — third generation.
//
// What the reference does (3D/dcn/src/cuda/deform_conv_cuda.cu:226-251): columns = W^T * grad_out (at::mm into a
// 27*C x B*N buffer in HBM), deformable_col2im_coord_gpu_kernel (grad_offset, deform_im2col_cuda.cuh:336-405) and
// deformable_col2im_gpu_kernel (grad_input, cuh:267-334: one global fp32 atomicAdd per column element and corner).
//
// Measured on MI355X (profiles/archive/r01d_accum_ubench.txt): global fp32 atomics top out at ~3.3e11 lane-atomics/s whatever
// the locality (1.0-1.6 ms for the 4.5e8 corner updates of one C=32, 32^3, B=2 call), LDS *fp32* atomics (ds_add_f32)
// are no faster (0.33 lanes/clk/CU), but LDS *fp64* atomics (ds_add_f64) run at 6.7 lanes/clk/CU — 20x.  Hence:
//
//   cl_deform_goff2_kernel  grad_offset only: a pure gather, no atomics.  MFMA orientation D[channel][voxel]
//                           (A = W_tap^T from LDS, B = grad_out rows), corner rows gathered in the line-friendly layout of
//                           cl_gather.h; optionally stores the interpolated samples for the weight gradient.
//   cl_deform_gx_kernel     grad_input: a workgroup owns a brick of OUTPUT voxels and a 4-channel slice; the scatter
//                           accumulates in an fp64 LDS window (brick + 3-voxel halo) with ds_add_f64 — the sum is exact
//                           to fp64 and independent of the order —; MFMA rows = 8 taps x 4 channels, columns = voxels.
//                           Corners beyond the halo (|tap + offset| > ~2 voxels outside the brick) fall back to global atomics.
//   cl_deform_gx_gather_kernel   every input voxel sums the <= few windows that cover it (plain loads, fixed order) and
//                           adds the result to grad_input: no flush atomics.
#include <math.h>
#include <stdlib.h>

#include <atomic>
#include "cl_args.h"
#include "cl_gather.h"
#include "dlka_kernels.h"

namespace dlka {

namespace {

constexpr int HALO = 3;
constexpr int CS = 4;     // channels per grad_input slice
constexpr int TG = 8;     // taps per MFMA row group (TG * CS = 32 MFMA rows)

}  // namespace

// =====================================================================================================================
// grad_offset
// =====================================================================================================================
// ---------------------------------------------------------------------------------------------------------------------
// grad_offset with the line-friendly gather of cl_gather.h.
//   goff_axis[m, tap] = sum_c Col[m, tap, c] * Dax[m, tap, c],   Dax = sum_q dcoef_axis(q) * x_q   (cuh:111-190)
// Per (32-voxel tile, tap, 32-channel chunk) a wave
//   1. runs the Col chunk on the matrix cores (orientation D[channel][voxel] as above; the corner loads issued one unit
//      earlier land meanwhile),
//   2. in the gather layout (lane = (row of 8, 16-byte piece of 8): every load covers 8 whole 128-byte rows) turns the
//      8 corner pieces of each of its 4 rows into the three derivative samples and writes them to wave-private LDS tiles,
//   3. back in the MFMA layout (lane = voxel, 16 channels in registers) reads its 16 channels of the three tiles and
//      dots them with Col: 48 FMAs and one cross-half add per axis — no cross-lane reduction tree, no atomics.
// (An intermediate version kept Col in the gather layout and reduced 8-lane groups with DPP; it needed 435 registers,
//  ran one wave per SIMD and was no faster than the first kernel.)
// ---------------------------------------------------------------------------------------------------------------------
// SAMP: also store the samples S[tap][m][c] (DeformBwdArgs::samp) for the weight gradient (the launcher picks NKC_REG by measurement).
// B16 (T = bf16_t, p.wp16 set): Col on the bf16 matrix cores.  The grad_out row of a voxel is bf16 in memory, 16 contiguous bytes ARE the MFMA B operand of a
// k-group (channels 16 h + 8 mf .. + 7: raw words, no conversion); the weights come as two-term bf16 records (hi + lo) that each lane reads from L2 with 16-byte
// loads — no LDS weight tile, no workgroup barrier per (tap, chunk) stage; 4 v_mfma_f32_32x32x16_bf16 per stage instead of 16 v_mfma_f32_32x32x2_f32
// (128 against 1024 matrix-pipe cycles).  The D layout is the same, so everything behind the contraction is unchanged.
// B16 with T = float (round 4): the same on fp32 activations with a TWO-TERM split of the grad_out row (g = hi + lo, formed once per tile in registers — the
// row is reused by every tap) and the products  W_hi G_hi + W_lo G_hi + W_hi G_lo : each bf16 x bf16 product is exact in fp32, the accumulator is fp32, so Col
// is reproduced to ~3 * 2^-18 (1.1e-5 worst case, zero-mean) — the rule DESIGN 4.7 applies to the offset conv's gradient contractions; 6 MFMAs of 32 cycles
// per stage instead of 16 of 64.  Gradients are held to 1e-3; the forward pass keeps the exact fp32-input MFMA.
template <int NKC_REG, typename T = float, bool SAMP = false, bool B16 = false>   // T: storage of `in` and `g` (channels-last); offsets / grad_offset are fp32 planar
__global__ __launch_bounds__(256, 2) void cl_deform_goff2_kernel(DeformBwdArgs p, int taps_per_block)
{
    constexpr bool SPL = B16 && sizeof(T) == 4;   // fp32 activations, two-term split of grad_out
    constexpr unsigned XB = sizeof(T);
    const T *gin = reinterpret_cast<const T *>(p.g);
    constexpr int SROW = 36;
    __shared__ __attribute__((aligned(16))) float Bs2[B16 ? 1 : 2][B16 ? 4 : 32 * 32];          // W[tap][co chunk][ci chunk], double buffered (fp32-input MFMA only)
    __shared__ __attribute__((aligned(16))) float Tsm[4][3][32 * SROW];     // per wave: derivative-sample tiles (d, h, w)
    __shared__ __attribute__((aligned(16))) float Dsm[4][32 * GATHER_DESC_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    using GG = GatherGeom<T>;
    const int gr = lane >> GG::PSHIFT, gp = lane & ((1 << GG::PSHIFT) - 1);
    const int bx = DLKA_XCD_BX(p.xcd_nx);
    if (bx < 0) return;
    const int mbase = (bx * 4 + wave) * 32;
    const int m = mbase + j;
    const bool row_ok = m < p.M;
    const int b = row_ok ? m / p.N : 0;
    const int v = row_ok ? m - b * p.N : 0;
    const int w0 = v % p.W, h0 = (v / p.W) % p.H, d0 = v / (p.W * p.H);
    const int HW = p.H * p.W, rowbytes = p.C * XB;
    const int ncc = p.C / 32, nkc = p.CoutP / 32;
    const int tap_lo = blockIdx.y * taps_per_block, tap_hi = min(p.K, tap_lo + taps_per_block);
    // blockIdx.z: slice of the 32-channel input chunks (tiny volumes only: the partial dots then meet in atomics on a zeroed goff)
    const int cc_lo = blockIdx.z * p.cc_per_block, cc_hi = min(ncc, cc_lo + p.cc_per_block), ncb = cc_hi - cc_lo;
    const BufRsrc rin = make_rsrc(p.in, (size_t)p.M * p.C * XB);
    float *Tt = &Tsm[wave][0][0], *Dt = Dsm[wave];
    // samples for the weight gradient: S[tap][m][c], this lane's rows mbase + RPI*g + gr, channels PE*gp .. of the current chunk
    // (S has the storage type of the activations: fp32, or bf16 on the DLKA_BF16 path)
    const BufRsrc rsamp = make_rsrc(p.samp, SAMP ? (size_t)p.K * p.M * p.C * XB : 0);
    const unsigned samp_v0 = (unsigned)((mbase + gr) * p.C + GG::PE * gp) * XB;
    const bool tile_full = mbase + 32 <= p.M;   // wave-uniform: only the last tile of a ragged M needs per-row checks

    if (p.goff_cpad && blockIdx.y == 0 && blockIdx.z == 0 && h == 0 && row_ok) {   // the zero planes 3K .. goff_cpad-1 of the packed layout
        float *dst = p.goff + ((long)b * p.goff_cpad + 3 * p.K) * p.N + v;
        for (int c = 3 * p.K; c < p.goff_cpad; ++c, dst += p.N) *dst = 0.f;
    }
    // B16: this lane's grad_out operands — [kc][mf] = channels kc * 32 + 16 h + 8 mf .. + 7 of voxel j: the raw words of a bf16 row, or (SPL) the
    // hi term of an fp32 row with the lo term in graw_lo
    f32x4 graw[(B16 && NKC_REG > 0) ? NKC_REG : 1][2], graw_lo[(SPL && NKC_REG > 0) ? NKC_REG : 1][2];
    auto load_graw = [&](int kc, f32x4 out[2], f32x4 out_lo[2]) {
        const bool okg = row_ok && kc < nkc && kc * 32 + 16 * h < p.Cout;
        if (SPL) {
            const float *gp_ = reinterpret_cast<const float *>(p.g) + (okg ? (long)m * p.Cout + kc * 32 + 16 * h : 0);
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                const f32x4 t0 = *reinterpret_cast<const f32x4 *>(gp_ + 8 * mf), t1 = *reinterpret_cast<const f32x4 *>(gp_ + 8 * mf + 4);
                const float v8[8] = {okg ? t0[0] : 0.f, okg ? t0[1] : 0.f, okg ? t0[2] : 0.f, okg ? t0[3] : 0.f, okg ? t1[0] : 0.f, okg ? t1[1] : 0.f, okg ? t1[2] : 0.f, okg ? t1[3] : 0.f};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned short h0 = bf16_bits(v8[2 * w]), h1 = bf16_bits(v8[2 * w + 1]);
                    const unsigned short l0 = bf16_bits(v8[2 * w] - bf16_value(h0)), l1 = bf16_bits(v8[2 * w + 1] - bf16_value(h1));
                    out[mf][w] = __uint_as_float((unsigned)h0 | ((unsigned)h1 << 16));
                    out_lo[mf][w] = __uint_as_float((unsigned)l0 | ((unsigned)l1 << 16));
                }
            }
            return;
        }
        const bf16_t *gp_ = reinterpret_cast<const bf16_t *>(p.g) + (okg ? (long)m * p.Cout + kc * 32 + 16 * h : 0);
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(gp_ + 8 * mf);
            out[mf][0] = okg ? t[0] : 0.f; out[mf][1] = okg ? t[1] : 0.f; out[mf][2] = okg ? t[2] : 0.f; out[mf][3] = okg ? t[3] : 0.f;
        }
    };
    if (B16 && NKC_REG > 0) {
#pragma unroll
        for (int kc = 0; kc < NKC_REG; ++kc) load_graw(kc, graw[kc], graw_lo[SPL ? kc : 0]);
    }
    float greg[(NKC_REG > 0 && !B16) ? NKC_REG : 1][16];
    if (NKC_REG > 0 && !B16) {
#pragma unroll
        for (int kc = 0; kc < NKC_REG; ++kc) {
            const bool okg = row_ok && kc < nkc && kc * 32 + 16 * h < p.Cout;
            const long gi = okg ? (long)m * p.Cout + kc * 32 + 16 * h : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 t = act_load4(gin, gi + 4 * e);
                greg[kc][4 * e] = okg ? t[0] : 0.f; greg[kc][4 * e + 1] = okg ? t[1] : 0.f;
                greg[kc][4 * e + 2] = okg ? t[2] : 0.f; greg[kc][4 * e + 3] = okg ? t[3] : 0.f;
            }
        }
    }

    GatherPiece<T> xr[GG::NG][8];   // corner pieces of the unit in flight
    float onx[3] = {0.f, 0.f, 0.f};   // offsets of the next tap to describe, loaded a whole tap ahead
    auto load_offsets = [&](int tap) {
        const float *op = p.off + ((long)b * 3 * p.K + 3 * tap) * p.N + v;
        onx[0] = op[0]; onx[1] = op[p.N]; onx[2] = op[2 * (long)p.N];
    };
    auto describe = [&](int tap) {
        int ti, tj, tk;
        tap_decode(tap, p.kw, p.kh, ti, tj, tk);
        wave_sync();
        if (h == 0) {
            RowDesc r;
            r.base = 0; r.okm = 0; r.ld = r.lh = r.lw = 0.f;
            if (row_ok) r = gather_describe3(onx[0], onx[1], onx[2], p.N, b, d0 + ti * p.dd - p.pd, h0 + tj * p.dh - p.ph, w0 + tk * p.dw - p.pw, p.D, p.H, p.W);
            gather_publish(Dt, j, r, rowbytes);
        }
        wave_sync();   // (the descriptions stay in the LDS table; issue() and the interpolation read their rows back when they need them —
                       //  held in registers across the MFMA phase they cost 20 VGPRs the kernel does not have)
    };
    auto issue = [&](int cc) {
        const unsigned cbyte = (unsigned)(cc * 32 + GG::PE * gp) * XB;
#pragma unroll
        for (int g = 0; g < GG::NG; ++g) {
            const RowLook r = gather_lookup(Dt, GG::RPI * g + gr);
#pragma unroll
            for (int q = 0; q < 8; ++q) xr[g][q] = gather_load<T>(rin, gather_offset(r, q, HW, p.W, rowbytes, cbyte));
        }
    };

    // weight tile of stage s = (tap, cc, kc) in flight in a register while the previous stage computes
    const int nstage = (tap_hi - tap_lo) * ncb * nkc;
    f32x4 wreg;
    f32x4 wrec[B16 ? 4 : 1];   // B16: this lane's four records (hi / lo x k-group) of the stage in flight: [part * 2 + mf]
    auto load_w = [&](int s) {
        int kc, ccr;
        const int sq = divmod_fast(s, nkc, kc);
        const int tq = divmod_fast(sq, ncb, ccr);
        const int cc = cc_lo + ccr, tap = tap_lo + tq;
        if (B16) {
            const float *src = p.wp16 + ((long)tap * p.CoutP + kc * 32) * p.C;   // unit (tap, kc): [part][mf][h][C][8 bf16]
#pragma unroll
            for (int q = 0; q < 4; ++q) wrec[q] = *reinterpret_cast<const f32x4 *>(src + ((long)(q * 2 + h) * p.C + cc * 32 + j) * 4);
            return;
        }
        const int rr = tid >> 3, c4 = tid & 7;
        wreg = *(reinterpret_cast<const f32x4 *>(p.wp + ((long)tap * p.CoutP + kc * 32 + rr) * p.C + cc * 32) + c4);
    };
    int stage = 0;
    if (tap_lo < tap_hi) {
        load_w(0);
        load_offsets(tap_lo);
        describe(tap_lo);
        issue(cc_lo);
        if (tap_lo + 1 < tap_hi) load_offsets(tap_lo + 1);
    }
    for (int tap = tap_lo; tap < tap_hi; ++tap) {
        float gd = 0.f, gh = 0.f, gw = 0.f;
        for (int cc = cc_lo; cc < cc_hi; ++cc) {
            // ---- 1. Col chunk on the matrix cores ----
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 1
            for (int kc = 0; kc < nkc; ++kc) {
                if (B16) {
                    bf16x8 wa[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float w4[4] = {wrec[q][0], wrec[q][1], wrec[q][2], wrec[q][3]}; wa[q] = bf16x8_from_words(w4); }
                    ++stage;
                    if (stage < nstage) load_w(stage);
                    f32x4 gl2[2], gl2_lo[2];
                    if (NKC_REG == 0) load_graw(kc, gl2, gl2_lo);
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) {
                        const f32x4 gsel = NKC_REG == 0 ? gl2[mf] : (NKC_REG == 1 ? graw[0][mf] : (kc == 0 ? graw[0][mf] : graw[NKC_REG > 1 ? 1 : 0][mf]));
                        const float g4[4] = {gsel[0], gsel[1], gsel[2], gsel[3]};
                        const bf16x8 gb = bf16x8_from_words(g4);
                        acc = mfma_32x32x16_bf16(wa[mf], gb, acc);        // hi term of the weights
                        acc = mfma_32x32x16_bf16(wa[2 + mf], gb, acc);    // lo term
                        if (SPL) {                                        // ... and W_hi x the lo term of grad_out
                            const f32x4 lsel = NKC_REG == 0 ? gl2_lo[mf] : (NKC_REG == 1 ? graw_lo[0][mf] : (kc == 0 ? graw_lo[0][mf] : graw_lo[(SPL && NKC_REG > 1) ? 1 : 0][mf]));
                            const float l4[4] = {lsel[0], lsel[1], lsel[2], lsel[3]};
                            acc = mfma_32x32x16_bf16(wa[mf], bf16x8_from_words(l4), acc);
                        }
                    }
                    continue;
                }
                float *Bs = Bs2[stage & 1];
                reinterpret_cast<f32x4 *>(Bs)[tid] = wreg;
                ++stage;
                if (stage < nstage) load_w(stage);
                float gl[16];
                if (NKC_REG == 0) {
                    const bool okg = row_ok && kc * 32 + 16 * h < p.Cout;
                    const long gi = okg ? (long)m * p.Cout + kc * 32 + 16 * h : 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x4 t = act_load4(gin, gi + 4 * e);
                        gl[4 * e] = okg ? t[0] : 0.f; gl[4 * e + 1] = okg ? t[1] : 0.f; gl[4 * e + 2] = okg ? t[2] : 0.f; gl[4 * e + 3] = okg ? t[3] : 0.f;
                    }
                }
                __syncthreads();   // this stage's tile staged; the other buffer (read one stage ago) is free for the next
                const float *arow = Bs + (16 * h) * 32 + j;
                if (NKC_REG == 1) {
#pragma unroll
                    for (int st = 0; st < 16; ++st) acc = mfma_32x32x2(arow[st * 32], greg[0][st], acc);
                } else if (NKC_REG == 2) {
                    if (kc == 0) {
#pragma unroll
                        for (int st = 0; st < 16; ++st) acc = mfma_32x32x2(arow[st * 32], greg[0][st], acc);
                    } else {
#pragma unroll
                        for (int st = 0; st < 16; ++st) acc = mfma_32x32x2(arow[st * 32], greg[NKC_REG > 1 ? 1 : 0][st], acc);
                    }
                } else {
#pragma unroll
                    for (int st = 0; st < 16; ++st) acc = mfma_32x32x2(arow[st * 32], gl[st], acc);
                }
            }
            // ---- 2. derivative samples of this lane's 4 gather rows -> LDS tiles (SAMP: and the sample itself -> S) ----
            // (a separate pass for the samples was tried to shorten live ranges: the scheduler interleaved it anyway, 700 bytes of spills)
            wave_sync();   // previous tiles consumed
#pragma unroll
            for (int g = 0; g < GG::NG; ++g) {
                const RowLook rdg = gather_lookup(Dt, GG::RPI * g + gr);
                const bool srow = SAMP && (tile_full || mbase + GG::RPI * g + gr < p.M);
                f32x4 s4_lo = {0.f, 0.f, 0.f, 0.f};   // bf16 storage: the piece's first four samples wait for the other four (one 16-byte store)
#pragma unroll
                for (int v = 0; v < GG::PE / 4; ++v) {
                    // SEPARABLE trilinear sample + its three coordinate derivatives (cuh:111-190), axis by axis: w, then h, then d.  The per-corner
                    // form (8 corners x (3 derivative + 1 sample) fmas, plus 32 coefficient products per row and lane) was ~45 % of this kernel's
                    // vector instructions, and the kernel is bound by their issue (scripts/isa_loop_mix.py); 22 lerps / differences do the same:
                    //   along w:  sw(cd,ch) = x0 + lw (x1 - x0),  dq(cd,ch) = x1 - x0
                    //   along h:  shw(cd) = lerp_h sw,  eh(cd) = sw(cd,1) - sw(cd,0),  ew(cd) = lerp_h dq
                    //   along d:  S = lerp_d shw,  dS/dd = shw(1) - shw(0),  dS/dh = lerp_d eh,  dS/dw = lerp_d ew
                    // Dropped corners (outside the volume, or the whole sample outside the guard) arrive as zeros from the range-checked loads.
                    f32x4 sw[4], dq[4];
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr) {
                        const f32x4 x0 = xr[g][2 * pr].get(v), x1 = xr[g][2 * pr + 1].get(v);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { dq[pr][e] = x1[e] - x0[e]; sw[pr][e] = fmaf(rdg.lw, dq[pr][e], x0[e]); }
                    }
                    f32x4 shw[2], eh[2], ew[2];
#pragma unroll
                    for (int cd = 0; cd < 2; ++cd)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            eh[cd][e] = sw[2 * cd + 1][e] - sw[2 * cd][e];
                            shw[cd][e] = fmaf(rdg.lh, eh[cd][e], sw[2 * cd][e]);
                            ew[cd][e] = fmaf(rdg.lh, dq[2 * cd + 1][e] - dq[2 * cd][e], dq[2 * cd][e]);
                        }
                    f32x4 dd, dh, dw, s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dd[e] = shw[1][e] - shw[0][e];
                        if (SAMP) s4[e] = fmaf(rdg.ld, dd[e], shw[0][e]);
                        dh[e] = fmaf(rdg.ld, eh[1][e] - eh[0][e], eh[0][e]);
                        dw[e] = fmaf(rdg.ld, ew[1][e] - ew[0][e], ew[0][e]);
                    }
                    float *dst = Tt + (GG::RPI * g + gr) * SROW + GG::PE * gp + 4 * v;
                    *reinterpret_cast<f32x4 *>(dst) = dd;
                    *reinterpret_cast<f32x4 *>(dst + 32 * SROW) = dh;
                    *reinterpret_cast<f32x4 *>(dst + 2 * 32 * SROW) = dw;
                    if (SAMP && XB == 4) {
                        const unsigned so = (unsigned)((tap * p.M + GG::RPI * g) * p.C + cc * 32 + 4 * v);   // element offset beyond this lane's (row, piece)
                        if (p.samp_f16) buf_store_f16x4(rsamp, srow ? (samp_v0 >> 1) + so * 2u : DLKA_OOB, s4);   // (uniform) halves: samp_v0 is a byte offset of fp32 elements
                        else buf_store_f32x4(rsamp, srow ? samp_v0 + so * 4u : DLKA_OOB, s4);
                    }
                    if (SAMP && XB == 2) { if (v == 0) s4_lo = s4; else buf_store_bf16x8(rsamp, srow ? samp_v0 + (unsigned)((tap * p.M + GG::RPI * g) * p.C + cc * 32) * 2u : DLKA_OOB, s4_lo, s4); }
                }
            }
            // the next unit's corner loads go out now: in flight under the dots below, the next staging and the next MFMAs
            if (cc + 1 < cc_hi) issue(cc + 1);
            else if (tap + 1 < tap_hi) { describe(tap + 1); issue(cc_lo); }
            wave_sync();
            // ---- 3. dot with Col in the MFMA layout: acc[r] = Col[voxel j][channel (r&3) + 8*(r>>2) + 4h] ----
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float *src = Tt + j * SROW + 8 * g4 + 4 * h;
                const f32x4 td = *reinterpret_cast<const f32x4 *>(src), th = *reinterpret_cast<const f32x4 *>(src + 32 * SROW),
                            tw = *reinterpret_cast<const f32x4 *>(src + 2 * 32 * SROW);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gd = fmaf(acc[4 * g4 + e], td[e], gd); gh = fmaf(acc[4 * g4 + e], th[e], gh); gw = fmaf(acc[4 * g4 + e], tw[e], gw);
                }
            }
        }
        gd += __shfl_xor(gd, 32);   // the two halves hold complementary channel sets of the same voxel
        gh += __shfl_xor(gh, 32);
        gw += __shfl_xor(gw, 32);
        if (h == 0 && row_ok) {
            float *dst = p.goff + ((long)b * (p.goff_cpad ? p.goff_cpad : 3 * p.K) + 3 * tap) * p.N + v;
            if (gridDim.z > 1) { atomicAdd(dst, gd); atomicAdd(dst + p.N, gh); atomicAdd(dst + 2 * (long)p.N, gw); }
            else if (p.goff_cpad) { dst[0] = pack_split2(gd); dst[p.N] = pack_split2(gh); dst[2 * (long)p.N] = pack_split2(gw); }
            else { dst[0] = gd; dst[p.N] = gh; dst[2 * (long)p.N] = gw; }
        }
        if (tap + 2 < tap_hi) load_offsets(tap + 2);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// grad_offset with 16-row waves (round 3) for the stage the step spends most of its time in: C = C_out = 32, all taps in one workgroup.
// The kernel above runs two waves per SIMD at 243 registers and s_memtime stamps show its (tile, tap) unit as one serial chain — weights to LDS +
// barrier 1.2 k ticks | 16 MFMAs 1.2 k | interpolation + tile stores 3.0 k | description 0.8 k | 32 corner loads 2.1 k | tile reads + dots 2.1 k —
// with nothing to overlap it with.  Here, as in cl_deform_fwd16_kernel:
//   * a wave owns 16 rows and gathers 64 bytes of each per pass (lane = (row of 16, 16-byte piece of 4); fp32: two passes of 16 channels per tap,
//     bf16: one of 32): 32 registers of corner pieces in flight instead of 128, v_mfma_f32_16x16x4_f32 (same FLOP rate), 168 registers, three
//     waves per SIMD.  The register budget is the design constraint: a spilled value is reloaded through scratch, and every scratch reload waits
//     with vmcnt(0) — i.e. for ALL corner loads in flight (the first version of this kernel, with 50 - 140 spilled registers: 290 us);
//   * four taps are described at once, lane = (row, sub-tap) — all 64 lanes form one description each;
//   * the transposition between the MFMA layout (lane = voxel, consecutive channels) and the gather layout moves Col — one 16-byte LDS write and
//     one read per lane and pass — instead of the three derivative tiles (12 + 12 per 32 rows above); the dots with Col happen in the gather
//     layout, channel by channel, and the channel sum of a row is two DPP adds over its 4 lanes.
// Same arithmetic per (voxel, tap, channel) as the kernel above (same separable interpolation, same sample stored for the weight gradient); the
// order of the channel sum differs (rounding only).
// ---------------------------------------------------------------------------------------------------------------------
// B16 (T = bf16_t, p.wp16 set): Col on v_mfma_f32_16x16x32_bf16 — ONE instruction contracts all 32 grad_out channels of a 16 x 16 tile.  A = the weights'
// two-term bf16 records (prep mode 2 | 8; the records of ci = 2 i and 2 i + 1 are adjacent: one 32-byte read per term), read by every lane from L2 — no LDS
// weight tile, no workgroup barrier per tap; B = the 16 bytes of the voxel's bf16 grad_out row that hold channels 8 g4 .. 8 g4 + 7, as loaded.  4 MFMAs per tap
// instead of 16; the D layout is that of v_mfma_f32_16x16x4_f32, so everything behind the contraction is unchanged.
// (The two-term split of fp32 rows that cl_deform_goff2_kernel / cl_deform_gx_fx2_kernel use does NOT pay here: 148 against 139 us at stage 0 — two passes of
// three MFMAs plus four record loads per tap push the kernel over its 168-register budget, and a spilled register costs a vmcnt(0) per reload.  fp32 keeps
// v_mfma_f32_16x16x4_f32.)
template <typename T, bool SAMP, int WAVES, int OCC, bool B16 = false>   // WAVES per workgroup (4 | 8), OCC = waves per SIMD the register budget is set for
__global__ __launch_bounds__(64 * WAVES, OCC) void cl_deform_goff16_kernel(DeformBwdArgs p)
{
    static_assert(!B16 || sizeof(T) == 2, "B16 needs bf16 activations");
    constexpr unsigned XB = sizeof(T);
    constexpr int TGRP = 4;
    constexpr int PE = 16 / XB;            // channels per 16-byte piece: 4 (fp32) | 8 (bf16)
    constexpr int CPP = 4 * PE;            // channels per pass (4 pieces = 64 bytes of a row): 16 | 32
    constexpr int NPASS = 32 / CPP;        // 2 | 1
    constexpr int TP = CPP / 16;           // 16-row MFMA tiles per pass: 1 | 2
    constexpr int SROW = CPP + 4;          // padded Col tile row (floats)
    const T *gin = reinterpret_cast<const T *>(p.g);
    __shared__ __attribute__((aligned(16))) float Bs2[B16 ? 1 : 2][B16 ? 4 : 32 * 32];      // W[tap][co][ci], double buffered (fp32-input MFMA only)
    __shared__ __attribute__((aligned(16))) float Csm[WAVES][16 * SROW];                    // per wave: Col[row][channel of the pass]
    __shared__ __attribute__((aligned(16))) float Dsm[WAVES][TGRP * 16 * GATHER_DESC_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g4 = lane >> 4;   // MFMA roles: voxel (column) / W row, k group; description role: row, sub-tap
    const int gr = lane >> 2, gp = lane & 3;   // gather roles: row, 16-byte piece of the pass's 64 bytes
    const int bx = DLKA_XCD_BX(p.xcd_nx);
    if (bx < 0) return;
    const int mbase = (bx * WAVES + wave) * 16;
    const int HW = p.H * p.W, rowbytes = p.C * XB;
    const BufRsrc rin = make_rsrc(p.in, (size_t)p.M * p.C * XB);
    const BufRsrc rsamp = make_rsrc(p.samp, SAMP ? (size_t)p.K * p.M * p.C * XB : 0);
    const BufRsrc rgoff = make_rsrc(p.goff, (size_t)p.B * 3 * p.K * p.N * 4);
    float *Ct = Csm[wave], *Dt = Dsm[wave];

    // B operand of all taps' MFMAs: grad_out[voxel i][co = 8 g4 + s]   (B16: the same eight channels as the four raw words of the bf16 row)
    float greg[8];
    bf16x8 graw;
    {
        const bool okr = mbase + i < p.M;
        const long gi = okr ? (long)(mbase + i) * p.Cout + 8 * g4 : 0;
        if (B16) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const bf16_t *>(p.g) + gi);
            const float w4[4] = {okr ? t[0] : 0.f, okr ? t[1] : 0.f, okr ? t[2] : 0.f, okr ? t[3] : 0.f};
            graw = bf16x8_from_words(w4);
        } else {
            const f32x4 t0 = act_load4(gin, gi), t1 = act_load4(gin, gi + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { greg[e] = okr ? t0[e] : 0.f; greg[4 + e] = okr ? t1[e] : 0.f; }
        }
    }
    // where the row this lane reduces (gather role, piece 0) goes: grad_offset is planar, [b][3 K][N]; byte offset of plane 0, DLKA_OOB = no store
    unsigned gbyte = DLKA_OOB;
    {
        const int mr = mbase + gr;
        if (mr < p.M && gp == 0) { const int br = mr / p.N; gbyte = (unsigned)(br * 3 * p.K * p.N + (mr - br * p.N)) * 4u; }
    }
    const unsigned samp_v0 = (mbase + gr < p.M) ? (unsigned)((mbase + gr) * p.C + PE * gp) * XB : DLKA_OOB;   // S[tap][m][c]: + (tap M C + channel) bytes

    GatherPiece<T> xr[8];   // corner pieces of the unit (tap, pass) in flight
    float onx[3] = {0.f, 0.f, 0.f};
    auto load_offsets = [&](int grp) {
        const int tap = TGRP * grp + g4, m = mbase + i;
        if (m < p.M && tap < p.K) {
            const int b = m / p.N;
            const float *op = p.off + ((long)b * 3 * p.K + 3 * tap) * p.N + (m - b * p.N);
            onx[0] = op[0]; onx[1] = op[p.N]; onx[2] = op[2 * (long)p.N];
        }
    };
    int cur_grp = -1;
    auto issue = [&](int tap, int pass) {
        const int grp = tap / TGRP;
        if (grp != cur_grp) {   // uniform: describe the four taps of the group, lane = (row i, sub-tap g4)
            cur_grp = grp;
            const int mytap = TGRP * grp + g4, m = mbase + i;
            int ti, tj, tk;
            tap_decode(mytap < p.K ? mytap : 0, p.kw, p.kh, ti, tj, tk);
            wave_sync();        // every lane has consumed the previous table
            RowDesc r;
            r.base = 0; r.okm = 0; r.ld = r.lh = r.lw = 0.f;
            r.zd = r.zh = r.zw = 0;
            if (m < p.M && mytap < p.K) {
                // (row coordinates re-derived here, every fourth tap, rather than held in registers across the loop)
                const int b = m / p.N, v = m - b * p.N;
                const int w0 = v % p.W, h0 = (v / p.W) % p.H, d0 = v / HW;
                r = gather_describe3(onx[0], onx[1], onx[2], p.N, b, d0 + ti * p.dd - p.pd, h0 + tj * p.dh - p.ph, w0 + tk * p.dw - p.pw, p.D, p.H, p.W);
            }
            gather_publish(Dt, g4 * 16 + i, r, rowbytes);
            load_offsets(grp + 1);
            wave_sync();
        }
        const RowLook r = gather_lookup(Dt + (tap - TGRP * grp) * 16 * GATHER_DESC_WORDS, gr);
        const unsigned cbyte = (unsigned)(CPP * pass + PE * gp) * XB;
#pragma unroll
        for (int q = 0; q < 8; ++q) xr[q] = gather_load<T>(rin, gather_offset(r, q, HW, p.W, rowbytes, cbyte));
    };
    f32x4 wreg = {0.f, 0.f, 0.f, 0.f};
    f32x4 wrec[B16 ? 4 : 1];   // B16: [part * 2 + t] = the record of (term part, ci = 2 i + t, k-group g4) of the tap in flight
    const bool wload = tid < 256;   // a tap's 32 x 32 weight tile: one 16-byte piece per thread of the first four waves
    auto load_w = [&](int tap) {
        if (B16) {   // unit (tap, co chunk 0) = [part][mf][h][C = 32][8 bf16], co = 16 h + 8 mf + e = 8 g4 + e  <=>  h = g4 >> 1, mf = g4 & 1
            const float *src = p.wp16 + (long)tap * p.CoutP * p.C;
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const f32x4 *rec = reinterpret_cast<const f32x4 *>(src + ((long)((part * 2 + (g4 & 1)) * 2 + (g4 >> 1)) * p.C + 2 * i) * 4);   // tiles t = 0, 1: ci = 2 i + t
                wrec[part * 2] = rec[0];
                wrec[part * 2 + 1] = rec[1];
            }
            return;
        }
        if (wload) wreg = *(reinterpret_cast<const f32x4 *>(p.wp + ((long)tap * p.CoutP + (tid >> 3)) * p.C) + (tid & 7));
    };

    if (p.K > 0) {
        load_w(0);
        load_offsets(0);
        issue(0, 0);
    }
#pragma unroll 1
    for (int tap = 0; tap < p.K; ++tap) {
        float *Bs = Bs2[B16 ? 0 : (tap & 1)];
        bf16x8 wa[B16 ? 4 : 1];
        if (B16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float w4[4] = {wrec[q][0], wrec[q][1], wrec[q][2], wrec[q][3]}; wa[q] = bf16x8_from_words(w4); }
        } else {
            if (wload) reinterpret_cast<f32x4 *>(Bs)[tid] = wreg;
            __syncthreads();   // this tap's tile staged; the other buffer (read one tap ago) is free for the next
        }
        if (tap + 1 < p.K) load_w(tap + 1);
        const RowLook rdg = gather_lookup(Dt + (tap & (TGRP - 1)) * 16 * GATHER_DESC_WORDS, gr);
        float gd = 0.f, gh = 0.f, gw = 0.f;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            // ---- 1. Col[voxel][ci] = sum_co grad_out[voxel][co] W[tap][co][ci] for the pass's channels: D rows = ci, columns = voxels ----
            f32x4 acc[TP];
#pragma unroll
            for (int t = 0; t < TP; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (B16) {   // tile t <-> ci = 2 i + t (as below); hi and lo term of the weights
#pragma unroll
                for (int t = 0; t < TP; ++t) {
                    acc[t] = mfma_16x16x32_bf16(wa[t], graw, acc[t]);
                    acc[t] = mfma_16x16x32_bf16(wa[2 + t], graw, acc[t]);
                }
            } else {
                // A[row][k = co = 8 g4 + s].  One tile per pass (fp32): row i <-> ci = 16 pass + i, the lane ends up with channels 4 g4 .. 4 g4 + 3 of
                // the pass.  Two tiles (bf16): one 8-byte read feeds both, tile t <-> ci = 2 i + t, the lane ends up with channels 8 g4 .. 8 g4 + 7.
                const float *arow = Bs + (8 * g4) * 32 + (TP == 2 ? 2 * i : CPP * pass + i);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
#pragma unroll
                    for (int t = 0; t < TP; ++t) acc[t] = mfma_16x16x4(arow[s * 32 + t], greg[s], acc[t]);
                }
            }
            wave_sync();   // the previous pass's Col tile has been read by every lane
            if (TP == 1) *reinterpret_cast<f32x4 *>(Ct + i * SROW + 4 * g4) = acc[0];
            else {
                f32x4 c0, c1;
                c0[0] = acc[0][0]; c0[1] = acc[TP - 1][0]; c0[2] = acc[0][1]; c0[3] = acc[TP - 1][1];
                c1[0] = acc[0][2]; c1[1] = acc[TP - 1][2]; c1[2] = acc[0][3]; c1[3] = acc[TP - 1][3];
                f32x4 *dst = reinterpret_cast<f32x4 *>(Ct + i * SROW + 8 * g4);
                dst[0] = c0; dst[1] = c1;
            }
            wave_sync();
            // ---- 2. in the gather layout: derivative samples of this lane's 16-byte piece, channel by channel, dotted with Col on the spot ----
            f32x4 sraw = {0.f, 0.f, 0.f, 0.f};   // the piece's samples as they are stored: 4 floats, or 8 bf16 in 4 words (packed per 4 channels: 8 floats held spill)
#pragma unroll
            for (int vv = 0; vv < PE / 4; ++vv) {
                const f32x4 col = *reinterpret_cast<const f32x4 *>(Ct + gr * SROW + PE * gp + 4 * vv);
                float sv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // separable (see the kernel above, cuh:111-190): along w, then h, then d
                    const int k = 4 * vv + e;
                    const float x0 = xr[0].elem(k), x1 = xr[1].elem(k), x2 = xr[2].elem(k), x3 = xr[3].elem(k);
                    const float x4 = xr[4].elem(k), x5 = xr[5].elem(k), x6 = xr[6].elem(k), x7 = xr[7].elem(k);
                    const float dq0 = x1 - x0, dq1 = x3 - x2, dq2 = x5 - x4, dq3 = x7 - x6;
                    const float sw0 = fmaf(rdg.lw, dq0, x0), sw1 = fmaf(rdg.lw, dq1, x2), sw2 = fmaf(rdg.lw, dq2, x4), sw3 = fmaf(rdg.lw, dq3, x6);
                    const float eh0 = sw1 - sw0, eh1 = sw3 - sw2;
                    const float shw0 = fmaf(rdg.lh, eh0, sw0), shw1 = fmaf(rdg.lh, eh1, sw2);
                    const float ew0 = fmaf(rdg.lh, dq1 - dq0, dq0), ew1 = fmaf(rdg.lh, dq3 - dq2, dq2);
                    const float dd = shw1 - shw0;
                    if (SAMP) sv[e] = fmaf(rdg.ld, dd, shw0);
                    const float dh = fmaf(rdg.ld, eh1 - eh0, eh0);
                    const float dw = fmaf(rdg.ld, ew1 - ew0, ew0);
                    gd = fmaf(col[e], dd, gd); gh = fmaf(col[e], dh, gh); gw = fmaf(col[e], dw, gw);
                }
                if (SAMP && XB == 4) { sraw[0] = sv[0]; sraw[1] = sv[1]; sraw[2] = sv[2]; sraw[3] = sv[3]; }
                if (SAMP && XB == 2) {
                    sraw[2 * vv] = __uint_as_float((unsigned)bf16_bits(sv[0]) | ((unsigned)bf16_bits(sv[1]) << 16));
                    sraw[2 * vv + 1] = __uint_as_float((unsigned)bf16_bits(sv[2]) | ((unsigned)bf16_bits(sv[3]) << 16));
                }
            }
            if (SAMP && XB == 4 && p.samp_f16)   // (uniform) the fp32 path's samples as halves: 8 bytes per lane and pass
                buf_store_f16x4(rsamp, samp_v0 == DLKA_OOB ? DLKA_OOB : (samp_v0 >> 1) + (unsigned)(tap * p.M * p.C + CPP * pass) * 2u, sraw);
            else if (SAMP) buf_store_f32x4(rsamp, samp_v0 == DLKA_OOB ? DLKA_OOB : samp_v0 + (unsigned)(tap * p.M * p.C + CPP * pass) * XB, sraw);
            // the next unit's corner loads go out now: in flight under the reduction below, the next staging and the next MFMAs
            sched_fence();   // (the scheduler would hoist these loads above the interpolation into a SECOND set of piece registers — and spill)
            if (pass + 1 < NPASS) issue(tap, pass + 1);
            else if (tap + 1 < p.K) issue(tap + 1, 0);
            sched_fence();
        }
        // ---- 3. channel sum of the row: over the 4 lanes that hold its pieces ----
        gd = sum4(gd); gh = sum4(gh); gw = sum4(gw);
        {
            const unsigned o = gbyte == DLKA_OOB ? DLKA_OOB : gbyte + (unsigned)(3 * tap * p.N) * 4u;
            buf_store_f32(rgoff, o, gd);
            buf_store_f32(rgoff, o == DLKA_OOB ? DLKA_OOB : o + (unsigned)p.N * 4u, gh);
            buf_store_f32(rgoff, o == DLKA_OOB ? DLKA_OOB : o + 2u * (unsigned)p.N * 4u, gw);
        }
    }
}

// =====================================================================================================================
// grad_input: brick scatter into an fp64 LDS window
// =====================================================================================================================
struct GxGeom {
    int bd, bh, bw;         // brick of output voxels
    int nbd, nbh, nbw;      // bricks per axis
    int wvox_max;           // (bd + 2 HALO)(bh + 2 HALO)(bw + 2 HALO), each factor clipped to the volume
    int nslices;            // C / CS
    int ngroups;            // ceil(K / TG)
    int resident;           // all tap groups' weight tiles fit in LDS next to the window
    int xcd_nx;             // > 0: blockIdx.x is mapped through xcd_item() (set by the launcher)
    int item_grid;          // 1: 1-D grid over (brick, slice) items; 0: x = brick, y = slice
    int tap_far;            // 1: half-waves take taps 16 apart (K <= 32 = 4 groups of 8), see gx_tap()
    float fx_magic;         // 1.5 * 2^23 (second-generation kernel: rounding constant, passed in a register on purpose)
    float fx_lim;           // fixed-point window: largest scaled magnitude of one contribution, R*K * (fx_lim + 1/2) < 2^31 (R = rows of a
                            // brick, K = taps), see the kernel
};

// MFMA row t8 of tap group grp <-> tap.  The two half-waves scatter rows t8 = 2*r4 and 2*r4 + 1 in the same instruction; with consecutive
// taps there (tk, tk + 1) lane (j, 1) and lane (j + 1, 0) aim at the SAME window cell whenever their offsets floor alike — the
// "consecutive + 0/1 jitter" pattern that costs 21 instead of 9.4 cycles per ds_add_f64 in isolation (profiles/archive/r01e_lds_f64_patterns.txt).
// tap_far pairs tap x with tap x + 16 instead (another d-plane of the window).  In the real kernel it did NOT pay (390 vs 370 us at 32^3):
// the LDS unit serves the half-waves in separate passes, and neighbouring taps share offset / window cache lines.  Kept as an A/B knob.
__device__ __host__ __forceinline__ int gx_tap(int grp, int t8, int tap_far)
{
    return tap_far ? (t8 & 1) * 16 + grp * 4 + (t8 >> 1) : grp * TG + t8;
}

__device__ __host__ __forceinline__ void gx_window(int b0, int bsz, int size, int &lo, int &len)
{
    lo = b0 - HALO < 0 ? 0 : b0 - HALO;
    int hi = b0 + bsz + HALO;
    if (hi > size) hi = size;
    len = hi - lo;
}

// FX: the window holds 64-bit INTEGER cells, two channels per cell as 32-bit fixed-point fields (value = hi * 2^32 + lo in signed
// arithmetic, so the sum of packed words is the packed word of the two sums as long as each stays below 2^31).  One ds_add_u64 then
// carries two channels: half the LDS atomic instructions of the fp64 window (the kernel's bound), integer adds are
// exact and order-independent, ds_add_u64 is the faster instruction (8.2 vs 6.8 lanes/clk/CU), and the 61 KB window fits TWICE per CU
// (the fp64 one is 122 KB).  The price is a per-(brick, slice) power-of-two scale; it is chosen from a Cauchy-Schwarz bound of |Col| so
// that overflow is impossible for ANY offsets (see the kernel): 2^31 / (R*K) - 1 = 155 343 levels (17.2 bits) for the largest possible contribution
// at the default brick (R*K = 512 * 27), i.e. an error <= 3.2e-6 of that bound per add; measured 1.9e-4 of max|grad_input| at 32^3.  Default where it is faster (C <= 64, resident weights).
template <bool FX, typename T = float>   // T: storage of grad_out `g` (channels-last); the windows, their scratch slabs and grad_input stay fp32
__global__ __launch_bounds__(512, FX ? 4 : 2) void cl_deform_gx_kernel(DeformBwdArgs p, GxGeom gg, float *__restrict__ scratch)
{
    const T *gin = reinterpret_cast<const T *>(p.g);
    DLKA_DYN_SMEM(unsigned char, smem0);
    unsigned *smax = reinterpret_cast<unsigned *>(smem0);   // 16 bytes of scalars first (static LDS next to a 160 KB dynamic limit is refused)
    unsigned char *smem = smem0 + 16;
    double *Win = reinterpret_cast<double *>(smem);                                        // [CS][wvox]
    unsigned long long *WinI = reinterpret_cast<unsigned long long *>(smem);               // FX: [CS / 2][wvox]
    float *Bs = reinterpret_cast<float *>(smem + (size_t)(gg.wvox_max + 64) * (FX ? CS / 2 : CS) * sizeof(double));   // [CoutP][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    // XCD-aware order: an XCD owns a contiguous range of bricks (they share halo rows of grad_out / offsets); optionally (item_grid) a 1-D grid
    // over (brick, slice) items with the slices of a brick adjacent
    int item;
    if (gg.item_grid) item = gg.xcd_nx > 0 ? xcd_item((int)blockIdx.x, gg.xcd_nx) : (int)blockIdx.x;
    else {   // brick-major 2-D grid (x = brick, XCD-swizzled; y = slice)
        const int bk = DLKA_XCD_BX(gg.xcd_nx);
        item = bk < 0 ? -1 : bk * gg.nslices + (int)blockIdx.y;
    }
    if (item < 0) return;
    const int brick = item / gg.nslices;
    int bid = brick;
    const int bw_i = bid % gg.nbw; bid /= gg.nbw;
    const int bh_i = bid % gg.nbh; bid /= gg.nbh;
    const int bd_i = bid % gg.nbd; const int b = bid / gg.nbd;
    const int slice = item - brick * gg.nslices;
    const int bd0 = bd_i * gg.bd, bh0 = bh_i * gg.bh, bw0 = bw_i * gg.bw;
    int wd0, wh0, ww0, WD, WH, WW;
    gx_window(bd0, gg.bd, p.D, wd0, WD);
    gx_window(bh0, gg.bh, p.H, wh0, WH);
    gx_window(bw0, gg.bw, p.W, ww0, WW);
    const int wvox = WD * WH * WW, WHW = WH * WW;
    const int wstride = wvox + 64;     // channel plane stride: the window cells, then one trash cell per lane
    const int trash = wvox + lane;
    const int R = gg.bd * gg.bh * gg.bw, ntiles = cdiv(R, 32);
    const int nkc = p.CoutP / 32;

    for (int e = tid; e < wstride * (FX ? CS / 2 : CS); e += blockDim.x) Win[e] = 0.0;   // (0.0 and integer 0 share the bit pattern)
    float fx_scale = 1.f, fx_inv = 1.f;
    // A operand tile of group grp: Bs[co][t8*4 + c4] = W[co][slice*4 + c4][tap = grp*8 + t8]   (wp[tap][co][ci], zero beyond K)
    auto stage_weights = [&](int grp, float *dstB) {
        for (int e = tid; e < p.CoutP * TG; e += blockDim.x) {
            const int co = e / TG, t8 = e - co * TG, tap = gx_tap(grp, t8, gg.tap_far);
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (tap < p.K) val = *reinterpret_cast<const f32x4 *>(p.wp + ((long)tap * p.CoutP + co) * p.C + slice * CS);
            reinterpret_cast<f32x4 *>(dstB)[e] = val;
        }
    };
    auto load_g = [&](int tile, int kc, float *gl) {
        const int row = tile * 32 + j;
        const int rw = row % gg.bw, rh = (row / gg.bw) % gg.bh, rd = row / (gg.bw * gg.bh);
        const int vd = bd0 + rd, vh = bh0 + rh, vw = bw0 + rw;
        const bool ok = row < R && vd < p.D && vh < p.H && vw < p.W && kc * 32 + 16 * h < p.Cout;
        const long m = (long)b * p.N + (ok ? (vd * p.H + vh) * p.W + vw : 0);
        const long gi = ok ? m * p.Cout + kc * 32 + 16 * h : 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 t = act_load4(gin, gi + 4 * e);
            gl[4 * e] = ok ? t[0] : 0.f; gl[4 * e + 1] = ok ? t[1] : 0.f; gl[4 * e + 2] = ok ? t[2] : 0.f; gl[4 * e + 3] = ok ? t[3] : 0.f;
        }
    };

    // one (32-voxel tile, 8-tap group): Col on the matrix cores, then the scatter
    auto tile_group = [&](int tile, int grp, const float *Bg, const float (*gl)[16], bool reload) {
        const int row = tile * 32 + j;
        const int rw = row % gg.bw, rh = (row / gg.bw) % gg.bh, rd = row / (gg.bw * gg.bh);
        const int vd = bd0 + rd, vh = bh0 + rh, vw = bw0 + rw;
        const bool ok = row < R && vd < p.D && vh < p.H && vw < p.W;
        const int v = ok ? (vd * p.H + vh) * p.W + vw : 0;
        // the offsets of this lane's four samples of the group, requested before the MFMA phase (see cl_deform_gx_fx2_kernel: loaded inside the
        // scatter step, each step began with an exposed L2 round trip)
        float offv[4][3];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int tq = gx_tap(grp, 2 * r4 + h, gg.tap_far);
            const float *op = p.off + ((long)b * 3 * p.K + 3 * (tq < p.K ? tq : 0)) * p.N + v;
            offv[r4][0] = op[0]; offv[r4][1] = op[p.N]; offv[r4][2] = op[2 * (long)p.N];
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (!reload) {
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {   // static indices keep gl in registers
                if (kc >= nkc) break;
                const float *arow = Bg + (kc * 32 + 16 * h) * 32 + j;   // A[i = (t8, c4) = j][k = co]
#pragma unroll
                for (int st = 0; st < 16; ++st) acc = mfma_32x32x2(arow[st * 32], gl[kc][st], acc);
            }
        } else {
            for (int kc = 0; kc < nkc; ++kc) {
                float g1[16];
                load_g(tile, kc, g1);
                const float *arow = Bg + (kc * 32 + 16 * h) * 32 + j;
#pragma unroll
                for (int st = 0; st < 16; ++st) acc = mfma_32x32x2(arow[st * 32], g1[st], acc);
            }
        }
        {
            // acc[r]: MFMA row (r&3) + 8*(r>>2) + 4h  ->  c4 = r & 3, t8 = 2*(r>>2) + h;   column = voxel j
            // The window part of the scatter is branch-free: a corner outside the window keeps its 4 atomics but they add 0.0 to
            // this lane's private trash cell (index wvox + lane), so the 32 ds_add_f64 of a tap issue back to back with no exec-mask
            // juggling per corner.  Corners inside the volume but outside the window (rare: |offset| > HALO) take global atomics.
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int tap = gx_tap(grp, 2 * r4 + h, gg.tap_far);
            if (!ok || tap >= p.K) continue;
            int ti, tj, tk;
            if (p.kw == 3 && p.kh == 3) { ti = tap / 9; const int rr = tap - 9 * ti; tj = rr / 3; tk = rr - 3 * tj; }   // uniform
            else tap_decode(tap, p.kw, p.kh, ti, tj, tk);
            LaneTap s;
            lane_tap(s, offv[r4][0], offv[r4][1], offv[r4][2], vd + ti * p.dd - p.pd, vh + tj * p.dh - p.ph, vw + tk * p.dw - p.pw, p.D, p.H, p.W);
            if (!s.okm) continue;
            const float fd[2] = {1.f - s.ld, s.ld}, fh[2] = {1.f - s.lh, s.lh}, fw[2] = {1.f - s.lw, s.lw};
            const int xd = s.zd - wd0, xh = s.zh - wh0, xw = s.zw - ww0;
            // the window lies inside the volume: in-window implies in-volume
            const bool ad[2] = {(unsigned)xd < (unsigned)WD, (unsigned)(xd + 1) < (unsigned)WD};
            const bool ah[2] = {(unsigned)xh < (unsigned)WH, (unsigned)(xh + 1) < (unsigned)WH};
            const bool aw[2] = {(unsigned)xw < (unsigned)WW, (unsigned)(xw + 1) < (unsigned)WW};
            const int base = (xd * WH + xh) * WW + xw;
            const float wdh[4] = {fd[0] * fh[0], fd[0] * fh[1], fd[1] * fh[0], fd[1] * fh[1]};
            unsigned glb = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int cd = (q >> 2) & 1, ch = (q >> 1) & 1, cw = q & 1;
                const bool inwin = ad[cd] & ah[ch] & aw[cw];
                const float wq = wdh[2 * cd + ch] * fw[cw];
                glb |= ((!inwin) & ((s.okm >> q) & 1u)) << q;
                const int idx = inwin ? base + cd * WHW + ch * WW + cw : trash;
                const float wv = inwin ? wq : 0.f;
                double *cell = Win + idx;
                if (FX) {
                    const float ws = wv * fx_scale;
#pragma unroll
                    for (int pr = 0; pr < CS / 2; ++pr) {
                        const int i0 = rint_i32(acc[4 * r4 + 2 * pr] * ws), i1 = rint_i32(acc[4 * r4 + 2 * pr + 1] * ws);
                        // packed = (int64)i1 * 2^32 + (int64)i0: low word i0, high word i1 - 1 if i0 < 0
                        const unsigned long long pk = ((unsigned long long)(unsigned)(i1 + (i0 >> 31)) << 32) | (unsigned long long)(unsigned)i0;
                        atomicAdd(WinI + idx + pr * wstride, pk);
                    }
                    continue;
                }
#pragma unroll
                for (int c = 0; c < CS; ++c) atomicAdd(cell + c * wstride, (double)(acc[4 * r4 + c] * wv));
            }
            if (glb) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (!((glb >> q) & 1u)) continue;
                    const int cd = (q >> 2) & 1, ch = (q >> 1) & 1, cw = q & 1;
                    const float wq = wdh[2 * cd + ch] * fw[cw];
                    float *dst = p.gx + ((long)b * p.N + (long)((s.zd + cd) * p.H + s.zh + ch) * p.W + s.zw + cw) * p.C + slice * CS;
#pragma unroll
                    for (int c = 0; c < CS; ++c) atomicAdd(dst + c, acc[4 * r4 + c] * wq);
                }
            }
        }
        }
    };
    if (gg.resident) {
        // all tap groups' weights stay in LDS: a tile's grad_out rows are loaded ONCE (registers) for its 4 groups and the
        // main loop has no workgroup barrier.  (Measured before: grad_out re-read per group and slice = 377 MB of L2->HBM
        // traffic against 8 MB of data, profiles/archive/r01j_pmc.)
        for (int grp = 0; grp < gg.ngroups; ++grp) stage_weights(grp, Bs + (size_t)grp * p.CoutP * 32);
        if (FX && tid < 2) smax[tid] = 0u;
        __syncthreads();   // also: window zeroed
        if (FX) {
            // Power-of-two scale with a PROVABLE overflow bound.  Every contribution is Col * wq with 0 <= wq <= 1 and
            //   |Col(c, tap, o)| = |sum_co W[co][c][tap] G[o][co]| <= ||W[:, c, tap]||_2 * ||G[o, :]||_2      (Cauchy-Schwarz)
            //                    <= vmax := max over this slice's (c, tap) columns  x  max over this brick's rows,
            // and one cell receives at most ONE corner from each (row, tap) pair of the brick, i.e. <= R*K contributions.  With
            // |rint(Col * wq * scale)| <= vmax * scale + 1/2 <= fx_lim + 1/2 and the host's fx_lim (R*K * (fx_lim + 1/2) < 2^31), no 32-bit
            // field can overflow whatever the offsets are.  (The column norms come from the weight tiles just staged in LDS.)
            float wm = 0.f;
            if (tid < gg.ngroups * 32) {
                const float *col = Bs + (size_t)(tid >> 5) * p.CoutP * 32 + (tid & 31);
                float ss = 0.f;
                for (int co = 0; co < p.CoutP; ++co) ss = fmaf(col[co * 32], col[co * 32], ss);
                wm = sqrtf(ss);
            }
            float gm = 0.f;
            for (int row = tid; row < R; row += blockDim.x) {
                const int rw = row % gg.bw, rh = (row / gg.bw) % gg.bh, rd = row / (gg.bw * gg.bh);
                const int vd = bd0 + rd, vh = bh0 + rh, vw = bw0 + rw;
                if (vd < p.D && vh < p.H && vw < p.W) {
                    const long gi = ((long)b * p.N + (vd * p.H + vh) * p.W + vw) * p.Cout;
                    float ss = 0.f;
                    for (int co = 0; co < p.Cout; co += 4) {
                        const f32x4 g4 = act_load4(gin, gi + co);
                        ss = fmaf(g4[0], g4[0], fmaf(g4[1], g4[1], fmaf(g4[2], g4[2], fmaf(g4[3], g4[3], ss))));
                    }
                    gm = fmaxf(gm, sqrtf(ss));
                }
            }
            atomicMax(&smax[0], __float_as_uint(wm));   // non-negative floats order like their bit patterns
            atomicMax(&smax[1], __float_as_uint(gm));
            __syncthreads();
            // 1.0001: the fp32 rounding of the two norms and of the MFMA's own sum (each < 1e-5 relative)
            const float vmax = __uint_as_float(smax[0]) * __uint_as_float(smax[1]) * 1.0001f;
            // |Col * wq * scale| <= vmax * scale <= fx_lim,  R*K * (fx_lim + 1/2) < 2^31  (fx_lim from the host); the scale need not be a
            // power of two: the window holds integers, only the final read-back multiplies by 1/scale
            if (vmax > 1e-30f && vmax < 3.0e38f) {
                fx_scale = gg.fx_lim / vmax * 0.999999f;
                fx_inv = 1.f / fx_scale;
            }
        }
        if (ntiles < nwaves || nkc > 4) {
            // tiny volumes (fewer tiles than waves: at 4^3 / 8^3 a brick is 2 tiles and 6 of the 8 waves had nothing to do), or more grad_out
            // chunks than the register copy holds (Cout = 256): the waves share (tile, tap group) ITEMS and re-read the grad_out rows per item
            const int nitems = ntiles * gg.ngroups;
            for (int item = wave; item < nitems; item += nwaves) {
                const int tile = item / gg.ngroups, grp = item - tile * gg.ngroups;
                tile_group(tile, grp, Bs + (size_t)grp * p.CoutP * 32, nullptr, true);
            }
        } else
        for (int tile = wave; tile < ntiles; tile += nwaves) {
            float gl[4][16];
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                if (kc < nkc) load_g(tile, kc, gl[kc]);
            for (int grp = 0; grp < gg.ngroups; ++grp) tile_group(tile, grp, Bs + (size_t)grp * p.CoutP * 32, gl, false);
        }
    } else {
        for (int grp = 0; grp < gg.ngroups; ++grp) {
            __syncthreads();   // Bs consumed by every wave (first pass: window zeroed)
            stage_weights(grp, Bs);
            __syncthreads();
            for (int tile = wave; tile < ntiles; tile += nwaves) {
                tile_group(tile, grp, Bs, nullptr, true);
            }
        }
    }
    __syncthreads();
    // flush: scratch[brick][slice][cell] as float4 (the four channels of the slice)
    f32x4 *dst = reinterpret_cast<f32x4 *>(scratch) + ((long)brick * gg.nslices + slice) * gg.wvox_max;
    for (int e = tid; e < wvox; e += blockDim.x) {
        f32x4 o;
        if (FX) {
#pragma unroll
            for (int pr = 0; pr < CS / 2; ++pr) {
                const long long sv = (long long)WinI[pr * wstride + e];
                const int lo = (int)(unsigned)(sv & 0xffffffffll);          // sum of the low fields (sign carried by the two's complement)
                const long long hi = (sv - (long long)lo) >> 32;            // sum of the high fields
                o[2 * pr] = (float)lo * fx_inv;
                o[2 * pr + 1] = (float)hi * fx_inv;
            }
        } else {
        o[0] = (float)Win[e]; o[1] = (float)Win[wstride + e]; o[2] = (float)Win[2 * wstride + e]; o[3] = (float)Win[3 * wstride + e];
        }
        dst[e] = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fixed-point window, second generation.  The first one is VALU-bound, not LDS-bound: ~600 VALU instructions per (32 voxels x 2 taps) scatter
// step — per corner an in-window test, a trash-cell select, a weight select, the bookkeeping of the global-atomic fallback and runtime index
// arithmetic — against 16 LDS atomics (ISA count).  Here
//   * the window has COMPILE-TIME strides (SW cells per w-row, SH rows per plane, SD planes per channel pair): a sample's 16 atomics share
//     ONE address register, corner and channel-pair offsets ride in the instruction's immediate offset field;
//   * where the window is clipped by a volume face it gets one GUARD cell beyond the face: a corner outside the volume lands there (never
//     flushed) instead of being tested for — the guard q > -1 && q < size keeps every corner within one cell of the volume;
//   * one test per SAMPLE ("all eight corners inside window + guard") replaces the eight per-corner tests; the rare sample that fails it
//     (|offset| beyond the halo) sends its in-volume corners to global atomics, exactly as before.
// Same MFMA phase and scale (Cauchy-Schwarz bound, see cl_deform_gx_kernel); every contribution is rounded to nearest-even once, from the exact
// product (the first generation rounds the fp32 product): the window sums agree with the first generation's to a few quanta.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef DLKA_GX_QW
#define DLKA_GX_QW 24
#endif
constexpr int GX_QW = DLKA_GX_QW;   // far-sample records per wave (8 floats each)

// Drains a wave's far-sample queue: lane = (record r0 + (lane >> 5), corner (lane >> 2) & 7, channel lane & 3): every in-volume corner of a
// queued sample adds Col * weight to grad_input with a global fp32 atomic (grad_input is zero-initialised by the caller).
__device__ __forceinline__ void gx_drain_far(const DeformBwdArgs &p, const float *Qw, int count, int b, int slice, int lane)
{
    wave_sync();   // the records written by other lanes are visible
    const int q = (lane >> 2) & 7, c = lane & 3;
    const int cd = (q >> 2) & 1, ch = (q >> 1) & 1, cw = q & 1;
    for (int r0 = 0; r0 < count; r0 += 2) {
        const int r = r0 + (lane >> 5);
        if (r < count) {
            const float *rec = Qw + r * 8;
            const int z = __float_as_int(rec[0]);
            const int cz = (z >> 20) - 1 + cd, cy = ((z >> 10) & 1023) - 1 + ch, cx = (z & 1023) - 1 + cw;
            if ((unsigned)cz < (unsigned)p.D && (unsigned)cy < (unsigned)p.H && (unsigned)cx < (unsigned)p.W) {
                const float ld = rec[1], lh = rec[2], lw = rec[3];
                // same product order as the regular path: (fd * fh) * fw
                const float wq = ((cd ? ld : 1.f - ld) * (ch ? lh : 1.f - lh)) * (cw ? lw : 1.f - lw);
                atomicAdd(p.gx + ((long)b * p.N + (long)(cz * p.H + cy) * p.W + cx) * p.C + slice * CS + c, rec[4 + c] * wq);
            }
        }
    }
    wave_sync();   // the queue may be refilled
}

// B16 (T = bf16_t): Col on v_mfma_f32_32x32x16_bf16.  The weight tiles of the slice are staged in LDS as bf16 — two terms (hi, lo: w to 2^-17), TRANSPOSED:
// Bh[grp][term][row (t8, c4)][co], rows padded by 16 bytes — so that the A operand of a k-group is ONE 16-byte LDS read (channels 16 h + 8 mf .. + 7 of
// the lane's row); the B operand is the 16 bytes of the voxel's bf16 grad_out row with those channels, as loaded (no conversion, half the row registers).
// 4 MFMAs per (tile, tap group, 32-channel chunk) instead of 16; same D layout, the scatter behind it is unchanged.  gx3_b16_wbytes() = the LDS bytes of the tiles.
__host__ __device__ __forceinline__ int gx3_b16_rowh(int CoutP) { return CoutP + 8; }   // halfwords per padded row
__host__ __device__ __forceinline__ size_t gx3_b16_wbytes(int ngroups, int CoutP) { return (size_t)ngroups * 2 * 32 * gx3_b16_rowh(CoutP) * 2; }

// B16 with T = float: fp32 grad_out rows, split in two bf16 terms once per tile (W_hi G_hi + W_lo G_hi + W_hi G_lo, see cl_deform_goff2_kernel).
template <int SW, int SH, int SD, typename T = float, int NKC = 0, bool B16 = false>   // NKC: 32-channel chunks of a grad_out row known at compile time (1, 2), 0 = up to 4
__global__ __launch_bounds__(512, 4) void cl_deform_gx_fx2_kernel(DeformBwdArgs p, GxGeom gg, float *__restrict__ scratch)
{
    constexpr bool SPL = B16 && sizeof(T) == 4;
    constexpr int PS = SD * SH * SW;   // cells per channel-pair plane
    const T *gin = reinterpret_cast<const T *>(p.g);
    DLKA_DYN_SMEM(unsigned char, smem0);
    unsigned *smax = reinterpret_cast<unsigned *>(smem0);
    unsigned long long *WinI = reinterpret_cast<unsigned long long *>(smem0 + 16);                  // [CS / 2][SD][SH][SW]
    float *Bs = reinterpret_cast<float *>(smem0 + 16 + (size_t)PS * (CS / 2) * sizeof(double));     // [ngroups][CoutP][32]   (B16: Bh, see above)
    unsigned short *Bh = reinterpret_cast<unsigned short *>(Bs);
    const int RH = gx3_b16_rowh(p.CoutP);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    // per-wave queue of the rare samples that leave window + guard ("far"): record = {packed floor cell, ld, lh, lw, Col[4]}.  Handling them on
    // the spot — a divergent branch with 32 global atomics behind it — cost more than the whole regular scatter (311 vs 179 us at 32^3 with
    // ~1-voxel offsets: some lane of most waves is far); queued, they are drained with all 64 lanes busy (lane = (record, corner, channel)).
    float *Qw = (B16 ? reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(Bs) + gx3_b16_wbytes(gg.ngroups, p.CoutP)) : Bs + (size_t)gg.ngroups * p.CoutP * 32) +
                (size_t)wave * GX_QW * 8;
    int qcount = 0;   // wave-uniform
    const int bk = DLKA_XCD_BX(gg.xcd_nx);
    if (bk < 0) return;
    const int brick = bk, slice = (int)blockIdx.y;
    int bid = brick;
    const int bw_i = bid % gg.nbw; bid /= gg.nbw;
    const int bh_i = bid % gg.nbh; bid /= gg.nbh;
    const int bd_i = bid % gg.nbd; const int b = bid / gg.nbd;
    const int bd0 = bd_i * gg.bd, bh0 = bh_i * gg.bh, bw0 = bw_i * gg.bw;
    int wd0, wh0, ww0, WD, WH, WW;
    gx_window(bd0, gg.bd, p.D, wd0, WD);
    gx_window(bh0, gg.bh, p.H, wh0, WH);
    gx_window(bw0, gg.bw, p.W, ww0, WW);
    // guard cells beyond the volume faces the window touches; (od, oh, ow) = volume coordinates of cell (0, 0, 0)
    const int gld = wd0 == 0, glh = wh0 == 0, glw = ww0 == 0;
    const int ND = WD + gld + (wd0 + WD == p.D), NH = WH + glh + (wh0 + WH == p.H), NW = WW + glw + (ww0 + WW == p.W);
    const int od = wd0 - gld, oh = wh0 - glh, ow = ww0 - glw;
    const int wvox = WD * WH * WW, WHW = WH * WW;
    const int R = gg.bd * gg.bh * gg.bw, ntiles = cdiv(R, 32);
    constexpr int KCMAX = NKC ? NKC : 4;   // (register arrays are sized by it: with 4 the unused chunks' registers spill)
    const int nkc = NKC ? NKC : p.CoutP / 32;

    for (int e = tid; e < PS * (CS / 2); e += blockDim.x) WinI[e] = 0ull;
    for (int grp = 0; grp < gg.ngroups; ++grp) {   // A operand tiles: Bs[grp][co][t8*4 + c4] = W[co][slice*4 + c4][tap = grp*8 + t8]
        float *dstB = Bs + (size_t)grp * p.CoutP * 32;
        for (int e = tid; e < p.CoutP * TG; e += blockDim.x) {
            const int co = e / TG, t8 = e - co * TG, tap = grp * TG + t8;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (tap < p.K) val = *reinterpret_cast<const f32x4 *>(p.wp + ((long)tap * p.CoutP + co) * p.C + slice * CS);
            if (B16) {   // Bh[grp][term][t8*4 + c4][co]: hi and lo term of each weight
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const unsigned short hi = bf16_bits(val[c4]), lo = bf16_bits(val[c4] - bf16_value(hi));
                    Bh[((size_t)(grp * 2 + 0) * 32 + t8 * 4 + c4) * RH + co] = hi;
                    Bh[((size_t)(grp * 2 + 1) * 32 + t8 * 4 + c4) * RH + co] = lo;
                }
            } else {
                reinterpret_cast<f32x4 *>(dstB)[e] = val;
            }
        }
    }
    if (tid < 2) smax[tid] = 0u;
    __syncthreads();
    // scale with the provable overflow bound of the first generation (see there)
    float fx_scale = 1.f, fx_inv = 1.f;
    const float fx_magic = gg.fx_magic;   // 1.5 * 2^23
    {
        float wm = 0.f;
        if (tid < gg.ngroups * 32) {
            float ss = 0.f;
            if (B16) {   // (of the weights the MFMAs will see: hi + lo)
                const unsigned short *r0 = Bh + ((size_t)((tid >> 5) * 2 + 0) * 32 + (tid & 31)) * RH, *r1 = r0 + (size_t)32 * RH;
                for (int co = 0; co < p.CoutP; ++co) { const float w = bf16_value(r0[co]) + bf16_value(r1[co]); ss = fmaf(w, w, ss); }
            } else {
                const float *col = Bs + (size_t)(tid >> 5) * p.CoutP * 32 + (tid & 31);
                for (int co = 0; co < p.CoutP; ++co) ss = fmaf(col[co * 32], col[co * 32], ss);
            }
            wm = sqrtf(ss);
        }
        // largest grad_out row norm of the brick.  One row per THREAD — eight dependent 16-byte loads from 64 different rows per wave instruction —
        // cost 15 us of a workgroup's 79 (s_memtime stamps, round 3); here a wave instruction covers whole rows: lane = (row, 16-byte piece), the
        // squared pieces are summed across the row's lanes, all loads of a wave in flight at once.
        float gm = 0.f;
        const int P = p.Cout >> 2;   // 4-element pieces per row
        if (P == 8 || P == 16) {
            const int rpw = 64 / P, pl = lane & (P - 1), rl = lane / P;
            float g2 = 0.f;
            // eight row groups per trip (a 512-row brick of 128-byte rows: one trip): their loads are issued back to back — a plain loop waited for
            // each load in turn
            for (int row0 = wave * rpw; row0 < R; row0 += 8 * nwaves * rpw) {
                f32x4 g4[8];
                bool okr[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = row0 + u * nwaves * rpw + rl;
                    int rw, rh;
                    const int t = divmod_fast(row, gg.bw, rw);
                    const int rd = divmod_fast(t, gg.bh, rh);
                    const int vd = bd0 + rd, vh = bh0 + rh, vw = bw0 + rw;
                    okr[u] = row < R && vd < p.D && vh < p.H && vw < p.W;
                    const long gi = okr[u] ? ((long)b * p.N + (vd * p.H + vh) * p.W + vw) * p.Cout + 4 * pl : 0;
                    g4[u] = act_load4(gin, gi);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    float ss = fmaf(g4[u][0], g4[u][0], fmaf(g4[u][1], g4[u][1], fmaf(g4[u][2], g4[u][2], g4[u][3] * g4[u][3])));
                    ss = sum8(okr[u] ? ss : 0.f);
                    if (P == 16) ss += __shfl_xor(ss, 8);
                    g2 = fmaxf(g2, ss);
                }
            }
            gm = sqrtf(g2);
        } else {
            for (int row = tid; row < R; row += blockDim.x) {
                const int rw = row % gg.bw, rh = (row / gg.bw) % gg.bh, rd = row / (gg.bw * gg.bh);
                const int vd = bd0 + rd, vh = bh0 + rh, vw = bw0 + rw;
                if (vd < p.D && vh < p.H && vw < p.W) {
                    const long gi = ((long)b * p.N + (vd * p.H + vh) * p.W + vw) * p.Cout;
                    float ss = 0.f;
                    for (int co = 0; co < p.Cout; co += 4) {
                        const f32x4 g4 = act_load4(gin, gi + co);
                        ss = fmaf(g4[0], g4[0], fmaf(g4[1], g4[1], fmaf(g4[2], g4[2], fmaf(g4[3], g4[3], ss))));
                    }
                    gm = fmaxf(gm, sqrtf(ss));
                }
            }
        }
        // one LDS atomic per wave, not per lane (512 same-address atomics serialise)
        for (int o = 32; o >= 1; o >>= 1) { wm = fmaxf(wm, __shfl_xor(wm, o)); gm = fmaxf(gm, __shfl_xor(gm, o)); }
        if (lane == 0) {
            atomicMax(&smax[0], __float_as_uint(wm));
            atomicMax(&smax[1], __float_as_uint(gm));
        }
        __syncthreads();
        const float vmax = __uint_as_float(smax[0]) * __uint_as_float(smax[1]) * 1.0001f;
        if (vmax > 1e-30f && vmax < 3.0e38f) {
            fx_scale = gg.fx_lim / vmax * 0.999999f;
            fx_inv = 1.f / fx_scale;
        }
    }
    // A tile's grad_out rows are REQUESTED a tile ahead — behind the last MFMA phase of the previous tile, so that they fly under its last scatter
    // (loaded at the top of the tile they cost an exposed round trip per tile: 7.6 k of a workgroup's 190 k ticks, twice) — as raw words; the
    // "row outside the brick / chunk beyond Cout" zeroing happens when the tile starts.
    constexpr bool AHEAD = NKC == 1;   // (with more chunks the raw words do not fit next to the scatter's registers: 53 spilled at two chunks)
    f32x4 graw[(AHEAD && (!B16 || SPL)) ? KCMAX : 1][4];
    // B16 (bf16 rows): the row's MFMA operands as raw words, [kc][mf] = channels kc * 32 + 16 h + 8 mf .. + 7 (requested a tile ahead when NKC == 1)
    f32x4 grb[(B16 && !SPL) ? KCMAX : 1][2];
    auto load_rows_b16 = [&](bool ok, int v) {
#pragma unroll
        for (int kc = 0; kc < KCMAX; ++kc) {
            if (kc >= nkc) break;
            const bool okg = ok && kc * 32 + 16 * h < p.Cout;
            const bf16_t *gp_ = reinterpret_cast<const bf16_t *>(p.g) + (okg ? ((long)b * p.N + v) * p.Cout + kc * 32 + 16 * h : 0);
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(gp_ + 8 * mf);
                grb[kc][mf][0] = okg ? t[0] : 0.f; grb[kc][mf][1] = okg ? t[1] : 0.f; grb[kc][mf][2] = okg ? t[2] : 0.f; grb[kc][mf][3] = okg ? t[3] : 0.f;
            }
        }
    };
    auto request_rows = [&](int tile) {
        const int row = tile * 32 + j;
        int rw, rh;
        const int t = divmod_fast(row, gg.bw, rw);
        const int rd = divmod_fast(t, gg.bh, rh);
        const int vd = bd0 + rd, vh = bh0 + rh, vw = bw0 + rw;
        const bool ok = row < R && vd < p.D && vh < p.H && vw < p.W;
        const int v = ok ? (vd * p.H + vh) * p.W + vw : 0;
        if (B16 && !SPL) { load_rows_b16(ok, v); return; }
#pragma unroll
        for (int kc = 0; kc < (AHEAD ? KCMAX : 1); ++kc) {
            const bool okg = ok && kc * 32 + 16 * h < p.Cout;
            const long gi = okg ? ((long)b * p.N + v) * p.Cout + kc * 32 + 16 * h : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) graw[kc][e] = act_load4(gin, gi + 4 * e);
        }
    };
    if (AHEAD && wave < ntiles) request_rows(wave);
    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int row = tile * 32 + j;
        int rw, rh;
        const int rt = divmod_fast(row, gg.bw, rw);
        const int rd = divmod_fast(rt, gg.bh, rh);
        const int vd = bd0 + rd, vh = bh0 + rh, vw = bw0 + rw;
        const bool ok = row < R && vd < p.D && vh < p.H && vw < p.W;
        const int v = ok ? (vd * p.H + vh) * p.W + vw : 0;
        float gl[B16 ? 1 : KCMAX][16];   // this voxel's grad_out row (16 of each 32-channel chunk), loaded once for all tap groups
        bf16x8 gbo[B16 ? KCMAX : 1][2], gbo_lo[SPL ? KCMAX : 1][2];   // B16: the same as MFMA operands (SPL: hi and lo term)
        if (B16 && !SPL) {
            if (!AHEAD) load_rows_b16(ok, v);
#pragma unroll
            for (int kc = 0; kc < KCMAX; ++kc) {
                if (kc >= nkc) break;
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) { const float w4[4] = {grb[kc][mf][0], grb[kc][mf][1], grb[kc][mf][2], grb[kc][mf][3]}; gbo[kc][mf] = bf16x8_from_words(w4); }
            }
        }
        if (SPL) {
#pragma unroll
            for (int kc = 0; kc < KCMAX; ++kc) {
                if (kc >= nkc) break;
                const bool okg = ok && kc * 32 + 16 * h < p.Cout;
                const long gi = okg ? ((long)b * p.N + v) * p.Cout + kc * 32 + 16 * h : 0;
                float v16[16];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f32x4 t;
                    if constexpr (AHEAD) t = graw[kc][e];
                    else t = act_load4(gin, gi + 4 * e);
                    v16[4 * e] = okg ? t[0] : 0.f; v16[4 * e + 1] = okg ? t[1] : 0.f; v16[4 * e + 2] = okg ? t[2] : 0.f; v16[4 * e + 3] = okg ? t[3] : 0.f;
                }
                split_bf16x8(v16, gbo[kc][0], gbo_lo[kc][0]);
                split_bf16x8(v16 + 8, gbo[kc][1], gbo_lo[kc][1]);
            }
        }
#pragma unroll
        for (int kc = 0; kc < (B16 ? 0 : KCMAX); ++kc) {
            if (kc >= nkc) break;
            const bool okg = ok && kc * 32 + 16 * h < p.Cout;
            const long gi = okg ? ((long)b * p.N + v) * p.Cout + kc * 32 + 16 * h : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x4 t;
                if constexpr (AHEAD) t = graw[kc][e];
                else t = act_load4(gin, gi + 4 * e);
                gl[kc][4 * e] = okg ? t[0] : 0.f; gl[kc][4 * e + 1] = okg ? t[1] : 0.f; gl[kc][4 * e + 2] = okg ? t[2] : 0.f; gl[kc][4 * e + 3] = okg ? t[3] : 0.f;
            }
        }
        for (int grp = 0; grp < gg.ngroups; ++grp) {
            const float *Bg = Bs + (size_t)grp * p.CoutP * 32;
            // the offsets of this lane's four (voxel, tap) samples of the group are requested BEFORE the MFMA phase: loaded where they are
            // used, every scatter step began with an exposed L2 round trip (the kernel reacted to neither fewer VALU instructions nor the
            // removal of its LDS atomics — it was waiting)
            float offv[4][3];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int tq = grp * TG + 2 * r4 + h;
                const float *op = p.off + ((long)b * 3 * p.K + 3 * (tq < p.K ? tq : 0)) * p.N + v;
                offv[r4][0] = op[0]; offv[r4][1] = op[p.N]; offv[r4][2] = op[2 * (long)p.N];
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kc = 0; kc < KCMAX; ++kc) {
                if (kc >= nkc) break;
                if (B16) {
                    const unsigned short *a0 = Bh + ((size_t)(grp * 2) * 32 + j) * RH + kc * 32 + 16 * h;   // row j = (t8, c4), hi term; lo term 32 rows on
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) {
                        const f32x4 th = *reinterpret_cast<const f32x4 *>(a0 + 8 * mf), tl = *reinterpret_cast<const f32x4 *>(a0 + (size_t)32 * RH + 8 * mf);
                        const float h4[4] = {th[0], th[1], th[2], th[3]}, l4[4] = {tl[0], tl[1], tl[2], tl[3]};
                        acc = mfma_32x32x16_bf16(bf16x8_from_words(h4), gbo[kc][mf], acc);
                        acc = mfma_32x32x16_bf16(bf16x8_from_words(l4), gbo[kc][mf], acc);
                        if (SPL) acc = mfma_32x32x16_bf16(bf16x8_from_words(h4), gbo_lo[kc][mf], acc);
                    }
                    continue;
                }
                const float *arow = Bg + (kc * 32 + 16 * h) * 32 + j;   // A[i = (t8, c4) = j][k = co]
#pragma unroll
                for (int st = 0; st < 16; ++st) acc = mfma_32x32x2(arow[st * 32], gl[kc][st], acc);
            }
            if (AHEAD && grp == gg.ngroups - 1 && tile + nwaves < ntiles) request_rows(tile + nwaves);   // uniform
            // acc[r]: MFMA row (r&3) + 8*(r>>2) + 4h  ->  c4 = r & 3, t8 = 2*(r>>2) + h;   column = voxel j
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                // (no early exits: every lane reaches the ballot of the far-sample queue below; lanes without a sample carry valid = false)
                const int tap_ = grp * TG + 2 * r4 + h;
                const bool tv = ok && tap_ < p.K;
                const int tap = tap_ < p.K ? tap_ : 0;
                int ti, tj, tk;
                if (p.kw == 3 && p.kh == 3) { ti = tap / 9; const int rr = tap - 9 * ti; tj = rr / 3; tk = rr - 3 * tj; }   // uniform
                else tap_decode(tap, p.kw, p.kh, ti, tj, tk);
                // the one sampling rule (deform_sample.h: deform_im2col_cuda.cuh:244-259); outside the guard the cell is (0,0,0) and `valid` is false
                int zd, zh, zw;
                float ld, lh, lw;
                const bool inside = sample_cell3(offv[r4][0], offv[r4][1], offv[r4][2], vd + ti * p.dd - p.pd, vh + tj * p.dh - p.ph, vw + tk * p.dw - p.pw,
                                                 p.D, p.H, p.W, zd, zh, zw, ld, lh, lw);
                const bool valid = tv && inside;
                const float fd[2] = {1.f - ld, ld}, fh[2] = {1.f - lh, lh}, fw[2] = {1.f - lw, lw};
                const float wdh[4] = {fd[0] * fh[0], fd[0] * fh[1], fd[1] * fh[0], fd[1] * fh[1]};
                const int xd = zd - od, xh = zh - oh, xw = zw - ow;
                // all eight corners inside window + guard cells?  (a corner outside the volume is then in a guard cell)
                const bool near = valid & ((unsigned)xd < (unsigned)(ND - 1)) & ((unsigned)xh < (unsigned)(NH - 1)) & ((unsigned)xw < (unsigned)(NW - 1));
                if (near) {
                    unsigned long long *cell = WinI + (xd * SH + xh) * SW + xw;
                    const float fws[2] = {fw[0] * fx_scale, fw[1] * fx_scale};
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int cd = (q >> 2) & 1, ch = (q >> 1) & 1, cw = q & 1;
                        const float ws = wdh[2 * cd + ch] * fws[cw];
#pragma unroll
                        for (int pr = 0; pr < CS / 2; ++pr) {
                            // round(Col * ws) without v_rndne + v_cvt: |Col * ws| <= fx_lim < 2^22, so adding 1.5 * 2^23 in ONE fma leaves the
                            // integer (round to nearest even of the exact product) in the low mantissa bits: bits = 0x4B400000 + i
                            // (two channels at once: the pair sits in adjacent accumulator registers -> one v_pk_fma_f32; the constant comes from a
                            //  register — as a literal it forces the scalar v_fmaak form)
                            typedef float f32x2_t __attribute__((ext_vector_type(2)));
                            const f32x2_t cp = {acc[4 * r4 + 2 * pr], acc[4 * r4 + 2 * pr + 1]}, wp2 = {ws, ws}, mg = {fx_magic, fx_magic};
                            const f32x2_t tp = cp * wp2 + mg;
                            // bits(tp) = 0x4B400000 + i (i < 0 included: no wrap, |i| < 2^22), so the register PAIR read as one 64-bit integer is
                            // (M + i1) * 2^32 + (M + i0), and the packed word i1 * 2^32 + i0 (two's complement, borrow included) is that minus the
                            // constant M * (2^32 + 1): ONE 64-bit add instead of subtract / sign / add-with-borrow per field
                            const unsigned long long tt = ((unsigned long long)__float_as_uint(tp[1]) << 32) | (unsigned long long)__float_as_uint(tp[0]);
                            const unsigned long long pk = tt - 0x4B4000004B400000ull;
                            atomicAdd(cell + (cd * SH * SW + ch * SW + cw) + pr * PS, pk);
                        }
                    }
                }
                // far samples: into the wave's queue (prefix slot from the ballot); an overflowing batch takes the global atomics directly
                const bool far_s = valid && !near;
                const unsigned long long fm = __ballot(far_s);
                if (fm) {   // wave-uniform
                    const int nf = __popcll(fm);
                    if (qcount + nf <= GX_QW) {
                        if (far_s) {
                            float *rec = Qw + (qcount + __popcll(fm & ((1ull << lane) - 1ull))) * 8;
                            rec[0] = __int_as_float(((zd + 1) << 20) | ((zh + 1) << 10) | (zw + 1));
                            rec[1] = ld; rec[2] = lh; rec[3] = lw;
                            rec[4] = acc[4 * r4]; rec[5] = acc[4 * r4 + 1]; rec[6] = acc[4 * r4 + 2]; rec[7] = acc[4 * r4 + 3];
                        }
                        qcount += nf;
                    } else if (far_s) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int cd = (q >> 2) & 1, ch = (q >> 1) & 1, cw = q & 1;
                            const int cz = zd + cd, cy = zh + ch, cx = zw + cw;
                            if ((unsigned)cz >= (unsigned)p.D || (unsigned)cy >= (unsigned)p.H || (unsigned)cx >= (unsigned)p.W) continue;
                            const float wq = wdh[2 * cd + ch] * fw[cw];
                            float *dst = p.gx + ((long)b * p.N + (long)(cz * p.H + cy) * p.W + cx) * p.C + slice * CS;
#pragma unroll
                            for (int c = 0; c < CS; ++c) atomicAdd(dst + c, acc[4 * r4 + c] * wq);
                        }
                    }
                }
            }
            if (qcount > GX_QW / 2) { gx_drain_far(p, Qw, qcount, b, slice, lane); qcount = 0; }   // uniform
        }
    }
    if (qcount) gx_drain_far(p, Qw, qcount, b, slice, lane);
    __syncthreads();
    // flush the window proper (no guard cells) in compact (d, h, w) order: scratch[brick][slice][cell] as float4, as the first generation does
    f32x4 *dst = reinterpret_cast<f32x4 *>(scratch) + ((long)brick * gg.nslices + slice) * gg.wvox_max;
    for (int e = tid; e < wvox; e += blockDim.x) {
        const int d = e / WHW, r = e - d * WHW, hh = r / WW, w = r - hh * WW;
        const int c = ((d + gld) * SH + (hh + glh)) * SW + (w + glw);
        f32x4 o;
#pragma unroll
        for (int pr = 0; pr < CS / 2; ++pr) {
            const long long sv = (long long)WinI[pr * PS + c];
            const int lo = (int)(unsigned)(sv & 0xffffffffll);          // sum of the low fields (sign carried by the two's complement)
            const long long hi = (sv - (long long)lo) >> 32;            // sum of the high fields
            o[2 * pr] = (float)lo * fx_inv;
            o[2 * pr + 1] = (float)hi * fx_inv;
        }
        dst[e] = o;
    }
}

// grad_input[b][voxel][slice*4 .. +3] += sum over the bricks whose window covers the voxel (ascending brick order).
__global__ __launch_bounds__(256) void cl_deform_gx_gather_kernel(DeformBwdArgs p, GxGeom gg, const float *__restrict__ scratch)
{
    const long total = (long)p.B * p.N * gg.nslices;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int v = (int)(e % p.N);
        const int slice = (int)((e / p.N) % gg.nslices);
        const int b = (int)(e / p.N / gg.nslices);
        const int w = v % p.W, hh = (v / p.W) % p.H, d = v / (p.W * p.H);
        // brick i covers [i*bs - HALO, i*bs + bs + HALO): smallest such i satisfies i*bs > x - HALO - bs
        const int td = d - HALO - gg.bd + 1, th = hh - HALO - gg.bh + 1, tw = w - HALO - gg.bw + 1;
        const int id_lo = td > 0 ? cdiv(td, gg.bd) : 0, ih_lo = th > 0 ? cdiv(th, gg.bh) : 0, iw_lo = tw > 0 ? cdiv(tw, gg.bw) : 0;
        const int id_hi = min(gg.nbd - 1, (d + HALO) / gg.bd), ih_hi = min(gg.nbh - 1, (hh + HALO) / gg.bh), iw_hi = min(gg.nbw - 1, (w + HALO) / gg.bw);
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int id = id_lo; id <= id_hi; ++id) {
            int wd0, WD;
            gx_window(id * gg.bd, gg.bd, p.D, wd0, WD);
            for (int ih = ih_lo; ih <= ih_hi; ++ih) {
                int wh0, WH;
                gx_window(ih * gg.bh, gg.bh, p.H, wh0, WH);
                for (int iw = iw_lo; iw <= iw_hi; ++iw) {
                    int ww0, WW;
                    gx_window(iw * gg.bw, gg.bw, p.W, ww0, WW);
                    const long brick = (((long)b * gg.nbd + id) * gg.nbh + ih) * gg.nbw + iw;
                    const int cell = ((d - wd0) * WH + (hh - wh0)) * WW + (w - ww0);
                    const f32x4 t = reinterpret_cast<const f32x4 *>(scratch)[(brick * gg.nslices + slice) * gg.wvox_max + cell];
                    sum[0] += t[0]; sum[1] += t[1]; sum[2] += t[2]; sum[3] += t[3];
                }
            }
        }
        f32x4 *dst = reinterpret_cast<f32x4 *>(p.gx + ((long)b * p.N + v) * p.C + slice * CS);
        f32x4 o = *dst;
        o[0] += sum[0]; o[1] += sum[1]; o[2] += sum[2]; o[3] += sum[3];
        *dst = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
static GxGeom pick_gx_geom(const DeformBwdArgs &a)
{
    GxGeom g;
    // Bricks are long along W: the 32 voxels of a wave tile are then consecutive in w, and with spatially smooth offsets
    // their corner cells are consecutive fp64 cells of the window — distinct LDS banks (a cubic 8x8x8 brick puts 4 short
    // w-rows in a tile, whose cells collide).  DLKA_GX_BRICK=cube restores the cubic shape for A/B runs.
    constexpr bool cube = false;
    if (cube) {
        g.bd = a.D < 8 ? a.D : 8; g.bh = a.H < 8 ? a.H : 8; g.bw = a.W < 8 ? a.W : 8;
    } else {
        g.bw = a.W < 32 ? a.W : 32;
        g.bh = a.H < 4 ? a.H : 4;
        int rest = 512 / (g.bw * g.bh);
        if (rest < 1) rest = 1;
        g.bd = a.D < rest ? a.D : rest;
        if (g.bd > 8) g.bd = 8;
    }
    g.nslices = a.C / CS;
    g.ngroups = cdiv(a.K, TG);
    auto blocks = [&]() { return (long)a.B * cdiv(a.D, g.bd) * cdiv(a.H, g.bh) * cdiv(a.W, g.bw) * g.nslices; };
    // enough workgroups to cover the 256 CUs; keep at least one 32-row tile per brick
    while (blocks() < 256 && g.bd * g.bh * g.bw > 32) {
        if (g.bd >= g.bh && g.bd > 1) g.bd = cdiv(g.bd, 2);
        else if (g.bh > 1) g.bh = cdiv(g.bh, 2);
        else g.bw = cdiv(g.bw, 2);
    }
    g.nbd = cdiv(a.D, g.bd); g.nbh = cdiv(a.H, g.bh); g.nbw = cdiv(a.W, g.bw);
    auto ext = [](int bs, int size) { int e = bs + 2 * HALO; return e > size ? size : e; };
    g.wvox_max = ext(g.bd, a.D) * ext(g.bh, a.H) * ext(g.bw, a.W);
    return g;
}

size_t cl_deform_bwd2_scratch_floats(const DeformBwdArgs &a)
{
    if (a.C % CS) return 0;
    const GxGeom g = pick_gx_geom(a);
    return (size_t)a.B * g.nbd * g.nbh * g.nbw * g.nslices * g.wvox_max * CS;
}

// Slices of the input-channel chunks for grad_offset.  At C = 256 / 4^3 the (voxel-block, tap) grid is 27 workgroups, each
// running 64 chunk GEMMs in sequence (82 us, profiles/archive/r01n); slicing the channel chunks brings it to ~216 workgroups.
int cl_deform_goff_ccsplit(const DeformBwdArgs &a)
{
    const int mblocks = cdiv(a.M, 128);
    int tsplit = 1;
    while (mblocks * tsplit < 512 && tsplit < a.K) ++tsplit;
    tsplit = cdiv(a.K, cdiv(a.K, tsplit));
    const int ncc = a.C / 32;
    int split = 1;
    while (mblocks * tsplit * split * 2 <= 256 && split * 2 <= ncc) split *= 2;
    return cdiv(ncc, cdiv(ncc, split));
}

int launch_cl_deform_bwd2(const DeformBwdArgs &a, float *scratch, hipStream_t st)
{
    if (a.C % 32 || a.CoutP % 32) return DLKA_ERR_UNSUPPORTED;
    if ((long)a.M * a.C * 4 >= (1l << 31)) return DLKA_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    if (a.samp && (long)a.K * a.M * a.C * (a.act_bf16 ? 2 : 4) >= (1l << 31)) return DLKA_ERR_UNSUPPORTED;
    if (a.goff) {
        const int mblocks = cdiv(a.M, 128);
        int tsplit = 1;
        while (mblocks * tsplit < 512 && tsplit < a.K) ++tsplit;
        const int tpb = cdiv(a.K, tsplit);
        tsplit = cdiv(a.K, tpb);
        const int nkc = a.CoutP / 32;
        // (the first "lane = voxel" gather kernel, 225-250 us against 150 at 32^3, was removed in round 2)
        const int ccsplit = cl_deform_goff_ccsplit(a);
        DeformBwdArgs ag = a;
        ag.cc_per_block = cdiv(a.C / 32, ccsplit);
        if (a.goff_cpad && ccsplit > 1) return DLKA_ERR_UNSUPPORTED;   // the packed layout needs the single-writer path
        if (ccsplit > 1 && !a.goff_zeroed && launch_zero(a.goff, (size_t)a.B * 3 * a.K * a.N * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
        // 16-row waves where the stage is large and one 32-channel chunk wide (stage 0).  DLKA_GOFF16_MIN_ROWS lowers the row threshold so that small
        // test shapes take this kernel too (not cached: tests toggle it).
        bool done16 = false;
        {
            const char *e16 = getenv("DLKA_GOFF16_MIN_ROWS");
            const bool shape16 = a.C == 32 && a.Cout == 32 && a.CoutP == 32 && !a.goff_cpad && (long)a.B * 3 * a.K * a.N * 4 < (1l << 31);
            if (shape16 && (e16 ? a.M >= atoi(e16) : a.M >= 65536 - 127)) {
                // 4-wave workgroups at 168 registers, three waves per SIMD.  (8-wave workgroups at 128 registers — four per SIMD — spill 60 - 70 registers
                // in three of the four variants: 217 vs 149 us fp32, 150 vs 122 us bf16; the kernel above: 158 / 146 us on the same box.)
                constexpr int WV = 4;
                const int mb16 = cdiv(a.M, 16 * WV);
                dim3 grid16(mb16), block16(64 * WV);
                ag.xcd_nx = 0;
                if (xcd_swizzle_enabled() && mb16 >= xcd_min_blocks()) { ag.xcd_nx = mb16; grid16.x = xcd_grid(mb16); }
#define DLKA_G16(TT, SS) { auto k = cl_deform_goff16_kernel<TT, SS, WV, 3>; DLKA_LAUNCH(k, grid16, block16, 0, st, ag); }
#define DLKA_G16B(SS) { auto k = cl_deform_goff16_kernel<bf16_t, SS, WV, 3, true>; DLKA_LAUNCH(k, grid16, block16, 0, st, ag); }
                if (a.act_bf16 && a.wp16) { if (a.samp) DLKA_G16B(true) else DLKA_G16B(false) }   // Col on the bf16 matrix cores
                else if (a.act_bf16) { if (a.samp) DLKA_G16(bf16_t, true) else DLKA_G16(bf16_t, false) }
                else { if (a.samp) DLKA_G16(float, true) else DLKA_G16(float, false) }
#undef DLKA_G16
#undef DLKA_G16B
                DLKA_CHECK_LAUNCH();
                done16 = true;
            }
        }
        dim3 grid(mblocks, tsplit, ccsplit), block(256);
        if (!done16) {
            ag.xcd_nx = 0;
            if (xcd_swizzle_enabled() && mblocks >= xcd_min_blocks()) { ag.xcd_nx = mblocks; grid.x = xcd_grid(mblocks); }
        }
        if (!done16) {
            // storing variant: grad_out rows in registers only at Cout = 32 / fp32 (measured at 32^3: 179 vs 186 us; bf16 149 vs 155 us the other way)
#define DLKA_GOFF2(NK, TT)                                                                                                           \
    {                                                                                                                                \
        if (a.samp && (NK) == 1 && sizeof(TT) == 4) { auto k = cl_deform_goff2_kernel<1, TT, true>; DLKA_LAUNCH(k, grid, block, 0, st, ag, tpb); } \
        else if (a.samp) { auto k = cl_deform_goff2_kernel<0, TT, true>; DLKA_LAUNCH(k, grid, block, 0, st, ag, tpb); }       \
        else { auto k = cl_deform_goff2_kernel<NK, TT, false>; DLKA_LAUNCH(k, grid, block, 0, st, ag, tpb); }                 \
    }
#define DLKA_GOFF2_B16(NK)                                                                                                           \
    {                                                                                                                                \
        if (a.samp) { auto k = cl_deform_goff2_kernel<NK, bf16_t, true, true>; DLKA_LAUNCH(k, grid, block, 0, st, ag, tpb); }        \
        else { auto k = cl_deform_goff2_kernel<NK, bf16_t, false, true>; DLKA_LAUNCH(k, grid, block, 0, st, ag, tpb); }              \
    }
#define DLKA_GOFF2_SPL(NK)                                                                                                           \
    {                                                                                                                                \
        if (a.samp) { auto k = cl_deform_goff2_kernel<NK, float, true, true>; DLKA_LAUNCH(k, grid, block, 0, st, ag, tpb); }         \
        else { auto k = cl_deform_goff2_kernel<NK, float, false, true>; DLKA_LAUNCH(k, grid, block, 0, st, ag, tpb); }               \
    }
            if (a.act_bf16 && a.wp16) {   // Col on the bf16 matrix cores (two-term weight records, raw grad_out words)
                if (nkc == 1) DLKA_GOFF2_B16(1) else if (nkc == 2) DLKA_GOFF2_B16(2) else DLKA_GOFF2_B16(0)
            }
            else if (a.wp16) {            // ... fp32 rows, two-term split (rows in registers at one chunk only: the split doubles them)
                if (nkc == 1) DLKA_GOFF2_SPL(1) else DLKA_GOFF2_SPL(0)
            }
            else if (a.act_bf16) {
                if (nkc == 1) DLKA_GOFF2(1, bf16_t) else if (nkc == 2) DLKA_GOFF2(2, bf16_t) else DLKA_GOFF2(0, bf16_t)
            }
            else if (nkc == 1) DLKA_GOFF2(1, float) else if (nkc == 2) DLKA_GOFF2(2, float) else DLKA_GOFF2(0, float)
#undef DLKA_GOFF2
#undef DLKA_GOFF2_B16
#undef DLKA_GOFF2_SPL
        }
        DLKA_CHECK_LAUNCH();
    }
    if (a.gx) {
        if (!scratch) return DLKA_ERR_WORKSPACE;
        const GxGeom g = pick_gx_geom(a);
        if (!a.gx_zeroed && launch_zero(a.gx, (size_t)a.B * a.N * a.C * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
        GxGeom gl_ = g;
        // fixed-point window (two channels per 64-bit integer cell): default where it measured faster — C <= 64 with all weight tiles
        // resident in LDS (252 vs 362 us at C=32 / 32^3; at C >= 128 the scale pre-pass costs more than the atomics it saves).
        // DLKA_GX_FIXED=0 forces the fp64 window, =1 the fixed-point one wherever it is possible (A/B runs; not cached: tests toggle it).
        const char *fx_env = getenv("DLKA_GX_FIXED");
        const long rk = (long)g.bd * g.bh * g.bw * a.K;
        const double fx_lim = 2147483648.0 / (double)rk - 1.0;   // rk * (fx_lim + 1/2) = 2^31 - rk/2 < 2^31
        const int fx_bits = fx_lim > 1.0 ? (int)floor(log2(fx_lim)) : 0;
        const size_t lds_win_fx = 16 + (size_t)(g.wvox_max + 64) * (CS / 2) * sizeof(double);
        const bool fx_possible = lds_win_fx + (size_t)g.ngroups * a.CoutP * 32 * sizeof(float) <= 150 * 1024 && a.CoutP <= 128 && g.ngroups * 32 <= 512 &&
                                 fx_bits >= 12;
        const bool fixed = fx_possible && (fx_env ? atoi(fx_env) != 0 : a.C <= 64);
        // (capped below 2^22: the second-generation kernel rounds with the 1.5 * 2^23 trick, exact for |value| < 2^22; tiny bricks would allow more)
        gl_.fx_lim = (float)(fx_lim < 4.0e6 ? fx_lim : 4.0e6);
        gl_.fx_magic = 12582912.f;
        // scalars, window + one trash cell per lane and channel plane (the fixed-point window packs two channels per cell)
        const size_t lds_win = 16 + (size_t)(g.wvox_max + 64) * (fixed ? CS / 2 : CS) * sizeof(double);
        const size_t lds_all = lds_win + (size_t)g.ngroups * a.CoutP * 32 * sizeof(float);
        gl_.resident = (lds_all <= 150 * 1024 && (a.CoutP <= 128 || !fixed)) ? 1 : 0;   // (Cout = 256: resident weights, grad_out rows re-read per item)
        constexpr bool far_taps = false;   // (measured: 390 vs 370 us at 32^3 — slower)
        gl_.tap_far = (far_taps && g.ngroups == 4) ? 1 : 0;
        const size_t lds = gl_.resident ? lds_all : lds_win + (size_t)a.CoutP * 32 * sizeof(float);
        if (lds > 160 * 1024) return DLKA_ERR_UNSUPPORTED;
#if !defined(HIPEMU)
        // dynamic LDS above 64 KB has to be enabled per function AND per device (a process may drive several GPUs,
        // e.g. nn.DataParallel in the reference trainers); one bit per device, set after the attribute call succeeded
        static std::atomic<uint64_t> attr_done{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return DLKA_ERR_LAUNCH;
        const uint64_t bit = 1ull << (dev & 63);
        if (!(attr_done.load(std::memory_order_acquire) & bit)) {
#define DLKA_FX2_FNS(SWv, SHv, SDv, NK) reinterpret_cast<const void *>(cl_deform_gx_fx2_kernel<SWv, SHv, SDv, float, NK>), reinterpret_cast<const void *>(cl_deform_gx_fx2_kernel<SWv, SHv, SDv, bf16_t, NK>), reinterpret_cast<const void *>(cl_deform_gx_fx2_kernel<SWv, SHv, SDv, bf16_t, NK, true>), reinterpret_cast<const void *>(cl_deform_gx_fx2_kernel<SWv, SHv, SDv, float, NK, true>)
            const void *fns[28] = {reinterpret_cast<const void *>(cl_deform_gx_kernel<false>), reinterpret_cast<const void *>(cl_deform_gx_kernel<true>),
                                   reinterpret_cast<const void *>(cl_deform_gx_kernel<false, bf16_t>), reinterpret_cast<const void *>(cl_deform_gx_kernel<true, bf16_t>),
                                   DLKA_FX2_FNS(18, 10, 14, 0), DLKA_FX2_FNS(18, 10, 14, 1), DLKA_FX2_FNS(18, 10, 14, 2),
                                   DLKA_FX2_FNS(34, 10, 10, 0), DLKA_FX2_FNS(34, 10, 10, 1), DLKA_FX2_FNS(34, 10, 10, 2)};
#undef DLKA_FX2_FNS
            for (int f = 0; f < 28; ++f)
                if (hipFuncSetAttribute(fns[f], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DLKA_ERR_LAUNCH;
            attr_done.fetch_or(bit, std::memory_order_release);
        }
#endif
        const int bricks = a.B * g.nbd * g.nbh * g.nbw;
        const int items = bricks * g.nslices;
        // A/B switch.  The item order (slices of a brick adjacent) cuts the fabric traffic further, 279 -> 199 MB per launch at 32^3, but costs
        // 1 % of the whole step (15.6 vs 15.44 ms, two runs each): the eight slices of a brick then hit the same L2 channels at once.  Off.
        constexpr bool item_grid = false;
        gl_.item_grid = item_grid ? 1 : 0;
        const int nx = item_grid ? items : bricks;
        gl_.xcd_nx = (xcd_swizzle_enabled() && nx >= xcd_min_blocks()) ? nx : 0;
        const dim3 gx_grid(gl_.xcd_nx ? xcd_grid(nx) : nx, item_grid ? 1 : g.nslices);
        static int gx_threads = 0;
        if (!gx_threads) gx_threads = 512;
        // second-generation fixed-point kernel (compile-time window strides + guard cells) where one of its window shapes fits;
        // DLKA_GX_FIXED=2 keeps the first generation for A/B runs
        if (fixed && gl_.resident && !item_grid && !(fx_env && atoi(fx_env) == 2)) {
            auto ext2 = [](int bs, int size) { const int a = size + 2, b = bs + 2 * HALO; return a < b ? a : b; };   // window + guard cells, worst brick
            const int nd = ext2(g.bd, a.D), nh = ext2(g.bh, a.H), nw = ext2(g.bw, a.W);
            const bool b16 = a.wp16 != nullptr;   // (wp16 itself is not read here: the kernel stages its own bf16 tiles from wp) — bf16 rows, or fp32 rows split in two terms
            const size_t wbytes = b16 ? gx3_b16_wbytes(g.ngroups, a.CoutP) : (size_t)g.ngroups * a.CoutP * 32 * sizeof(float);
            const size_t qbytes = (size_t)8 * GX_QW * 8 * sizeof(float);   // 8 waves x GX_QW far-sample records
#define DLKA_GX2(SWv, SHv, SDv)                                                                                                    \
    if (nw <= SWv && nh <= SHv && nd <= SDv && 16 + (size_t)SDv * SHv * SWv * (CS / 2) * 8 + wbytes + qbytes <= 150 * 1024) {       \
        const size_t lds2 = 16 + (size_t)SDv * SHv * SWv * (CS / 2) * 8 + wbytes + qbytes;                                         \
        const int nk_ = a.CoutP / 32;                                                                                              \
        if (b16 && !a.act_bf16) {                                                                                                  \
            if (nk_ == 1) { auto k = cl_deform_gx_fx2_kernel<SWv, SHv, SDv, float, 1, true>; DLKA_LAUNCH(k, gx_grid, dim3(512), lds2, st, a, gl_, scratch); }       \
            else if (nk_ == 2) { auto k = cl_deform_gx_fx2_kernel<SWv, SHv, SDv, float, 2, true>; DLKA_LAUNCH(k, gx_grid, dim3(512), lds2, st, a, gl_, scratch); }  \
            else { auto k = cl_deform_gx_fx2_kernel<SWv, SHv, SDv, float, 0, true>; DLKA_LAUNCH(k, gx_grid, dim3(512), lds2, st, a, gl_, scratch); }                \
        } else if (b16) {                                                                                                          \
            if (nk_ == 1) { auto k = cl_deform_gx_fx2_kernel<SWv, SHv, SDv, bf16_t, 1, true>; DLKA_LAUNCH(k, gx_grid, dim3(512), lds2, st, a, gl_, scratch); }      \
            else if (nk_ == 2) { auto k = cl_deform_gx_fx2_kernel<SWv, SHv, SDv, bf16_t, 2, true>; DLKA_LAUNCH(k, gx_grid, dim3(512), lds2, st, a, gl_, scratch); } \
            else { auto k = cl_deform_gx_fx2_kernel<SWv, SHv, SDv, bf16_t, 0, true>; DLKA_LAUNCH(k, gx_grid, dim3(512), lds2, st, a, gl_, scratch); }               \
        } else if (a.act_bf16) {                                                                                                   \
            if (nk_ == 1) { auto k = cl_deform_gx_fx2_kernel<SWv, SHv, SDv, bf16_t, 1>; DLKA_LAUNCH(k, gx_grid, dim3(512), lds2, st, a, gl_, scratch); }      \
            else if (nk_ == 2) { auto k = cl_deform_gx_fx2_kernel<SWv, SHv, SDv, bf16_t, 2>; DLKA_LAUNCH(k, gx_grid, dim3(512), lds2, st, a, gl_, scratch); } \
            else { auto k = cl_deform_gx_fx2_kernel<SWv, SHv, SDv, bf16_t, 0>; DLKA_LAUNCH(k, gx_grid, dim3(512), lds2, st, a, gl_, scratch); }               \
        } else {                                                                                                                   \
            if (nk_ == 1) { auto k = cl_deform_gx_fx2_kernel<SWv, SHv, SDv, float, 1>; DLKA_LAUNCH(k, gx_grid, dim3(512), lds2, st, a, gl_, scratch); }       \
            else if (nk_ == 2) { auto k = cl_deform_gx_fx2_kernel<SWv, SHv, SDv, float, 2>; DLKA_LAUNCH(k, gx_grid, dim3(512), lds2, st, a, gl_, scratch); }  \
            else { auto k = cl_deform_gx_fx2_kernel<SWv, SHv, SDv, float, 0>; DLKA_LAUNCH(k, gx_grid, dim3(512), lds2, st, a, gl_, scratch); }                \
        }                                                                                                                          \
        launched = true;                                                                                                           \
    }
            bool launched = false;
            DLKA_GX2(18, 10, 14)
            else DLKA_GX2(34, 10, 10)
#undef DLKA_GX2
            if (launched) {
                DLKA_CHECK_LAUNCH();
                const long total = (long)a.B * a.N * g.nslices;
                long gb = cdivl(total, 256);
                if (gb > 4096) gb = 4096;
                DLKA_LAUNCH(cl_deform_gx_gather_kernel, dim3((unsigned)gb), dim3(256), 0, st, a, g, (const float *)scratch);
                DLKA_CHECK_LAUNCH();
                return DLKA_OK;
            }
        }
        if (a.act_bf16) {
            if (fixed) { auto k = cl_deform_gx_kernel<true, bf16_t>; DLKA_LAUNCH(k, gx_grid, dim3(gx_threads), lds, st, a, gl_, scratch); }
            else { auto k = cl_deform_gx_kernel<false, bf16_t>; DLKA_LAUNCH(k, gx_grid, dim3(gx_threads), lds, st, a, gl_, scratch); }
        }
        else if (fixed) { auto k = cl_deform_gx_kernel<true>; DLKA_LAUNCH(k, gx_grid, dim3(gx_threads), lds, st, a, gl_, scratch); }
        else { auto k = cl_deform_gx_kernel<false>; DLKA_LAUNCH(k, gx_grid, dim3(gx_threads), lds, st, a, gl_, scratch); }
        DLKA_CHECK_LAUNCH();
        const long total = (long)a.B * a.N * g.nslices;
        long gb = cdivl(total, 256);
        if (gb > 4096) gb = 4096;
        DLKA_LAUNCH(cl_deform_gx_gather_kernel, dim3((unsigned)gb), dim3(256), 0, st, a, g, (const float *)scratch);
        DLKA_CHECK_LAUNCH();
    }
    return DLKA_OK;
}

}  // namespace dlka
