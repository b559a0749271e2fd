// Channels-last implicit-GEMM convolution on the matrix cores (fp32-input MFMA, exact fp32).
//
// One kernel serves every dense contraction of the D-LKA block in token / NDHWC layout:
//     out[m][n] = bias[n] + sum_{tap} sum_{c} A(m, tap, c) * Wp[tap][c][n]          m = (b, voxel), n = out channel
//   AMODE 0  A = in[b][voxel + tap offset][c]                (zero padded)   1x1x1 projections, offset-predict conv
//   AMODE 1  A = trilinear sample of in at voxel + tap + Delta(b, tap, voxel)     deformable conv (D3D semantics)
//   AMODE 2  A = in_planar[b][c][voxel + tap offset]         (zero padded)   data-gradient of the offset conv
// (AMODE 1 never materialises the 27*C-wide column matrix the reference writes to HBM,
//  3D/dcn/src/cuda/deform_conv_cuda.cu:95 — samples go from registers straight into the MFMA A operand.)
//
// Mapping (gfx950, wave64): a wave owns a 32-row M tile and all N tiles (NT x 32 columns) -> NT accumulators of
// v_mfma_f32_32x32x2_f32.  Lane l = (i = l & 31, h = l >> 5) feeds A[row i][k-slot h]; k-slot h of step s is
// channel 16*h + s of the current 32-channel chunk, so every lane reads 16 CONTIGUOUS channels (four 16-byte loads)
// of its row — the summation order over k is permuted, nothing else.  The 4 waves of a workgroup (128 rows) share
// the 32 x NP weight chunk through LDS.  Small-spatial stages (C=256 at 4^3) are filled by splitting the taps over
// blockIdx.y and accumulating with fp32 atomics.
#include <stdlib.h>

#include "cl_arow.h"
#include "dlka_kernels.h"

namespace dlka {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// SPLIT: the contraction runs on the bf16 matrix cores with two-term split operands (dlka_intrin.h: hi*hi + hi*lo + lo*hi,
// fp32 accumulation, ~1e-5 relative) instead of the exact fp32-input MFMA: 6 x 32 cycles per 32-channel unit and column tile
// instead of 16 x 64.  The prepared weights then hold, per unit, the blocks [hi|lo][k half mf][lane half h][NP][8 bf16].
// SPLIT = 3: three-term operands and the six products above 2^-24 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid): fp32-equivalent
// (the dropped terms are below fp32's own rounding), 12 x 32 cycles per unit and tile — 2.7x the fp32-input MFMA.  Used for the FORWARD
// offset conv, whose output decides floor() of the sampling positions and must not move by 1e-5; records [(part*2+mf)*2+h], part 0..2.
// T: storage of the channels-last activation tensors (AMODE 0 `in`; OMODE 0 `out` / `out2` / `aux` / `aux2`): float, or bf16_t (DLKA_BF16
// token path).  A bf16 A operand is its own high term, so the split contraction drops the a_lo products.  Planar tensors stay fp32.
template <int AMODE, int OMODE, int NT, int SPLIT = 0, typename T = float>
__global__ __launch_bounds__(256) void cl_igemm_kernel(IgemmArgs p)
{
    constexpr bool A16 = AMODE == 0 && sizeof(T) == 2;   // A values are exact bf16
    constexpr int NPB = NT * 32;                 // columns handled by this block
    constexpr int UF = SPLIT == 3 ? 48 : 32;     // floats of prepared weights per unit and column
    constexpr int BV = (UF * NPB / 4 + 255) / 256;   // float4 of the weight chunk each of the 256 threads stages
    constexpr int BSZ = (OMODE == 1 && UF * NPB < 2 * 32 * 33) ? 2 * 32 * 33 : UF * NPB;   // OMODE 1 reuses Bs for 4 transpose tiles
    __shared__ __attribute__((aligned(16))) float Bs[2][BSZ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int bx = DLKA_XCD_BX(p.xcd_nx);                // XCD-aware tile order: an XCD owns a contiguous range of row blocks
    if (bx < 0) return;
    const int m = (bx * 4 + wave) * 32 + i;              // this lane's A row
    const bool row_ok = m < p.M;
    const int b = row_ok ? m / p.N : 0;
    const int v = row_ok ? m - b * p.N : 0;
    const int w0 = v % p.W, h0 = (v / p.W) % p.H, d0 = v / (p.W * p.H);
    const int n0 = blockIdx.z * NPB;    // first column

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nchunk = p.CinP / 32;
    const int unit_lo = blockIdx.y * p.units_per_split;
    const int unit_hi = min(p.K * nchunk, unit_lo + p.units_per_split);

    // software pipeline: while the MFMAs of unit u run, the weight chunk and the A values of unit u+1 are in flight
    const BufRsrc rin = make_rsrc(p.in, AMODE == 2 ? (size_t)p.B * p.CinReal * p.N * 4 : (size_t)p.M * p.Cin * sizeof(T));
    ARow<AMODE, T> arow;
    f32x4 breg[BV];   // ext_vector_type: stays in registers across iterations (HIP's float4 struct did not)
    float a_cur[16], a_nxt[16];
#define DLKA_LOAD_B(unit_)                                                                         \
    {                                                                                              \
        int ck_;                                                                                   \
        const int tap_ = divmod_fast((unit_), nchunk, ck_);                                        \
        const float *src_ = p.wp + ((long)tap_ * nchunk + ck_) * UF * p.NP + (SPLIT ? 0 : n0);      \
        _Pragma("unroll") for (int e = 0; e < BV; ++e) {                                           \
            const int idx_ = tid + e * 256;                                                        \
            if (SPLIT) {   /* 4 * SPLIT segments (part, mf, h) of NPB 16-byte column records */     \
                const int seg_ = idx_ / NPB, col_ = idx_ - seg_ * NPB;                              \
                if (idx_ < UF * NPB / 4) breg[e] = reinterpret_cast<const f32x4 *>(src_)[(long)seg_ * p.NP + n0 + col_]; \
            } else {                                                                               \
                const int rr_ = idx_ / (NPB / 4), c4_ = idx_ - rr_ * (NPB / 4);                    \
                breg[e] = reinterpret_cast<const f32x4 *>(src_ + (long)rr_ * p.NP)[c4_];          \
            }                                                                                      \
        }                                                                                          \
    }
    if (unit_lo < unit_hi) {
        DLKA_LOAD_B(unit_lo)
        int ck;
        const int tap = divmod_fast(unit_lo, nchunk, ck);
        arow.fetch(p, rin, tap, ck, h, row_ok, b, v, d0, h0, w0, a_nxt);
    }
    int buf = 0;
    for (int unit = unit_lo; unit < unit_hi; ++unit, buf ^= 1) {
#pragma unroll
        for (int e = 0; e < BV; ++e)
            if (tid + e * 256 < UF * NPB / 4) reinterpret_cast<f32x4 *>(Bs[buf])[tid + e * 256] = breg[e];
#pragma unroll
        for (int e = 0; e < 16; ++e) a_cur[e] = a_nxt[e];
        __syncthreads();   // Bs[buf] staged; Bs[buf^1] (read two iterations ago) is free again
#ifndef DLKA_ABL   // -DDLKA_ABL=bits builds a TIMING-ONLY ablation of the three-term kernel (wrong results): 1 no split arithmetic, 2 one MFMA per
#define DLKA_ABL 0  // tile instead of six, 4 no A fetch in the loop, 8 no weight fetch in the loop
#endif
        if (unit + 1 < unit_hi) {
            if (!(SPLIT == 3 && (DLKA_ABL & 8))) DLKA_LOAD_B(unit + 1)
            int ck;
            const int tap = divmod_fast(unit + 1, nchunk, ck);
            if (!(SPLIT == 3 && (DLKA_ABL & 4))) arow.fetch(p, rin, tap, ck, h, row_ok, b, v, d0, h0, w0, a_nxt);
        }
#ifdef DLKA_IGLP
        __builtin_amdgcn_iglp_opt(DLKA_IGLP);   // (experiment: scripts/build_variant.sh NAME -DDLKA_IGLP=0|1)
#endif
        if (SPLIT == 3) {
            const bf16x8 *B16 = reinterpret_cast<const bf16x8 *>(Bs[buf]);   // [(part*2 + mf)*2 + h][NPB] records of 8 bf16, part = hi, mid, lo
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                bf16x8 ahi, amid, alo;
                if (DLKA_ABL & 1) { ahi = bf16x8_from_words(a_cur + 8 * mf); amid = bf16x8_from_words(a_cur + 8 * mf + 4); alo = ahi; }
                else split3_bf16x8(a_cur + 8 * mf, ahi, amid, alo);
#ifdef DLKA_MFMA_PM   // (experiment: product-major, tile-minor — consecutive MFMAs on different accumulators)
                bf16x8 bhi[NT], bmid[NT], blo[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    bhi[t] = B16[((0 * 2 + mf) * 2 + h) * NPB + t * 32 + i]; bmid[t] = B16[((1 * 2 + mf) * 2 + h) * NPB + t * 32 + i];
                    blo[t] = B16[((2 * 2 + mf) * 2 + h) * NPB + t * 32 + i];
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma_32x32x16_bf16(alo, bhi[t], acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma_32x32x16_bf16(ahi, blo[t], acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma_32x32x16_bf16(amid, bmid[t], acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma_32x32x16_bf16(amid, bhi[t], acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma_32x32x16_bf16(ahi, bmid[t], acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma_32x32x16_bf16(ahi, bhi[t], acc[t]);
#else
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 bhi = B16[((0 * 2 + mf) * 2 + h) * NPB + t * 32 + i], bmid = B16[((1 * 2 + mf) * 2 + h) * NPB + t * 32 + i],
                                 blo = B16[((2 * 2 + mf) * 2 + h) * NPB + t * 32 + i];
                    acc[t] = mfma_32x32x16_bf16(alo, bhi, acc[t]);   // small terms first
                    if (DLKA_ABL & 2) continue;
                    acc[t] = mfma_32x32x16_bf16(ahi, blo, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(amid, bmid, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(amid, bhi, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(ahi, bmid, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(ahi, bhi, acc[t]);
                }
#endif
            }
        } else if (SPLIT) {
            const bf16x8 *B16 = reinterpret_cast<const bf16x8 *>(Bs[buf]);   // [(part*2 + mf)*2 + h][NPB] records of 8 bf16
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                bf16x8 ahi, alo;
                if (A16) ahi = alo = bf16x8_from_words(a_cur + 4 * mf);   // raw bf16 rows (ARow): their own high term, no low term
                else if (AMODE == 2 && p.a_packed) unpack_split2x8(a_cur + 8 * mf, ahi, alo);   // split once by the producer (cl_deform_goff2_kernel)
                else split_bf16x8(a_cur + 8 * mf, ahi, alo);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 bhi = B16[((0 * 2 + mf) * 2 + h) * NPB + t * 32 + i], blo = B16[((1 * 2 + mf) * 2 + h) * NPB + t * 32 + i];
                    if (!A16) acc[t] = mfma_32x32x16_bf16(alo, bhi, acc[t]);   // small terms first (a bf16 A has no low term)
                    acc[t] = mfma_32x32x16_bf16(ahi, blo, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(ahi, bhi, acc[t]);
                }
            }
        } else {
            const float *brow = Bs[buf] + (16 * h) * NPB + i;
#pragma unroll
            for (int st = 0; st < 16; ++st) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma_32x32x2(a_cur[st], brow[st * NPB + t * 32], acc[t]);
            }
        }
    }

#undef DLKA_LOAD_B
    // ---- epilogue: D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
    const int mbase = (bx * 4 + wave) * 32;
    const bool split = gridDim.y > 1;
    if (OMODE == 1) {
        // Planar output [B][Cout][N]: in the D layout a store instruction would scatter 32 lanes over 32 planes (4 bytes
        // per cache line).  Transpose each 32 x 32 tile through LDS so that lanes run over voxels: every store / atomic
        // then covers 128 contiguous bytes of one plane.  (The weight buffers are free once the main loop is done.)
        __syncthreads();
        float *Tt = &Bs[0][0] + wave * (32 * 33);
        const int mr = mbase + i;
        const bool rok = mr < p.M;
        const int bb = rok ? mr / p.N : 0, vv = rok ? mr - bb * p.N : 0;
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
                float *dst = p.out + ((long)bb * p.Cout + n) * p.N + vv;
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
            T *outp = reinterpret_cast<T *>(p.out), *out2p = reinterpret_cast<T *>(p.out2);
            const T *auxp = reinterpret_cast<const T *>(p.aux), *aux2p = reinterpret_cast<const T *>(p.aux2);
            const float auxv = (p.epi >= 2 && !(split && blockIdx.y != 0)) ? ((sizeof(T) == 2 && p.aux_f32) ? p.aux[o] : act_load1(auxp, o)) : 0.f;
            if (split) {   // partial sums meet in an fp32 buffer (for bf16 storage the caller converts it afterwards)
                if (p.epi == 3 && blockIdx.y == 0) val += auxv;   // the residual / fan-in term enters once
                atomicAdd(p.out + o, val);
            } else if (p.epi == 0) {
                act_store1(outp, o, val);
            } else if (p.epi == 1) {
                act_store1(outp, o, val);
                act_store1(out2p, o, gelu_erf(val));
            } else if (p.epi == 2) {
                act_store1(outp, o, val);
                act_store1(out2p, o, auxv * val);
            } else if (p.epi == 3) {
                act_store1(outp, o, val + auxv);
            } else {
                act_store1(outp, o, val * auxv);
                act_store1(out2p, o, val * act_load1(aux2p, o));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight preparation:  reference layout W[co][ci][tap]  ->  Wp[tap'][k][n]  (zero padded to CinP x NP)
//   mode 0 (forward):        k = ci, n = co, tap' = tap
//   mode 1 (data gradient):  k = co, n = ci, tap' = K-1-tap   (correlation with the flipped kernel)
//   mode 2 (column matrix):  k = co, n = ci, tap' = tap       (Col = G * W[:, :, tap] in the deformable backward)
// ---------------------------------------------------------------------------------------------
// prepared value of element (tap' tp, k, n); mode & 8 selects the bf16 split layout (see prep_store)
__device__ __forceinline__ float prep_value(const float *__restrict__ w, int Cout, int Cin, int K, int mode, int tp, int k, int n)
{
    const int m = mode & 7;
    if (m == 0) return (k < Cin && n < Cout) ? w[((long)n * Cin + k) * K + tp] : 0.f;
    if (m == 1) return (k < Cout && n < Cin) ? w[((long)k * Cin + n) * K + (K - 1 - tp)] : 0.f;
    return (k < Cout && n < Cin) ? w[((long)k * Cin + n) * K + tp] : 0.f;
}

// plain: wp[(tp*KP + k)*NP + n] = val.   split (mode & 8): unit = (tp, k / 32) of 32*NP floats holds bf16 records
// [(part*2 + mf)*2 + h][NP][8] with k % 32 = 16h + 8mf + e, part 0 = hi, 1 = lo.
__device__ __forceinline__ void prep_store(float *__restrict__ wp, int KP, int NP, int mode, int tp, int k, int n, float val)
{
    if (!(mode & 24)) { wp[((long)tp * KP + k) * NP + n] = val; return; }
    const int kk = k & 31, h = kk >> 4, mf = (kk >> 3) & 1, e = kk & 7;
    if (mode & 16) {   // three-term records: a unit takes 48 * NP floats
        unsigned short *u = reinterpret_cast<unsigned short *>(wp + ((long)tp * (KP / 32) + (k >> 5)) * 48 * NP);
        const unsigned short hi = bf16_bits(val);
        const float r1 = val - bf16_value(hi);
        const unsigned short mid = bf16_bits(r1), lo = bf16_bits(r1 - bf16_value(mid));
        u[((long)((0 * 2 + mf) * 2 + h) * NP + n) * 8 + e] = hi;
        u[((long)((1 * 2 + mf) * 2 + h) * NP + n) * 8 + e] = mid;
        u[((long)((2 * 2 + mf) * 2 + h) * NP + n) * 8 + e] = lo;
        return;
    }
    unsigned short *u = reinterpret_cast<unsigned short *>(wp + ((long)tp * KP + (k & ~31)) * NP);
    const unsigned short hi = bf16_bits(val), lo = bf16_bits(val - bf16_value(hi));
    u[((long)((0 * 2 + mf) * 2 + h) * NP + n) * 8 + e] = hi;
    u[((long)((1 * 2 + mf) * 2 + h) * NP + n) * 8 + e] = lo;
}

__global__ void cl_prep_weight_kernel(const float *__restrict__ w, float *__restrict__ wp, int Cout, int Cin, int K, int KP, int NP, int mode)
{
    const long n_el = (long)K * KP * NP;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_el; e += (long)gridDim.x * blockDim.x) {
        const int n = (int)(e % NP), k = (int)((e / NP) % KP), tp = (int)(e / NP / KP);
        prep_store(wp, KP, NP, mode, tp, k, n, prep_value(w, Cout, Cin, K, mode, tp, k, n));
    }
}

int launch_cl_prep_weight(const float *w, float *wp, int Cout, int Cin, int K, int KP, int NP, int mode, hipStream_t st)
{
    const long n_el = (long)K * KP * NP;
    long blocks = cdivl(n_el, 256);
    if (blocks > 2048) blocks = 2048;
    DLKA_LAUNCH(cl_prep_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, wp, Cout, Cin, K, KP, NP, mode);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// All weight re-layouts of one D-LKA block in ONE launch (the per-conv prep launches were ~17 x 5 us per block).
// job.mode 0/1/2: cl_prep_weight_kernel's modes; 3: depthwise W[c][tap] -> Wp[tap][c]; 4: the same with flipped taps; 5: zero fill.
// A workgroup covers PREP_TABLE_CHUNK elements of ONE job (b.first[k] = first workgroup of job k): the job is found once per workgroup — the first
// version searched the job list linearly for EVERY element of a grid-stride loop (21 us average per launch in the nn.Module path of the full net).
constexpr int PREP_TABLE_CHUNK = 2048;
// Dense jobs (modes 0 - 2) with K <= 27 taps go tile by tile through LDS (round 5): a workgroup owns (8 k, 32 n, all K taps) of the prepared tensor.  In the
// reference layout W[co][ci][tap] that tile is 32 contiguous runs of 8 K floats (mode 0: k = ci) or 8 runs of 32 K floats (modes 1 / 2: n = ci): read as such, re-laid
// in LDS, and written as 128-byte rows (plain) or 16-byte bf16 records, 512 contiguous bytes per (tap, term) (split layouts: the 8 k of a record are the tile's 8).
// The element-per-lane form below read 4-byte pieces Cin K 4 (mode 0) or K 4 bytes apart and wrote 2-byte pieces 16 bytes apart: 47 us for the C = 256 block's
// 28 MB (0.6 TB/s), 16 us at C = 128 (profiles/r08z_tblock_stage3).
constexpr int PREP_TILE_KMAX = 27;
struct alignas(16) PrepU4 { unsigned x, y, z, w; };
__host__ __device__ inline bool prep_job_tiled(const PrepJob &j)
{
    return (j.mode & 7) <= 2 && !(j.mode & 32) && j.K >= 8 && j.K <= PREP_TILE_KMAX && j.KP % 32 == 0 && j.NP % 32 == 0 && j.Cin % 32 == 0 &&
           (long)j.Cout * j.Cin * j.K * 4 < (1l << 31);   // (bit 32: DLKA_PREP_TILED=0)
}
__host__ __device__ inline long prep_job_blocks(const PrepJob &j)
{
    return prep_job_tiled(j) ? (long)(j.KP / 8) * (j.NP / 32) : (j.n + PREP_TABLE_CHUNK - 1) / PREP_TABLE_CHUNK;
}

__device__ __forceinline__ void prep_job_tile(const PrepJob &j, int blk, float *tile)
{
    const int K = j.K, m = j.mode & 7;
    const int nbn = j.NP / 32, kb = blk / nbn, nb = blk % nbn;
    const int k0 = kb * 8, n0 = nb * 32;
    const int tid = threadIdx.x;
    // mode 0 (k = ci, n = co): W[n][k0 .. k0 + 8)[tap] = a run of 8 K contiguous floats per n -> tile[nn][kk * K + tap], 32 rows of stride 8 K + 1
    // modes 1 / 2 (k = co, n = ci): W[k][n0 .. n0 + 32)[tap] = a run of 32 K contiguous floats per k -> tile[kk][nn * K + tap], 8 rows of stride 32 K + 1
    // (Cin % 32 == 0: a row is wholly inside or wholly outside the tensor.)  Dword loads — parameters carved from a flat buffer are only 4-byte aligned — one
    // row (mode 0: 8 K <= 216 floats) or a quarter row (32 K <= 4 x 256) per slot, and all 32 slots of a work-item are issued before the first LDS store:
    // the tile's latency is ONE trip to memory, not one per loop iteration.
    {
        const int run = (m == 0 ? 8 : 32) * K;
        const BufRsrc rs = make_rsrc(j.src, (size_t)j.Cout * j.Cin * K * 4);   // (range-checked loads: an offset of ~0 reads 0 — no branch per slot)
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int row = m == 0 ? u : (u >> 2), r = m == 0 ? tid : (u & 3) * 256 + tid;
            const bool in = m == 0 ? (n0 + row < j.Cout && k0 < j.Cin) : (k0 + row < j.Cout && n0 < j.Cin);
            const unsigned s0 = m == 0 ? (unsigned)((n0 + row) * j.Cin + k0) * (unsigned)K : (unsigned)((k0 + row) * j.Cin + n0) * (unsigned)K;
            v[u] = buf_load_f32(rs, (in && r < run) ? (s0 + (unsigned)r) * 4u : 0xffffffffu);
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int row = m == 0 ? u : (u >> 2), r = m == 0 ? tid : (u & 3) * 256 + tid;
            if (r < run) tile[row * (run + 1) + r] = v[u];
        }
    }
    __syncthreads();
    const int h = (k0 >> 4) & 1, mf = (k0 >> 3) & 1;
    for (int idx = tid; idx < 32 * K; idx += 256) {
        const int nn = idx & 31, tp = idx >> 5;
        const int ts = m == 1 ? K - 1 - tp : tp;   // (data gradient: the flipped kernel)
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = m == 0 ? tile[nn * (8 * K + 1) + e * K + ts] : tile[e * (32 * K + 1) + nn * K + ts];
        const int n = n0 + nn;
        if (!(j.mode & 24)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) j.dst[((long)tp * j.KP + k0 + e) * j.NP + n] = v[e];
            continue;
        }
        unsigned hi[4], md[4], lo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned short a[2], b[2], c[2];
#pragma unroll
            for (int z = 0; z < 2; ++z) {
                const float val = v[2 * q + z];
                a[z] = bf16_bits(val);
                const float r1 = val - bf16_value(a[z]);
                b[z] = bf16_bits(r1);
                c[z] = bf16_bits(r1 - bf16_value(b[z]));
            }
            hi[q] = (unsigned)a[0] | ((unsigned)a[1] << 16); md[q] = (unsigned)b[0] | ((unsigned)b[1] << 16); lo[q] = (unsigned)c[0] | ((unsigned)c[1] << 16);
        }
        if (j.mode & 16) {   // three-term records: a unit takes 48 NP floats
            PrepU4 *u = reinterpret_cast<PrepU4 *>(j.dst + ((long)tp * (j.KP / 32) + (k0 >> 5)) * 48 * j.NP);
            u[(long)((0 * 2 + mf) * 2 + h) * j.NP + n] = PrepU4{hi[0], hi[1], hi[2], hi[3]};
            u[(long)((1 * 2 + mf) * 2 + h) * j.NP + n] = PrepU4{md[0], md[1], md[2], md[3]};
            u[(long)((2 * 2 + mf) * 2 + h) * j.NP + n] = PrepU4{lo[0], lo[1], lo[2], lo[3]};
        } else {             // two-term records (second term = bf16(val - hi), the three-term form's "mid")
            PrepU4 *u = reinterpret_cast<PrepU4 *>(j.dst + ((long)tp * j.KP + (k0 & ~31)) * j.NP);
            u[(long)((0 * 2 + mf) * 2 + h) * j.NP + n] = PrepU4{hi[0], hi[1], hi[2], hi[3]};
            u[(long)((1 * 2 + mf) * 2 + h) * j.NP + n] = PrepU4{md[0], md[1], md[2], md[3]};
        }
    }
}

__device__ __forceinline__ void prep_job_chunk(const PrepJob &j, long blk)
{
    __shared__ float tile[32 * (8 * PREP_TILE_KMAX + 1)];   // >= 8 * (32 * PREP_TILE_KMAX + 1)
    if (prep_job_tiled(j)) { prep_job_tile(j, (int)blk, tile); return; }
    const long l0 = blk * PREP_TABLE_CHUNK;
    for (long l = l0 + threadIdx.x; l < l0 + PREP_TABLE_CHUNK && l < j.n; l += 256) {
        if (j.mode == 5) { j.dst[l] = 0.f; continue; }   // a zero fill riding along (split outputs of the forward pass)
        if (j.mode == 3 || j.mode == 4) {
            const int c = (int)(l % j.Cin), tap = (int)(l / j.Cin);
            j.dst[l] = j.src[(long)c * j.K + (j.mode == 4 ? j.K - 1 - tap : tap)];
        } else {
            const int n = (int)(l % j.NP), k = (int)((l / j.NP) % j.KP), tp = (int)(l / j.NP / j.KP);
            prep_store(j.dst, j.KP, j.NP, j.mode, tp, k, n, prep_value(j.src, j.Cout, j.Cin, j.K, j.mode, tp, k, n));
        }
    }
}

__global__ __launch_bounds__(256) void cl_prep_batch_kernel(PrepBatch b)
{
    int ji = 0;
    while (ji + 1 < b.njobs && (int)blockIdx.x >= b.first[ji + 1]) ++ji;   // (njobs <= PREP_MAX_JOBS, wave-uniform)
    prep_job_chunk(b.j[ji], (long)((int)blockIdx.x - b.first[ji]));
}

// The same re-layouts for MANY blocks in one launch: the job table lives in device memory (built once per model, the pointers do not
// change), `first[j]` = first workgroup of job j; a workgroup finds its job by bisection and covers one tile / PREP_TABLE_CHUNK elements of it.
// (this launch covers jobs [job_lo, job_hi); its workgroup 0 is workgroup first[job_lo] of the whole table)
__global__ __launch_bounds__(256) void cl_prep_table_kernel(const PrepJob *__restrict__ jobs, const int *__restrict__ first, int job_lo, int job_hi)
{
    const int wg = (int)blockIdx.x + first[job_lo];
    int lo = job_lo, hi = job_hi - 1;
    while (lo < hi) {   // last job whose first workgroup is <= wg
        const int mid = (lo + hi + 1) >> 1;
        if (first[mid] <= wg) lo = mid; else hi = mid - 1;
    }
    const PrepJob j = jobs[lo];
    prep_job_chunk(j, (long)(wg - first[lo]));
}

int cl_prep_table_blocks(const PrepJob &j) { return (int)prep_job_blocks(j); }

int launch_cl_prep_table(const PrepJob *jobs_dev, const int *first_dev, int job_lo, int job_hi, int nblocks, hipStream_t st)
{
    if (job_hi <= job_lo || nblocks <= 0) return DLKA_OK;
    DLKA_LAUNCH(cl_prep_table_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, jobs_dev, first_dev, job_lo, job_hi);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_prep_batch(const PrepBatch &b_, hipStream_t st)
{
    if (b_.njobs <= 0) return DLKA_OK;
    PrepBatch b = b_;
    int blk = 0;
    for (int k = 0; k < b.njobs; ++k) {
        b.first[k] = blk;
        blk += (int)prep_job_blocks(b.j[k]);
    }
    b.first[b.njobs] = blk;
    if (blk <= 0) return DLKA_OK;
    DLKA_LAUNCH(cl_prep_batch_kernel, dim3((unsigned)blk), dim3(256), 0, st, b);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

template <int AMODE, int OMODE, int SPLIT = 0, typename T = float>
static int launch_igemm_nt(const IgemmArgs &a, int splits, hipStream_t st)
{
    const int NT_total = a.NP / 32;
    // column tiles per block: all of them when there are plenty of row blocks, fewer (-> gridDim.z) for small M
    int NT = NT_total;
    const int mblocks = cdiv(a.M, 128);
    if (NT_total == 6) NT = 2;          // 192 / 384 columns (the 2-D block's offset-net data gradients at 28^2 / 14^2: 147 / 37 row blocks): two tiles per workgroup
    else if (NT_total == 12) NT = 2;    // (round 6; three / four tiles measured 184 against 160 us and 144 against 129: more, smaller workgroups balance the chip)
    else if (NT_total == 8 && mblocks * splits < 256) NT = (mblocks * splits * 2 < 256) ? 2 : 4;
    else if (NT_total == 4 && mblocks * splits < 256) NT = (mblocks * splits * 2 < 256) ? 1 : 2;
    // Round 6: 128 columns in ONE workgroup only where the row blocks alone balance the chip.  The 2-D block's 7x7 offset net at 56^2 (588 row blocks, 98 -> 128 columns) was 2.3
    // workgroups per CU, three resident: the CUs that took three set the time, the matrix pipe 44 % busy and 1.3 resident waves per SIMD on average (profiles/r10_notes.md);
    // two column halves = 1176 workgroups of half the LDS: 560 -> ~508 us per launch, the 2-D step 17.10 -> 16.74 ms.  (One tile per workgroup: slower — 471 against 339 us averaged
    // over the two nets — the A rows are fetched and split four times.)
    else if (NT_total == 4 && mblocks * splits < 1024) NT = 2;
    else if (NT_total == 2 && mblocks * splits < 128) NT = 1;
    {   // A/B runs: DLKA_IGEMM_NT=n (read per launch) forces n column tiles per workgroup where n divides the tile count
        const char *e = getenv("DLKA_IGEMM_NT");
        const int n = e ? atoi(e) : 0;
        if (n > 0 && NT_total % n == 0 && a.K > 1) NT = n;
    }
    dim3 grid(mblocks, splits, NT_total / NT), block(256);
    IgemmArgs ax = a;
    ax.xcd_nx = 0;
    if (xcd_swizzle_enabled() && a.K > 1 && mblocks >= (unsigned)xcd_min_blocks()) { ax.xcd_nx = mblocks; grid.x = xcd_grid(mblocks); }
#define DLKA_IG(NTV)                                              \
    {                                                             \
        auto k = cl_igemm_kernel<AMODE, OMODE, NTV, SPLIT, T>;    \
        DLKA_LAUNCH(k, grid, block, 0, st, ax);            \
    }
    switch (NT) {
        case 1: DLKA_IG(1) break;
        case 2: DLKA_IG(2) break;
        case 3: DLKA_IG(3) break;
        case 4: DLKA_IG(4) break;
        case 8: DLKA_IG(8) break;
        default: return DLKA_ERR_UNSUPPORTED;
    }
#undef DLKA_IG
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// Picks the tap split so that small-spatial stages still fill the chip; returns the number of splits.
int cl_igemm_pick_splits(int M, int units, int epi, int K)
{
    if ((epi != 0 && epi != 3) || units == 1) return 1;   // epilogues 0 and 3 are linear in the accumulator: splittable
    // pointwise convs (K = 1) have C/32 <= 8 units: splitting 2 or 4 ways buys nothing that pays for the zero fill + atomics
    // (C = 64 / 16^3: 24.6 us split vs 11.8 us unsplit for the same GEMM, profiles/archive/r01n); they run unsplit on cl_pointwise.hip
    static int pw_min = -1;
    if (pw_min < 0) pw_min = 99;
    if (K == 1 && units < pw_min) return 1;
    const int mblocks = cdiv(M, 128);
    // Split partial sums meet in global fp32 atomics on the SAME addresses: measured on MI355X (profiles/archive/r01e), 216-way
    // splits of the C=256 / 4^3 offset conv cost 130 us, almost all of it same-address serialisation in L2.  Bound the
    // contention instead of chasing block count.
    static int cap = -1;
#ifndef DLKA_SPLIT_CAP
#define DLKA_SPLIT_CAP 32
#endif
    if (cap < 0) cap = DLKA_SPLIT_CAP;
    static int want = -1;   // workgroups to aim for (tuning knob; -DDLKA_SPLIT_WANT=n builds a variant for scripts/build_variant.sh)
#ifndef DLKA_SPLIT_WANT
#define DLKA_SPLIT_WANT 512
#endif
    if (want < 0) want = DLKA_SPLIT_WANT;
    int splits = 1;
    while (mblocks * splits < want && splits < units && splits < cap) ++splits;
    const int ups = cdiv(units, splits);
    return cdiv(units, ups);
}

int launch_cl_igemm(int amode, int omode, IgemmArgs a, int splits, hipStream_t st)
{
    if ((long)a.M * (amode == 2 ? a.CinReal : a.Cin) * 4 >= (1l << 31)) return DLKA_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    a.units_per_split = cdiv(a.K * (a.CinP / 32), splits);
    splits = cdiv(a.K * (a.CinP / 32), a.units_per_split);
    constexpr bool pw_v1 = false;
    if (!pw_v1 && amode == 0 && omode == 0 && a.K == 1 && splits == 1 && !a.split_bf16) {
        const int rc = launch_cl_pointwise(a, st);
        if (rc != DLKA_ERR_UNSUPPORTED) return rc;
    }
    if (a.zero.n > 0) {   // only the pointwise kernel carries riding zero fills: anything else gets them as a launch of their own
        if (launch_zero_batch(a.zero, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
        a.zero.n = 0;
    }
    if (amode == 0 && omode == 1 && splits == 1 && a.split_bf16 == 3) {   // the offset conv's forward at the wide stage: LDS-brick kernel
        const int rc = launch_cl_conv_brick3(a, st);
        if (rc != DLKA_ERR_UNSUPPORTED) return rc;
    }
    if (amode == 2 && omode == 0 && a.split_bf16 == 2 && (splits == 1) == (cl_conv_brick_split(a) == 1)) {   // the offset conv's data gradient: LDS-brick kernel
        const int rc = launch_cl_conv_brick(a, st);
        if (rc != DLKA_ERR_UNSUPPORTED) return rc;
    }
    // small volumes (the row tiling alone would not fill the chip): the contraction splits over the waves of a workgroup, deterministically (cl_conv_kw.hip).  The
    // C-ABI sequencing code has asked cl_conv_kw_applies() and passes splits = 1 for these (no zero fill, no fp32 staging buffer)
    if (splits == 1 && a.K > 1 && a.split_bf16 && cl_igemm_pick_splits(a.M, a.K * (a.CinP / 32), a.epi, a.K) > 1) {
        const int rc = launch_cl_conv_kw(amode, omode, a, st);
        if (rc != DLKA_ERR_UNSUPPORTED) return rc;
    }
    if (splits > 1 && !a.out_zeroed) {
        const long n = (long)a.M * a.Cout;
        if (launch_zero(a.out, (size_t)n * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
    }
    if (a.act_bf16) {   // bf16 activation storage (DLKA_BF16 token path): the two 27-tap convs of the block, two-term weight layout
        if (a.split_bf16 != 2) return DLKA_ERR_UNSUPPORTED;
        if (amode == 0 && omode == 1) {   // offset-predict conv forward (fp32 planar out): wave-granular kernel where the fp32 path uses it too
            const int rc = launch_cl_conv_wave(amode, omode, a, splits, st);
            if (rc != DLKA_ERR_UNSUPPORTED) return rc;
            return launch_igemm_nt<0, 1, 2, bf16_t>(a, splits, st);
        }
        if (amode == 2 && omode == 0) return launch_igemm_nt<2, 0, 2, bf16_t>(a, splits, st);   // its data gradient (fp32 planar in, bf16 out)
        return DLKA_ERR_UNSUPPORTED;
    }
    if (a.split_bf16) {   // bf16 x3 split contraction (the prepared weights must be in the split layout)
        {   // wave-granular variant first (no LDS staging, no barriers); the zero fill for split partial sums was done above
            const int rc = launch_cl_conv_wave(amode, omode, a, splits, st);
            if (rc != DLKA_ERR_UNSUPPORTED) return rc;
        }
        if (a.split_bf16 == 3) {
            if (amode == 0 && omode == 1) return launch_igemm_nt<0, 1, 3>(a, splits, st);
            if (amode == 0 && omode == 0) return launch_igemm_nt<0, 0, 3>(a, splits, st);
            return DLKA_ERR_UNSUPPORTED;
        }
        if (amode == 0 && omode == 0) return launch_igemm_nt<0, 0, 2>(a, splits, st);
        if (amode == 0 && omode == 1) return launch_igemm_nt<0, 1, 2>(a, splits, st);
        if (amode == 2 && omode == 0) return launch_igemm_nt<2, 0, 2>(a, splits, st);
        return DLKA_ERR_UNSUPPORTED;
    }
    if (amode == 0 && omode == 0) return launch_igemm_nt<0, 0>(a, splits, st);
    if (amode == 0 && omode == 1) return launch_igemm_nt<0, 1>(a, splits, st);
    if (amode == 1 && omode == 0) return launch_igemm_nt<1, 0>(a, splits, st);
    if (amode == 2 && omode == 0) return launch_igemm_nt<2, 0>(a, splits, st);
    return DLKA_ERR_UNSUPPORTED;
}

}  // namespace dlka
