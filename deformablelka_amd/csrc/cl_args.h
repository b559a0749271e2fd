// Argument blocks of the channels-last kernels (shared between the kernel files and the C-ABI sequencing code).
#pragma once
#include "dlka_common.h"

namespace dlka {

// Several zero fills in one launch (every dependent kernel node costs ~4.5 us of dispatch latency inside a hipGraph on
// MI355X, profiles/archive/r01n): regions are 16-byte aligned float arrays.
struct ZeroBatch {
    int n;
    float *p[8];
    long cnt[8];          // floats
    unsigned block0[9];   // first workgroup of each region (set by the launcher)
    bool overflow;        // more than 8 regions were added: the launcher refuses (a dropped zero fill would be a silent wrong answer)
    void add(float *ptr, size_t floats) { if (!ptr || !floats) return; if (n >= 8) { overflow = true; return; } p[n] = ptr; cnt[n] = (long)floats; ++n; }
};

// zero-fill workgroup `zb` of a ZeroBatch (4096 floats of one region), run by `nthreads` threads of any kernel
__device__ __forceinline__ void zero_batch_block(const ZeroBatch &b, unsigned zb, int tid, int nthreads)
{
    if (zb >= b.block0[b.n]) return;
    int r = 0;
    while (r + 1 < b.n && zb >= b.block0[r + 1]) ++r;
    float *p = b.p[r];
    const long n = b.cnt[r], base = (long)(zb - b.block0[r]) * 4096;
    for (int k = tid; k < 1024; k += nthreads) {
        const long i = base + (long)k * 4;
        if (i + 3 < n) *reinterpret_cast<f32x4 *>(p + i) = f32x4{0.f, 0.f, 0.f, 0.f};
        else for (long j = i; j < n && j < base + 4096; ++j) p[j] = 0.f;
    }
}

struct IgemmArgs {
    const float *in;     // AMODE 0/1: [B][N][Cin] channels-last; AMODE 2: [B][CinReal][N] planar
    const float *off;    // AMODE 1: [B][3K][N] planar offsets (reference layout)
    const float *wp;     // [K][CinP][NP] prepared weights (zero padded)
    const float *bias;   // [Cout] or null
    const float *aux;    // epilogue operand (channels-last [M][Cout]) or null
    const float *aux2;   // second epilogue operand (epi 4)
    float *out;          // OMODE 0: [M][Cout] channels-last; OMODE 1: [B][Cout][N] planar
    float *out2;         // second epilogue output or null
    float *out2_f32;     // act_bf16, epi 1 (pointwise kernel): ALSO store out2 = gelu(out) unrounded, fp32 [M][Cout] — the input of the fp32 chain
                         // that decides the sampling cells (dlka_capi_cl.hip: "offset-determining chain"); null = off
    int B, D, H, W, N, M;
    int Cin, CinReal, CinP, Cout, NP;
    int kd, kh, kw, pd, ph, pw, dd, dh, dw, K;
    int units_per_split; // work units (tap, 32-channel chunk) per blockIdx.y; K * CinP/32 when gridDim.y == 1
    int split_bf16;      // 2 / 3: contraction on the bf16 matrix cores with two- / three-term split operands (weights prepared with mode | 8 / | 16)
    int xcd_nx;          // set by the launcher: > 0 = gridDim.x is xcd_grid(xcd_nx) and blockIdx.x is mapped through xcd_item() (dlka_common.h)
    int out_zeroed;      // 1: the caller has already zero-filled `out` (split partial sums meet there in atomics)
    int a_packed;        // AMODE 2 + split_bf16 == 2: `in` holds pack_split2() words (CinReal = CinP channel planes per batch, zero padded)
    int act_bf16;        // 1: the activation tensors (in, aux, aux2, out, out2 — whatever is channels-last) are bf16 storage (DLKA_BF16 token path;
                         //    fp32 arithmetic, fp32 weights / bias); planar tensors (offsets, grad_offset) are always fp32
    int aux_f32;         // act_bf16 only: `aux` is fp32 all the same (the grad_input accumulation target of the deformable conv)
    ZeroBatch zero;      // optional zero fills that RIDE in this launch (cl_pointwise_kernel only: extra workgroups behind the row tiles) — one
    int zero_xblocks;    //   dependent graph node less per block and direction; zero_xblocks (set by the launcher) = those extra blockIdx.x
    int epi;             // 0: out = acc+bias | 1: out = acc+bias, out2 = gelu(out) | 2: out = acc+bias, out2 = aux*out | 3: out = acc+bias+aux
                         // 4: out = (acc+bias)*aux, out2 = (acc+bias)*aux2   (gate backward fused into proj_2's data gradient)
};

// Two pointwise convs back to back in one launch (cl_pointwise_pair_kernel, C <= 64):
//   forward  (bwd = 0):  t = in * W1 + bias1;  out1 = t;  out1b = a * t;   out2 = out1b * W2 + bias2 + b          (conv1 + gate, proj_2 + shortcut)
//   backward (bwd = 1):  t = in * W1;          out1 = t * a;  out1b = t * b;  out2 = out1 * W2                       (proj_2^T + gate backward, conv1^T)
// W1 / W2: prepared [C][C] operand layouts (cl_prep, mode 0 forward / mode 1 data gradient); all tensors channels-last [M][C].
struct PwPairArgs {
    const float *in, *wp1, *bias1, *wp2, *bias2, *a, *b;
    float *out1, *out1b, *out2;
    int M, C, bwd, act_bf16;
    ZeroBatch zero;      // riding zero fills, as in IgemmArgs
    int zero_blocks;     // set by the launcher
};

struct WgradArgs {
    const float *g;
    const float *in;    // [B][N][Cin] channels-last
    const float *off;   // AMODE 1: planar offsets [B][3K][N]
    const float *samp;  // AMODE 1, optional: [K][M][Cin] samples (fp32; bf16 when act_bf16) stored by cl_deform_goff2_kernel (DeformBwdArgs::samp) — no gather
    int samp_f16;       // fp32 activations only: the samples are IEEE halves (DeformBwdArgs::samp_f16)
    int samp_b16mfma;   // ... and the contraction runs on the bf16 matrix cores with two-term operands (DLKA_SAMP_B16MFMA, A/B)
    float *part;        // [chunks][K][CoutP][Cin] partial weight-gradient tiles, followed by [chunks][CoutP] partial bias sums
    float *bpart;       // = part + chunks*K*CoutP*Cin when the bias gradient is wanted, else null (set by the launcher)
    int B, D, H, W, N, M;
    int Cin, Cout, CoutP;
    int kd, kh, kw, pd, ph, pw, dd, dh, dw, K;
    int rows_per_chunk;  // multiple of 32
    int act_bf16;        // 1: `in` (and a channels-last `g`) are bf16 storage; a planar `g` (grad_offset) stays fp32
    int CT;              // Cin / 32
    int w16;             // set by the launcher: W % 16 == 0 (fast row addressing in cl_wgrad_dense_kernel)
    int no_win3;         // set by the launcher: DLKA_WGRAD_WIN3=0 (A/B switch: the three w-taps of a wave load their rows separately, as before round 4)
    int xcd_ny, xcd_nz, xcd_total;   // set by the launcher: > 0 = 1-D XCD-swizzled grid over (chunk, y, z) work items, see xcd_item()
    int g_cpad;          // GMODE 1 only, > 0: g holds pack_split2() words with g_cpad channel planes per batch (see DeformBwdArgs::goff_cpad)
    // Round 5 — the ZERO-PADDED copy of `in` (cl_wgrad.hip, cl_pad_copy_kernel): [B][DP][HP][WP][Cin] with the conv's reach as a halo, so that a tap's row is
    // "base of the output voxel + a wave-uniform tap offset" with no validity test at all.  pad != null selects the padded kernels.
    const float *pad;    // or null
    int DP, HP, WP;      // padded extents: D + (kd - 1) * dd etc.
};

struct DwArgs {
    const float *in;    // [B][D][H][W][C]
    const float *wp;    // [K][C] prepared weights (tap-major, channel contiguous)
    const float *bias;  // [C] or null
    float *out;         // [B][D][H][W][C]
    float *out_lo;      // fp32 kernels only, plain (non-GELU) epilogue: ALSO store the result rounded to bf16 there ([B][D][H][W][C] bf16), or null
    int xcd_nx;            // set by the launcher: > 0 = blockIdx.x is mapped through xcd_item()
    const float *gelu_x;   // optional fused epilogue (data gradient of dw 5^3 inside the D-LKA block): out = (acc + gelu_add) * gelu'(gelu_x)
    const float *gelu_add;
    int B, D, H, W, C;
    int act_bf16;          // 1: in / out / gelu_x / gelu_add are bf16 storage
    int kd, kh, pd, ph, pw, dd, dh;   // kw / dw are template parameters
    // LDS-brick kernels (cl_dwconv_lds.hip); blk == null: never taken
    float *blk;            // scratch for the class-blocked fp32 copy of `in` (blk_floats floats available)
    size_t blk_floats;
    int in_blocked;        // 1: blk already holds the blocked input (a preceding LDS-brick conv wrote it through ITS out_blk)
    float *out_blk;        // also write the result class-blocked (fp32) for a following depthwise conv of dilation out_blk_dil; null: no
    int out_blk_dil;
};

// two chained depthwise convs of a small volume in one launch (cl_dwpair.hip): in -> conv A -> outA -> conv B -> outB
struct DwPairArgs {
    const float *in;                 // [B][D][H][W][C], storage T (float | bf16 when act_bf16)
    const float *wpA, *wpB;          // prepared tap weights [K][C] (launch_cl_dw_prep_weight: forward or flipped form)
    const float *biasA, *biasB;      // [C] or null
    float *outA, *outB;              // storage T
    float *outA_lo, *outB_lo;        // fp32 kernels only: ALSO store the result rounded to bf16 there, or null
    const float *gelu_x, *gelu_add;  // optional epilogue of conv B: outB = (acc + gelu_add) * gelu'(gelu_x)   (storage T)
    int B, D, H, W, C;
    int kA, dA, pA, KA;              // conv A: cubic kernel size, dilation, padding, kA^3
    int kB, dB, pB, KB;
    int act_bf16;
};

// channels-last 2-D depthwise deformable conv (cl_ddw2d.hip): forward uses in / off / wp / out; backward in / off / wp / g / gx / goff / part
struct DwArgs2d {
    const float *in;     // [B][H][W][C]
    const float *off;    // [B][2 kh kw][H][W] planar (dy, dx) per tap
    const float *wp;     // [kh kw][C] prepared tap weights (launch_cl_dw_prep_weight, unflipped)
    const float *g;      // [B][H][W][C] grad_out
    float *out;          // [B][H][W][C]  (forward; may be null when only out_lo is wanted)
    float *out_lo;       // forward, optional: the result rounded to bf16, [B][H][W][C] bf16 (the mixed-precision DLKA_BF16 2-D block)
    float *gx;           // [B][H][W][C] fp32, zero-filled by the caller
    float *goff;         // [B][2 kh kw][H][W]
    float *part;         // cl_ddw2d_part_floats() floats of weight-gradient partials
    int B, H, W, C, kh, kw, ph, pw, dh, dw;
    int act_bf16;        // backward: in / g are bf16 storage (gx, goff, the weight-gradient partials stay fp32); the forward kernel is fp32-in only
};

struct DwWgradArgs {
    const float *g;     // [B][D][H][W][C]
    const float *in;    // [B][D][H][W][C]
    float *gwp;         // [K][C] fp32, zero-initialised
    float *gb;          // [C] fp32, zero-initialised bias gradient (column sums of g) or null
    int B, D, H, W, C;
    int kd, kh, pd, ph, pw, dd, dh;
    int rows_per_block;
    int act_bf16;       // 1: g / in are bf16 storage (the staging buffer gwp stays fp32)
    int xcd_nx;         // set by the launcher: > 0 = blockIdx.x is mapped through xcd_item()
};

struct PrepJob {
    const float *src;   // weight in the reference layout
    float *dst;         // prepared layout
    int Cout, Cin, K, KP, NP;
    int mode;           // 0 fwd, 1 data-grad (flipped), 2 column matrix (| 8 / | 16: bf16 split layouts); 3 depthwise, 4 depthwise flipped; 5 zero fill of dst
    long n;             // elements of dst
};
constexpr int PREP_MAX_JOBS = 32;   // (the wrapper block's forward pass puts its own six forms, the attention's fifteen and their zero fills in ONE launch)
struct PrepBatch {
    PrepJob j[PREP_MAX_JOBS];
    int njobs;
    long total;
    int first[PREP_MAX_JOBS + 1];   // set by launch_cl_prep_batch: first workgroup of job k (a workgroup covers PREP_TABLE_CHUNK elements of ONE job)
};

// one weight-gradient finalisation: fold the row-chunk partials (kind 0) or re-lay a depthwise [tap][c] buffer (kind 1)
struct FinalizeJob {
    const float *part;   // kind 0: [chunks][K][CoutP][Cin] partial tiles; kind 1: gwp [K][C]
    const float *bpart;  // kind 0: [chunks][CoutP] partial bias sums or null
    float *gw, *gb;      // outputs in the reference layout
    int chunks, K, CoutP, Cout, Cin;
    int kind;
    int tr;              // kind 0: 1 = "transposing" fold (set by the launcher: few partials, many outputs), see cl_wgrad_finalize_kernel
    long n;              // outputs (weights + bias entries)
    long block0;         // first workgroup of this job
};
struct FinalizeBatch {
    FinalizeJob j[12];   // (a D-LKA block has 8; the wrapper block appends its 3 so that one launch folds both)
    int njobs;
    long nblocks;
};

struct DeformBwdArgs {
    const float *in;    // [B][N][C] channels-last
    const float *off;   // [B][3K][N] planar
    const float *g;     // [M][Cout] channels-last grad_out
    const float *wp;    // [K][CoutP][C]: wp[tap][co][ci] = W[co][ci][tap], rows co >= Cout are zero
    const float *wp16;  // optional (act_bf16): the same weights as two-term bf16 records (prep mode 2 | 8: unit (tap, co / 32) = [part][mf][h][C][8], k = co) — the
                        //   grad_offset kernels then form Col on v_mfma_f32_32x32x16_bf16 and read no LDS weight tile; null = fp32-input MFMA from wp
    float *gx;          // [B][N][C] fp32, zero-initialised (atomics)
    float *goff;        // [B][3K][N] planar
    float *samp;        // optional [K][M][C], fp32 (bf16 when act_bf16): the grad_offset kernel also stores the trilinear samples S(m, tap, c) it has the corners of —
                        //   the deformable weight gradient then contracts G^T S as a dense stream instead of gathering again (cl_wgrad_samp_kernel)
    int B, D, H, W, N, M;
    int C, Cout, CoutP;
    int kd, kh, kw, pd, ph, pw, dd, dh, dw, K;
    int cc_per_block;   // 32-channel input chunks per blockIdx.z; grad_offset uses atomics when gridDim.z > 1
    int xcd_nx;         // set by the launchers (per kernel): > 0 = blockIdx.x is mapped through xcd_item()
    int gx_zeroed;      // 1: the caller has already zero-filled gx
    int goff_zeroed;    // 1: the caller has already zero-filled goff (needed when cl_deform_goff_ccsplit() > 1)
    int act_bf16;       // 1: in / g are bf16 storage (gx, goff stay fp32: atomics / planar)
    int samp_f16;       // fp32 activations only (round 6): store the samples as IEEE halves — half the bytes of the hand-over in both directions (250 -> 125 MB written per
                        //   stage-0 launch, the same read back by cl_wgrad_samp_kernel).  A sample is an interpolated activation: rounding it to 11 significant bits moves a
                        //   weight-gradient element by ~3e-4 of its own magnitude at worst over 65 536 random-sign terms (bf16's 8 bits would sit ON the 1e-3 contract);
                        //   grad_out and the accumulation stay fp32.  DLKA_SAMP_F16=0 / DLKA_EXACT_FP32 keep fp32 samples
    int goff_cpad;      // > 0: goff is written as pack_split2() words with goff_cpad channel planes per batch (planes >= 3K zero) for the
                        //      split-MFMA consumers (offset-conv data / weight gradient); 0: plain fp32 [B][3K][N]
};

}  // namespace dlka
