// Channels-last 2-D DEPTHWISE deformable convolution — the two large-kernel convs of the 2-D D-LKA block
// (2D/deformable_LKA/deformable_LKA.py:93-94: 5x5 pad 2 and 7x7 dilation 3 pad 9, groups = C, ONE offset field shared by all
// channels, bias-free), torchvision 0.12 `deform_conv2d` semantics (offset channels (dy, dx) per tap, guard
// q <= -1 || q >= size -> 0, per-corner zeroing; torchvision is un-vendored: the rule is restated from its published kernel, include/dlka.h).
//
//     out[m][c] = sum_tap w[c][tap] * S(m, tap, c),   S = bilinear sample of in[b][:, :, c] at  base(m, tap) + offset[b][2 tap .. +1][m]
//
// The reference goes through torchvision: an im2col buffer of C*K x B*H*W floats (722 MB / 1.4 GB at C = 96, 56^2, B = 24) and C
// separate 1 x K GEMMs.  Depthwise = no contraction over channels, and the sampling positions are the SAME for every channel, so in
// channels-last layout one (pixel, tap) costs one sampling description and four whole-row gathers: lane (r = lane >> 3, p = lane & 7)
// serves pixel r of the wave's 8 pixels and channels 4p .. 4p+3 of the current 32-channel chunk — every load instruction covers 8 whole
// 128-byte rows, the access pattern that gathers at L1 speed (cl_gather.h) — and all C/32 chunks of a pixel reuse the description.
// The matrix cores have nothing to do here (north_star: "depthwise / gather-bound, not a dense contraction").
//
//   cl_ddw2d_fwd_kernel      forward
//   cl_ddw2d_bwd_kernel      one traversal for grad_offset (channel reduction in registers + one DPP sum over the 8 lanes of a pixel, no
//                            atomics) and the weight-gradient partials (tap-outer loop: 4 accumulators per chunk and lane, folded across
//                            the pixel lanes with shuffles and across waves through LDS)
//   cl_ddw2d_gx_kernel       grad_input: tile scatter into an fp64 LDS window, see there
//   cl_ddw2d_fold_kernel     sums the per-block weight-gradient partials into the reference layout [C][1][kh][kw]
#include <stdlib.h>

#include <atomic>
#include "cl_args.h"
#include "cl_ddw2d_describe.h"
#include "dlka_kernels.h"

namespace dlka {


struct Ddw2dArgs {
    const float *in;     // [B][N][C] channels-last
    const float *off;    // [B][2K][N] planar (dy, dx) per tap
    const float *wp;     // [K][C] prepared tap weights
    const float *g;      // [B][N][C] grad_out (backward)
    float *out;          // forward: [B][N][C] (may be null)
    float *out_lo;       // forward: optional bf16 copy [B][N][C]
    float *gx;           // backward: [B][N][C], ZERO-FILLED by the caller (atomics)
    float *goff;         // backward: [B][2K][N]
    float *part;         // backward: [nblocks][K][C] weight-gradient partials
    int B, H, W, N, M, C, K, kh, kw, ph, pw, dh, dw;
    int px_per_block;    // backward: pixels per workgroup (multiple of 32)
    int tpg;             // backward kernel: taps per blockIdx.y (grad_offset and the weight-gradient partials are per tap: a tap split needs no atomics) — fills the chip at
                         // the 14 x 14 stage, where the pixels alone give 147 workgroups
    int xcd_nx;          // forward / backward kernels: > 0 = gridDim.x is xcd_grid(xcd_nx) and blockIdx.x is mapped through xcd_item() (dlka_common.h): an XCD owns a
                         // contiguous range of pixel blocks, i.e. whole images, so the corner rows its blocks gather stay in ITS L2.  Round 5: with the plain order
                         // (block b on XCD b % 8) every XCD walks every image: 1479 MB of L2 misses per launch of the forward kernel at (96, 56^2, B = 24) for
                         // 73 MB of tensors (profiles/pmc_traffic_lka2d.json), 6 TB/s — the kernel was bound by the fabric, not by its gathers.
};

// grid (ceil(M / 32), C / (32 * NCH)); block 256 = 4 waves x 8 pixels.  NCH channel chunks of 32 per block.
template <int NCH>
__global__ __launch_bounds__(256) void cl_ddw2d_fwd_kernel(Ddw2dArgs p)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane >> 3, pp = lane & 7;
    const int bx = DLKA_XCD_BX(p.xcd_nx);
    if (bx < 0) return;   // (padding block of the swizzled grid; uniform)
    const int m = bx * 32 + wave * 8 + r;
    const bool ok = m < p.M;
    const int b = ok ? m / p.N : 0, n = ok ? m - b * p.N : 0;
    const int x0 = n % p.W, y0 = n / p.W;
    const int c0 = blockIdx.y * NCH * 32 + 4 * pp;
    const int rowbytes = p.C * 4;
    const BufRsrc rin = make_rsrc(p.in, (size_t)p.M * p.C * 4), rw = make_rsrc(p.wp, (size_t)p.K * p.C * 4);
    f32x4 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *offp = p.off + (long)b * 2 * p.K * p.N + n;
    // A wave's timeline was K x (offset round trip -> description -> corner round trip -> FMAs) with nothing of tap t + 1 in flight while tap t computes (waves parked 49 % of
    // their time, issue-stalled 38 %, active 13 %: profiles/r10_sq_counters_lka2d_C96_56.csv).  Round 6: the two offsets of a tap are requested PF taps ahead, which takes
    // the first of the two round trips off the chain: 246 -> 239 us (7x7), 133 -> 130 (5x5) at (96, 56^2, B = 24), bit-identical — small, because the kernel is NOT
    // latency-bound: its 4 corner rows per (pixel, tap) are 5.7 GB of L1 reads per 7x7 launch = 24 TB/s, 60 % of the L1's 64 bytes per clock and CU.  (Corner rows + weights
    // of tap t + 1 in a second register set as well: 234 registers at three chunks instead of 66 — two waves per SIMD instead of seven; not kept.)
    constexpr int PF = 4;
    float oyn[PF], oxn[PF];   // offsets of taps t .. t + PF - 1, slot = tap % PF
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        oyn[u] = (ok && u < p.K) ? offp[(long)(2 * u) * p.N] : 0.f;
        oxn[u] = (ok && u < p.K) ? offp[(long)(2 * u + 1) * p.N] : 0.f;
    }
    for (int tap0 = 0; tap0 < p.K; tap0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int tap = tap0 + u;
            if (tap >= p.K) break;   // uniform
            const int ti = tap / p.kw, tj = tap - ti * p.kw;
            Tap2 s;
            describe2(s, oyn[u], oxn[u], b, y0 - p.ph + ti * p.dh, x0 - p.pw + tj * p.dw, p.H, p.W, p.N, rowbytes);
            if (!ok) s.off[0] = s.off[1] = s.off[2] = s.off[3] = DLKA_OOB;
            oyn[u] = (ok && tap + PF < p.K) ? offp[(long)(2 * (tap + PF)) * p.N] : 0.f;
            oxn[u] = (ok && tap + PF < p.K) ? offp[(long)(2 * (tap + PF) + 1) * p.N] : 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const unsigned cb = (unsigned)(c0 + 32 * c) * 4u;
                f32x4 x4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) x4[q] = buf_load_f32x4(rin, s.off[q] == DLKA_OOB ? DLKA_OOB : s.off[q] + cb);
                const f32x4 w4 = buf_load_f32x4(rw, (unsigned)(tap * p.C) * 4u + cb);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float sv = s.wt[0] * x4[0][e];
                    sv = fmaf(s.wt[1], x4[1][e], sv); sv = fmaf(s.wt[2], x4[2][e], sv); sv = fmaf(s.wt[3], x4[3][e], sv);
                    acc[c][e] = fmaf(w4[e], sv, acc[c][e]);
                }
            }
        }
    }
    if (!ok) return;
    if (p.out) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) *reinterpret_cast<f32x4 *>(p.out + (long)m * p.C + c0 + 32 * c) = acc[c];
    }
    if (p.out_lo) {   // uniform
        bf16_t *lo = reinterpret_cast<bf16_t *>(p.out_lo);
#pragma unroll
        for (int c = 0; c < NCH; ++c) act_store4(lo, (long)m * p.C + c0 + 32 * c, acc[c]);
    }
}

// grid (ceil(M / px_per_block)); block 256 = 4 waves; a wave walks its share of the block's pixels 8 at a time, for ONE tap at a time
// (tap-outer), all NCH chunks of the row per (pixel, tap).
template <int NCH, typename T>   // T: storage of `in` and `g` (float | bf16_t); arithmetic, grad_offset and the weight-gradient partials are fp32
__global__ __launch_bounds__(256) void cl_ddw2d_bwd_kernel(Ddw2dArgs p)
{
    constexpr unsigned SB = sizeof(T);
    __shared__ float red[4][NCH * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane >> 3, pp = lane & 7;
    const int rowbytes = p.C * SB;
    const BufRsrc rin = make_rsrc(p.in, (size_t)p.M * p.C * SB), rg = make_rsrc(p.g, (size_t)p.M * p.C * SB), rw = make_rsrc(p.wp, (size_t)p.K * p.C * 4);
    const int bx = DLKA_XCD_BX(p.xcd_nx);
    if (bx < 0) return;   // (padding block of the swizzled grid; uniform)
    const int m_lo = bx * p.px_per_block, m_hi = min(p.M, m_lo + p.px_per_block);
    const int tap_lo = blockIdx.y * p.tpg, tap_hi = min(p.K, tap_lo + p.tpg);
    for (int tap = tap_lo; tap < tap_hi; ++tap) {
        const int ti = tap / p.kw, tj = tap - ti * p.kw;
        f32x4 gwacc[NCH], w4[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            gwacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            w4[c] = buf_load_f32x4(rw, (unsigned)(tap * p.C + 32 * c + 4 * pp) * 4u);
        }
        for (int mb = m_lo + wave * 8; mb < m_hi; mb += 32) {
            const int m = mb + r;
            const bool ok = m < m_hi;
            const int b = ok ? m / p.N : 0, n = ok ? m - b * p.N : 0;
            const int x0 = n % p.W, y0 = n / p.W;
            const float *offp = p.off + ((long)b * 2 * p.K + 2 * tap) * p.N + n;
            Tap2 s;
            describe2(s, ok ? offp[0] : 0.f, ok ? offp[p.N] : 0.f, b, y0 - p.ph + ti * p.dh, x0 - p.pw + tj * p.dw, p.H, p.W, p.N, rowbytes);
            if (!ok) { s.off[0] = s.off[1] = s.off[2] = s.off[3] = DLKA_OOB; s.okm = 0; s.wt[0] = s.wt[1] = s.wt[2] = s.wt[3] = 0.f; }
            // d(sample)/dy = (1-lx) (x10 - x00) + lx (x11 - x01), d/dx = (1-ly) (x01 - x00) + ly (x11 - x10); corners outside the image read 0
            // (torchvision get_coordinate_weight: per-corner bounds only, no guard)
            const float hy = 1.f - s.ly, hx = 1.f - s.lx;
            const float dyw[4] = {-hx, -s.lx, hx, s.lx}, dxw[4] = {-hy, hy, -s.ly, s.ly};
            float goy = 0.f, gox = 0.f;
            const unsigned gb = ok ? (unsigned)m * (unsigned)rowbytes : DLKA_OOB;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const unsigned cb = (unsigned)(32 * c + 4 * pp) * SB;
                f32x4 x4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) x4[q] = act_buf_load4<T>(rin, s.off[q] == DLKA_OOB ? DLKA_OOB : s.off[q] + cb);
                const f32x4 g4 = act_buf_load4<T>(rg, gb == DLKA_OOB ? DLKA_OOB : gb + cb);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float sv = s.wt[0] * x4[0][e];
                    sv = fmaf(s.wt[1], x4[1][e], sv); sv = fmaf(s.wt[2], x4[2][e], sv); sv = fmaf(s.wt[3], x4[3][e], sv);
                    gwacc[c][e] = fmaf(g4[e], sv, gwacc[c][e]);
                    const float col = g4[e] * w4[c][e];                       // d loss / d sample
                    float dy = 0.f, dx = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { dy = fmaf(dyw[q], x4[q][e], dy); dx = fmaf(dxw[q], x4[q][e], dx); }
                    goy = fmaf(col, dy, goy); gox = fmaf(col, dx, gox);
                }
            }
            goy = sum8(goy);   // over the 8 channel pieces of the pixel
            gox = sum8(gox);
            if (ok && pp == 0) {
                float *gop = p.goff + ((long)b * 2 * p.K + 2 * tap) * p.N + n;
                gop[0] = goy;
                gop[p.N] = gox;
            }
        }
        // fold the weight gradient of this tap: over the 8 pixel lanes (DPP-free path: shuffles), then over the 4 waves (LDS)
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = gwacc[c][e];
                v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
                gwacc[c][e] = v;
            }
        __syncthreads();   // red free
        if (r == 0) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wave][32 * c + 4 * pp + e] = gwacc[c][e];
        }
        __syncthreads();
        for (int e = tid; e < NCH * 32; e += 256)
            p.part[((long)bx * p.K + tap) * p.C + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    }
}

// grad_input of the depthwise deformable conv:  gx[b][v][c] = sum over (pixel o, tap) whose sample touches v of  G[o][c] w[c][tap] wt_v.
// The reference (torchvision deformable_col2im) issues one global fp32 atomicAdd per (pixel, tap, channel, corner); measured here the same
// scheme in channels-last layout costs 12 ms at C = 96 / 56^2 / B = 24 (1.4e9 atomics on heavily shared rows, profiles/archive/r03e).  As in the
// 3-D block (cl_deform_bwd2.hip) the scatter therefore goes into an LDS window first: a workgroup owns a tile of OUTPUT pixels and a
// 4-channel slice, its window = the tile plus the kernel reach plus a 3-pixel offset margin, clipped to the image, in fp64 cells —
// ds_add_f64 is the fast LDS atomic on gfx950 (6.8 lanes/clk/CU against 0.33 for ds_add_f32, profiles/archive/r01e) and makes the sum
// order-independent.  Corners beyond the window (|offset| > 3 pixels outside the reach) go straight to global atomics; the window is
// flushed with one fp32 atomic per non-zero (cell, channel).
constexpr int GX2_CS = 4;        // channels per slice
constexpr int GX2_MARGIN = 3;    // offset margin of the window beyond the kernel reach

template <typename T>   // T: storage of grad_out `g`
__global__ __launch_bounds__(256) void cl_ddw2d_gx_kernel(Ddw2dArgs p, int TH, int TW, int ntx, int nty, int reach_y, int reach_x)
{
    const T *gin = reinterpret_cast<const T *>(p.g);
    DLKA_DYN_SMEM(unsigned char, smem);
    const int tid = threadIdx.x, lane = tid & 63;
    int t = blockIdx.x;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty; const int b = t / nty;
    const int slice = blockIdx.y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int wy0 = max(0, oy0 - reach_y - GX2_MARGIN), wy1 = min(p.H, oy0 + TH + reach_y + GX2_MARGIN);
    const int wx0 = max(0, ox0 - reach_x - GX2_MARGIN), wx1 = min(p.W, ox0 + TW + reach_x + GX2_MARGIN);
    const int WH = wy1 - wy0, WW = wx1 - wx0, wcells = WH * WW;
    const int wstride = wcells + 64;          // per channel plane: the window, then one trash cell per lane
    double *Win = reinterpret_cast<double *>(smem);                                  // [CS][wstride]
    float *Ws = reinterpret_cast<float *>(smem + (size_t)GX2_CS * wstride * 8);      // [K][CS] tap weights of this slice
    for (int e = tid; e < GX2_CS * wstride; e += 256) Win[e] = 0.0;
    for (int e = tid; e < p.K * GX2_CS; e += 256) Ws[e] = p.wp[(long)(e / GX2_CS) * p.C + slice * GX2_CS + (e % GX2_CS)];
    __syncthreads();
    const int npx = TH * TW;
    const int trash = wcells + lane;
    for (int px = tid; px < npx; px += 256) {
        const int oy = oy0 + px / TW, ox = ox0 + px % TW;
        if (oy >= p.H || ox >= p.W) continue;
        const int n = oy * p.W + ox;
        const f32x4 g4 = act_load4(gin, ((long)b * p.N + n) * p.C + slice * GX2_CS);
        const float *offp = p.off + (long)b * 2 * p.K * p.N + n;
        for (int tap = 0; tap < p.K; ++tap) {
            const int ti = tap / p.kw, tj = tap - ti * p.kw;
            int y0, x0;
            float ly, lx;
            bool reach;
            if (!sample_cell2(offp[(long)(2 * tap) * p.N], offp[(long)(2 * tap + 1) * p.N], oy - p.ph + ti * p.dh, ox - p.pw + tj * p.dw, p.H, p.W, y0, x0, ly, lx,
                              reach))
                continue;   // the sample's guard (the one sampling rule, deform_sample.h)
            const float wy[2] = {1.f - ly, ly}, wx[2] = {1.f - lx, lx};
            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(Ws + tap * GX2_CS);
            const float col[4] = {g4[0] * w4[0], g4[1] * w4[1], g4[2] * w4[2], g4[3] * w4[3]};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int yy = y0 + (q >> 1), xx = x0 + (q & 1);
                const bool inimg = ((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W);
                const bool inwin = inimg & (yy >= wy0) & (yy < wy1) & (xx >= wx0) & (xx < wx1);
                const float wq = wy[q >> 1] * wx[q & 1];
                // branch-free window part: a corner outside the window adds 0.0 to this lane's trash cell
                const int idx = inwin ? (yy - wy0) * WW + (xx - wx0) : trash;
                const float wv = inwin ? wq : 0.f;
#pragma unroll
                for (int c = 0; c < GX2_CS; ++c) atomicAdd(Win + c * wstride + idx, (double)(col[c] * wv));
                if (inimg & !inwin) {   // rare: beyond the offset margin
                    float *dst = p.gx + ((long)b * p.N + yy * p.W + xx) * p.C + slice * GX2_CS;
#pragma unroll
                    for (int c = 0; c < GX2_CS; ++c) atomicAdd(dst + c, col[c] * wq);
                }
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < wcells; e += 256) {
        const int yy = wy0 + e / WW, xx = wx0 + e % WW;
        float *dst = p.gx + ((long)b * p.N + yy * p.W + xx) * p.C + slice * GX2_CS;
#pragma unroll
        for (int c = 0; c < GX2_CS; ++c) {
            const float v = (float)Win[c * wstride + e];
            if (v != 0.f) atomicAdd(dst + c, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// grad_input, second generation (round 4): INPUT-pixel tiles, lane = channel, no atomics.
// The window kernel above gives a workgroup a tile of OUTPUT pixels and a 4-channel slice: its window has to hold the tile plus the kernel
// reach (9 pixels for the 7x7 dilation-3 conv: at 56^2 the window IS the image), the sampling description of every (pixel, tap) is recomputed by
// each of the C / 4 slices, and every corner costs 4 ds_add_f64 — 1.18 ms at (96, 56^2, B = 24), 1.3 % of the vector roof.
// Depthwise = no contraction over channels, and all channels of a pixel share one sampling position.  So here
//   * a wave owns an 8 x 4 tile of INPUT pixels and 128 channels (lane = a channel pair): its window is the tile itself, [cell][channel],
//     16 KB of fp32 — the halo is in the ENUMERATION, not in LDS: per tap, the samples that can touch the tile are those whose base position
//     lies within the tile +- (margin + 1), 15 x 11 candidates;
//   * description with lane = candidate (64 at once: offsets, the one sampling rule sample_cell2, guard, per-corner validity against image AND
//     tile), then __ballot packs the hits and the wave walks them with lane = CHANNEL: one coalesced row read of grad_out, the tap weight, and up
//     to four read-add-write updates of window cells that only this wave touches and that lie in 64 consecutive banks — exact fp32, fixed order
//     (deterministic), no atomics of any kind, every element of grad_input written exactly once by plain stores (no zero fill needed);
//   * samples with |offset| > margin on either axis ("far", ~0.5 % at 1-pixel offsets) are left out here — a tile cannot know about them without
//     scanning the whole image — and added by cl_ddw2d_gx_far_kernel afterwards (thread = (pixel, tap), global fp32 atomics, rare).
// The split near / far is decided on the offset VALUES alone, so every tile that enumerates a sample takes the same decision.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef DLKA_GX3_TY
#define DLKA_GX3_TY 4   // (-D... for scripts/build_variant.sh).  Round 6: 4 x 4 instead of 8 x 4 — a 6 x 6 window is 18 KB per wave, EIGHT waves per CU instead of five, and one wave
#define DLKA_GX3_TX 4   // issues a vector instruction only every ~5 clocks (scripts/ubench/fma_rate.hip): 906 -> 767 us (7x7) / 507 -> 446 (5x5) at (96, 56^2, B = 24) although a tile
#endif                  // now walks 1.56 hits per pixel instead of 1.41; 4x3 465 / 781, 3x4 470, 4x2 498, 2x4 499, 6x4 498, 2x2 588, 8x2 484 / 849, 4x8 491 / 888, 16x2 648 / 1140
#ifndef DLKA_GX3_MIN_WAVES
#define DLKA_GX3_MIN_WAVES 1024   // launches with fewer waves keep the first-generation window kernel (with 4 x 4 tiles — us, window kernel / this one: (192, 28^2) 383 / 260 (5x5), 588 / 427 (7x7);
#endif                            // (384, 14^2) 173 / 164, 210 / 194: profiles/r10_notes.md)
constexpr int GX3_TY = DLKA_GX3_TY, GX3_TX = DLKA_GX3_TX;   // tile (input pixels)
constexpr int GX3_MG = 3;                             // offset margin: |dy|, |dx| <= GX3_MG are "near"
constexpr int GX3_NBY = GX3_TY + 2 * GX3_MG + 1, GX3_NBX = GX3_TX + 2 * GX3_MG + 1;   // candidate base positions per axis
constexpr int GX3_CW = 128;                           // channels per wave: lane = a PAIR of channels (8-byte LDS and grad_out accesses)
constexpr int GX3_NH = 8;                             // hits per group = grad_out row requests in flight per register set
// Round 5 — the window carries a one-cell RING around the tile: (TY + 2) x (TX + 2) cells of [lane = channel pair] float2 (30 KB per wave = 5 waves per CU with round 5's 8 x 4 tile; 18 KB = 8 waves with 4 x 4).  A hit's
// 2 x 2 footprint then ALWAYS lies inside the window (a hit has at least one corner in the tile, so its low corner is at most one cell outside), its four cells
// are ONE base address plus the compile-time offsets {0, 1, WX, WX + 1} (ds_read_b64 / ds_write_b64 with immediate offsets), and corners outside the tile simply
// land in ring cells that are never stored — no per-corner select, no per-corner address.  Round 4's version kept a tile-sized window (16 KB, 9 waves per CU) and
// redirected each corner it did not own to a trash cell: ~48 scalar and ~35 vector instructions per hit (ISA loop mix, scripts/isa_loop_mix.py), 110 wave
// instructions per hit in the PMC counters, 748 us at (96, 56^2, B = 24).  Now a hit is: one readlane of the row index, one of the window base, four of the
// corner weights, a packed multiply by the tap weight, four 8-byte LDS reads, four packed FMAs, four 8-byte LDS writes.
constexpr int GX3_WX = GX3_TX + 2, GX3_WY = GX3_TY + 2;
constexpr int GX3_NCELL = GX3_WX * GX3_WY;

__device__ __forceinline__ bool gx3_near(float oy, float ox) { return (fabsf(oy) <= (float)GX3_MG) && (fabsf(ox) <= (float)GX3_MG); }

typedef float gx3_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gx3_f2 gx3_load2(const float *p, long i) { const float2 v = *reinterpret_cast<const float2 *>(p + i); return gx3_f2{v.x, v.y}; }
__device__ __forceinline__ gx3_f2 gx3_load2(const bf16_t *p, long i)
{
    const unsigned w = *reinterpret_cast<const unsigned *>(p + i);   // two bf16: element i in the low half
    return gx3_f2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
}

template <typename T>   // T: storage of grad_out `g`
__global__ __launch_bounds__(64) void cl_ddw2d_gx3_kernel(Ddw2dArgs p, int ntx, int nty, int xcd_nx)
{
    __shared__ __attribute__((aligned(8))) gx3_f2 Win[GX3_NCELL * 64];   // [window cell][lane = channel pair]
    const T *gin = reinterpret_cast<const T *>(p.g);
    const int lane = threadIdx.x;
    int t = DLKA_XCD_BX(xcd_nx);   // an XCD owns a contiguous range of tiles (whole images): the grad_out rows its tiles re-read stay in its L2
    if (t < 0) return;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty; const int b = t / nty;
    const int c = blockIdx.y * GX3_CW + 2 * lane;
    const bool cok = c < p.C;
    const int cc = cok ? c : 0;   // (lanes beyond C run the same instruction stream on channels 0, 1 and store nothing)
    const int ty0 = ty * GX3_TY, tx0 = tx * GX3_TX;
#pragma unroll
    for (int e = 0; e < GX3_NCELL; ++e) Win[e * 64 + lane] = gx3_f2{0.f, 0.f};   // (each lane only ever touches its own column: no barrier anywhere)
    constexpr int NCT = GX3_NBY * GX3_NBX;   // candidates per tap
    const int ncand = p.K * NCT;
    // The candidates of batch k + 1 are decoded and their two offsets REQUESTED before the hits of batch k are walked (a batch otherwise begins with an
    // exposed L2 round trip that nothing hides).  Decoded state of the batch in flight: by / bx / n (n < 0: no candidate).
    int nby = 0, nbx = 0, nn = -1, ntap = 0;
    float nfy = 0.f, nfx = 0.f;
    gx3_f2 nwA = {0.f, 0.f}, nwB = {0.f, 0.f};
    auto decode_and_request = [&](int base) {
        const int tapA = base / NCT;   // the (at most two) taps of a batch — candidates are tap-major — and their weights for this lane's channels
        const int tapB = min(tapA + 1, p.K - 1);
        nwA = gx3_load2(p.wp, (long)min(tapA, p.K - 1) * p.C + cc);
        nwB = gx3_load2(p.wp, (long)tapB * p.C + cc);
        const int id = base + lane;
        nn = -1;
        ntap = tapA;
        nfy = nfx = 0.f;
        if (id < ncand) {
            const int tap = id / NCT;
            const int r = id - tap * NCT;
            const int cy = r / GX3_NBX, cx = r - cy * GX3_NBX;
            const int ti = tap / p.kw, tj = tap - ti * p.kw;
            nby = ty0 - GX3_MG - 1 + cy; nbx = tx0 - GX3_MG - 1 + cx;                   // base = output pixel - pad + tap * dilation
            const int oy = nby + p.ph - ti * p.dh, ox = nbx + p.pw - tj * p.dw;         // the output pixel this (tap, base) belongs to
            ntap = tap;
            if (((unsigned)oy < (unsigned)p.H) & ((unsigned)ox < (unsigned)p.W)) {
                nn = oy * p.W + ox;
                const float *offp = p.off + ((long)b * 2 * p.K + 2 * tap) * p.N + nn;
                nfy = offp[0]; nfx = offp[p.N];
            }
        }
    };
    decode_and_request(0);
    for (int base = 0; base < ncand; base += 64) {
        // ---- finish the description of this batch's 64 candidates: lane = (tap, base position) ----
        const int tapA = base / NCT;
        const gx3_f2 wA = nwA, wB = nwB;
        int m = 0, key = 0;        // key = (window cell of the low corner) << 1 | (second tap of the batch)
        bool hit = false;
        float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;   // corner weights; 0 for corners outside the IMAGE (those inside it but outside the tile go to ring cells)
        if (nn >= 0) {
            const float fy = nfy, fx = nfx;
            int y0, x0;
            float ly, lx;
            bool reach;
            const bool inside = sample_cell2(fy, fx, nby, nbx, p.H, p.W, y0, x0, ly, lx, reach);   // the one sampling rule (deform_sample.h)
            if (inside & gx3_near(fy, fx)) {
                const int ry = y0 - ty0, rx = x0 - tx0;   // low corner relative to the tile: a corner is in the tile iff it is in [0, TY) x [0, TX)
                const bool iy0 = y0 >= 0, iy1 = y0 + 1 <= p.H - 1, ix0 = x0 >= 0, ix1 = x0 + 1 <= p.W - 1;   // ... and counts iff it is in the image
                const bool ty_0 = (ry >= 0) & (ry < GX3_TY), ty_1 = (ry + 1 >= 0) & (ry + 1 < GX3_TY);
                const bool tx_0 = (rx >= 0) & (rx < GX3_TX), tx_1 = (rx + 1 >= 0) & (rx + 1 < GX3_TX);
                hit = (iy0 & ix0 & ty_0 & tx_0) | (iy0 & ix1 & ty_0 & tx_1) | (iy1 & ix0 & ty_1 & tx_0) | (iy1 & ix1 & ty_1 & tx_1);
                if (hit) {   // => ry in [-1, TY - 1], rx in [-1, TX - 1]: the 2 x 2 footprint lies in the ringed window
                    const float hy = 1.f - ly, hx = 1.f - lx;
                    w00 = (iy0 & ix0) ? hy * hx : 0.f; w01 = (iy0 & ix1) ? hy * lx : 0.f; w10 = (iy1 & ix0) ? ly * hx : 0.f; w11 = (iy1 & ix1) ? ly * lx : 0.f;
                    key = (((ry + 1) * GX3_WX + (rx + 1)) << 1) | (ntap != tapA ? 1 : 0);
                    m = b * p.N + nn;
                }
            }
        }
        if (base + 64 < ncand) decode_and_request(base + 64);   // in flight while this batch's hits are applied
        // ---- walk the hits with lane = channel pair, GX3_NH at a time: the grad_out rows of the NEXT group are requested before the current group is
        //      applied (two fixed register sets: no value has to be moved — and therefore waited for — while its load is in flight).  A slot past the last hit
        //      takes a lane that is NOT a hit: its weights are 0, its key 0 (window cell 0, a ring corner) and its row index 0, so the slot runs the same
        //      straight-line code and changes nothing.  Such a lane exists whenever a slot needs one: 64 hits fill GX3_NH | 64 slots exactly. ----
        unsigned long long mask = __ballot(hit);
        const int nz = (~mask) ? __builtin_ctzll(~mask) : 0;
        int la[GX3_NH], lb[GX3_NH];
        gx3_f2 ga[GX3_NH], gb[GX3_NH];
        bool more;
#define DLKA_GX3_LOAD(G, L)                                                                  \
        more = mask != 0;                                                                    \
        _Pragma("unroll") for (int q = 0; q < GX3_NH; ++q) {                                 \
            L[q] = mask ? __builtin_ctzll(mask) : nz;                                        \
            mask &= mask - 1;                                                                \
            G[q] = gx3_load2(gin, (long)lane_bcast(m, L[q]) * p.C + cc);                     \
        }
#define DLKA_GX3_APPLY(G, L)                                                                 \
        _Pragma("unroll") for (int q = 0; q < GX3_NH; ++q) {                                 \
            const int skey = lane_bcast(key, L[q]);                                          \
            const gx3_f2 col = G[q] * ((skey & 1) ? wB : wA);   /* d loss / d sample */      \
            const float s00 = lane_bcast(w00, L[q]), s01 = lane_bcast(w01, L[q]), s10 = lane_bcast(w10, L[q]), s11 = lane_bcast(w11, L[q]);  \
            gx3_f2 *wp_ = Win + ((skey >> 1) * 64 + lane);                                   \
            const gx3_f2 v0 = wp_[0], v1 = wp_[64], v2 = wp_[GX3_WX * 64], v3 = wp_[(GX3_WX + 1) * 64];      \
            wp_[0] = __builtin_elementwise_fma(gx3_f2{s00, s00}, col, v0);                   \
            wp_[64] = __builtin_elementwise_fma(gx3_f2{s01, s01}, col, v1);                  \
            wp_[GX3_WX * 64] = __builtin_elementwise_fma(gx3_f2{s10, s10}, col, v2);         \
            wp_[(GX3_WX + 1) * 64] = __builtin_elementwise_fma(gx3_f2{s11, s11}, col, v3);   \
        }
        DLKA_GX3_LOAD(ga, la)
        while (more) {
            DLKA_GX3_LOAD(gb, lb)
            DLKA_GX3_APPLY(ga, la)
            if (!more) break;
            DLKA_GX3_LOAD(ga, la)
            DLKA_GX3_APPLY(gb, lb)
        }
#undef DLKA_GX3_LOAD
#undef DLKA_GX3_APPLY
    }
    if (!cok) return;
#pragma unroll 4
    for (int e = 0; e < GX3_TY * GX3_TX; ++e) {
        const int yy = ty0 + e / GX3_TX, xx = tx0 + e % GX3_TX;
        if (yy < p.H && xx < p.W) {
            const gx3_f2 v = Win[((e / GX3_TX + 1) * GX3_WX + (e % GX3_TX + 1)) * 64 + lane];
            *reinterpret_cast<float2 *>(p.gx + ((long)b * p.N + yy * p.W + xx) * p.C + c) = make_float2(v.x, v.y);
        }
    }
}

// the far samples (|offset| > GX3_MG on either axis) of the kernel above: lane = (output pixel, tap) for the test, then the wave walks the far ones
// it found with lane = channel — coalesced global fp32 atomics, four corner rows per sample.
template <typename T>
__global__ __launch_bounds__(256) void cl_ddw2d_gx_far_kernel(Ddw2dArgs p)
{
    const T *gin = reinterpret_cast<const T *>(p.g);
    const int lane = threadIdx.x & 63;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;   // [b][tap][n]: consecutive threads read consecutive offsets
    int n = 0, tap = 0, b = 0, y0 = 0, x0 = 0;
    float ly = 0.f, lx = 0.f;
    bool far = false;
    if (id < (long)p.M * p.K) {
        n = (int)(id % p.N);
        tap = (int)((id / p.N) % p.K);
        b = (int)(id / ((long)p.N * p.K));
        const float *offp = p.off + ((long)b * 2 * p.K + 2 * tap) * p.N + n;
        const float fy = offp[0], fx = offp[p.N];
        if (!gx3_near(fy, fx)) {
            const int oy = n / p.W, ox = n - oy * p.W;
            const int ti = tap / p.kw, tj = tap - ti * p.kw;
            bool reach;
            far = sample_cell2(fy, fx, oy - p.ph + ti * p.dh, ox - p.pw + tj * p.dw, p.H, p.W, y0, x0, ly, lx, reach);
        }
    }
    unsigned long long mask = __ballot(far);
    while (mask) {
        const int l = __builtin_ctzll(mask);
        mask &= mask - 1;
        const int sn = lane_bcast(n, l), stap = lane_bcast(tap, l), sb = lane_bcast(b, l), sy0 = lane_bcast(y0, l), sx0 = lane_bcast(x0, l);
        const float sly = lane_bcast(ly, l), slx = lane_bcast(lx, l);
        const float wy[2] = {1.f - sly, sly}, wx[2] = {1.f - slx, slx};
        const long grow = ((long)sb * p.N + sn) * p.C;
        for (int c = lane; c < p.C; c += 64) {
            const float col = act_load1(gin, grow + c) * p.wp[(long)stap * p.C + c];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int yy = sy0 + (q >> 1), xx = sx0 + (q & 1);
                if (((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W))   // uniform
                    atomicAdd(p.gx + ((long)sb * p.N + yy * p.W + xx) * p.C + c, col * (wy[q >> 1] * wx[q & 1]));
            }
        }
    }
}

// gw[c][tap] (reference layout [C][1][kh][kw]) = sum_blocks part[block][tap][c]
// A workgroup folds 32 consecutive (tap, c) outputs: thread (e = tid & 31, slice = tid >> 5) sums the partial blocks slice, slice + 8, ... with four loads in flight, the
// eight slice sums meet in LDS (fixed order: deterministic).  Round 5 — the first version gave ONE thread all of an output's partials (784 of them at (96, 56^2, B = 24)):
// a chain of dependent-latency loads on 19 workgroups, 110 - 118 us per launch for 14.7 MB (scripts/time_ddw2d_gx.py), 1.0 ms of the 2-D step's 17.6.
__global__ __launch_bounds__(256) void cl_ddw2d_fold_kernel(const float *__restrict__ part, float *__restrict__ gw, int nblocks, int K, int C)
{
    __shared__ float red[8][33];
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;   // (tap, c)
    const long stride = (long)K * C;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < K * C) {
        const float *src = part + e;
        int bk = sl;
        for (; bk + 24 < nblocks; bk += 32) {
            a0 += src[(long)bk * stride]; a1 += src[(long)(bk + 8) * stride]; a2 += src[(long)(bk + 16) * stride]; a3 += src[(long)(bk + 24) * stride];
        }
        for (; bk < nblocks; bk += 8) a0 += src[(long)bk * stride];
    }
    red[sl][el] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0 && e < K * C) {
        const float t = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
        const int tap = e / C, c = e - tap * C;
        gw[(long)c * K + tap] = t;
    }
}

static int ddw2d_nch(int C)
{
    const int n = C / 32;
    return (C % 32 == 0 && (n == 1 || n == 2 || n == 3 || n == 4 || n == 6 || n == 8 || n == 12)) ? n : 0;
}

int cl_ddw2d_supported(int C) { return ddw2d_nch(C) != 0; }

int cl_ddw2d_bwd_px_per_block(int M)
{
    // ~1024 workgroups when the tensor is large enough; at least one 32-pixel pass per wave group
    int px = ((M / 1024) + 31) / 32 * 32;
    if (px < 32) px = 32;
    return px;
}

size_t cl_ddw2d_part_floats(int M, int K, int C) { return (size_t)cdiv(M, cl_ddw2d_bwd_px_per_block(M)) * K * C; }

static void fill_ddw(Ddw2dArgs &a, const DwArgs2d &d)
{
    memset(&a, 0, sizeof(a));
    a.in = d.in; a.off = d.off; a.wp = d.wp; a.g = d.g; a.out = d.out; a.out_lo = d.out_lo; a.gx = d.gx; a.goff = d.goff; a.part = d.part;
    a.B = d.B; a.H = d.H; a.W = d.W; a.N = d.H * d.W; a.M = d.B * d.H * d.W; a.C = d.C; a.K = d.kh * d.kw; a.kh = d.kh; a.kw = d.kw;
    a.ph = d.ph; a.pw = d.pw; a.dh = d.dh; a.dw = d.dw;
}

int launch_cl_ddw2d_fwd(const DwArgs2d &d, hipStream_t st)
{
    const int nch = ddw2d_nch(d.C);
    if (!nch) return DLKA_ERR_UNSUPPORTED;
    Ddw2dArgs a;
    fill_ddw(a, d);
    if ((long)a.M * a.C * 4 >= (1l << 31)) return DLKA_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    if (d.act_bf16 || (!d.out && !d.out_lo)) return DLKA_ERR_UNSUPPORTED;  // (fp32 input only: the DLKA_BF16 block feeds it its fp32 chain tensors)
    const int mblocks = cdiv(a.M, 32);
    // few pixels (14^2 stages): split the channel chunks over gridDim.y so that the chip still fills
    int per = nch;
    while (per > 1 && mblocks * (nch / per) < 1024 && per % 2 == 0) per /= 2;
    if (per == 3 && mblocks * (nch / 3) < 512) per = 1;
    dim3 grid(mblocks, nch / per), block(256);
    if (xcd_swizzle_enabled() && mblocks >= xcd_min_blocks()) { a.xcd_nx = mblocks; grid.x = xcd_grid(mblocks); }
#define DLKA_DDW_F(N_) case N_: { auto k = cl_ddw2d_fwd_kernel<N_>; DLKA_LAUNCH(k, grid, block, 0, st, a); } break;
    switch (per) {
        DLKA_DDW_F(1) DLKA_DDW_F(2) DLKA_DDW_F(3) DLKA_DDW_F(4) DLKA_DDW_F(6) DLKA_DDW_F(8) DLKA_DDW_F(12)
        default: return DLKA_ERR_UNSUPPORTED;
    }
#undef DLKA_DDW_F
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// gx must be zero-filled by the caller; gw receives the folded weight gradient in the reference layout
// gx_st: stream of the grad_input kernels (null = st).  They are independent of the grad_offset / weight-gradient kernel: a caller that has forked gx_st behind st's
// producers (and joins it in front of grad_input's consumer) runs the two beside each other (lka2d_cl_backward).
int launch_cl_ddw2d_bwd(const DwArgs2d &d, float *gw, hipStream_t st, hipStream_t gx_st)
{
    hipStream_t gst = gx_st ? gx_st : st;
    const int nch = ddw2d_nch(d.C);
    if (!nch) return DLKA_ERR_UNSUPPORTED;
    Ddw2dArgs a;
    fill_ddw(a, d);
    if ((long)a.M * a.C * 4 >= (1l << 31)) return DLKA_ERR_UNSUPPORTED;
    a.px_per_block = cl_ddw2d_bwd_px_per_block(a.M);
    const int nblocks = cdiv(a.M, a.px_per_block);
    dim3 grid(nblocks), block(256);
    {   // few pixel blocks (14 x 14: 147; 28 x 28: 588): split the taps over gridDim.y until ~1024 workgroups exist
        int groups = nblocks >= 1024 ? 1 : cdiv(1024, nblocks);
        if (groups > a.K) groups = a.K;
        a.tpg = cdiv(a.K, groups);
        grid.y = cdiv(a.K, a.tpg);
    }
    if (xcd_swizzle_enabled() && nblocks >= xcd_min_blocks()) { a.xcd_nx = nblocks; grid.x = xcd_grid(nblocks); }
#define DLKA_DDW_B(N_) case N_: { if (d.act_bf16) { auto k = cl_ddw2d_bwd_kernel<N_, bf16_t>; DLKA_LAUNCH(k, grid, block, 0, st, a); }   \
                                  else { auto k = cl_ddw2d_bwd_kernel<N_, float>; DLKA_LAUNCH(k, grid, block, 0, st, a); } } break;
    switch (nch) {
        DLKA_DDW_B(1) DLKA_DDW_B(2) DLKA_DDW_B(3) DLKA_DDW_B(4) DLKA_DDW_B(6) DLKA_DDW_B(8) DLKA_DDW_B(12)
        default: return DLKA_ERR_UNSUPPORTED;
    }
#undef DLKA_DDW_B
    DLKA_CHECK_LAUNCH();
    DLKA_LAUNCH(cl_ddw2d_fold_kernel, dim3(cdiv(a.K * a.C, 32)), dim3(256), 0, st, (const float *)a.part, gw, nblocks, a.K, a.C);
    DLKA_CHECK_LAUNCH();
    // Second generation where the image gives it enough tiles to fill the chip (one wave per (tile, 128 channels): 2352 waves at (96, 56^2, B = 24));
    // the smaller decoder shapes keep the window kernel (measured, profiles/r05_notes.md: 575 vs 478 us at (192, 28^2), 346 vs ~300 at (384, 14^2)).
    // DLKA_DDW2D_GX=window | tiles forces one of them (A/B runs, parity tests of both).
    const char *gxsel = getenv("DLKA_DDW2D_GX");
    const long gx3_waves = (long)a.B * cdiv(a.H, GX3_TY) * cdiv(a.W, GX3_TX) * cdiv(a.C, GX3_CW);
    const bool use_gx3 = (a.C & 1) == 0 && (gxsel ? gxsel[0] == 't' : gx3_waves >= DLKA_GX3_MIN_WAVES);
    if (use_gx3) {   // grad_input, second generation: input-pixel tiles, lane = channel pair (the first generation stays for A/B runs)
        const int ntx = cdiv(a.W, GX3_TX), nty = cdiv(a.H, GX3_TY);
        const int ntiles = a.B * nty * ntx;
        dim3 ggrid(ntiles, cdiv(a.C, GX3_CW));
        int xcd_nx = 0;
        if (xcd_swizzle_enabled() && ntiles >= xcd_min_blocks()) { xcd_nx = ntiles; ggrid.x = xcd_grid(ntiles); }
        const dim3 fgrid((unsigned)(((long)a.M * a.K + 255) / 256));
        if (d.act_bf16) {
            auto k = cl_ddw2d_gx3_kernel<bf16_t>; DLKA_LAUNCH(k, ggrid, dim3(64), 0, gst, a, ntx, nty, xcd_nx);
            auto f = cl_ddw2d_gx_far_kernel<bf16_t>; DLKA_LAUNCH(f, fgrid, dim3(256), 0, gst, a);
        } else {
            auto k = cl_ddw2d_gx3_kernel<float>; DLKA_LAUNCH(k, ggrid, dim3(64), 0, gst, a, ntx, nty, xcd_nx);
            auto f = cl_ddw2d_gx_far_kernel<float>; DLKA_LAUNCH(f, fgrid, dim3(256), 0, gst, a);
        }
        DLKA_CHECK_LAUNCH();
        return DLKA_OK;
    }
    {   // grad_input: LDS-window scatter
        const int reach_y = a.ph > (a.kh - 1) * a.dh - a.ph ? a.ph : (a.kh - 1) * a.dh - a.ph;   // |base - output pixel| <= reach
        const int reach_x = a.pw > (a.kw - 1) * a.dw - a.pw ? a.pw : (a.kw - 1) * a.dw - a.pw;
        int TH = a.H < 16 ? a.H : 16, TW = a.W < 32 ? a.W : 32;
        auto cells = [&](int th, int tw) {
            const int wh = th + 2 * (reach_y + GX2_MARGIN) < a.H ? th + 2 * (reach_y + GX2_MARGIN) : a.H;
            const int ww = tw + 2 * (reach_x + GX2_MARGIN) < a.W ? tw + 2 * (reach_x + GX2_MARGIN) : a.W;
            return (size_t)wh * ww;
        };
        auto lds_bytes = [&](int th, int tw) { return (size_t)GX2_CS * (cells(th, tw) + 64) * 8 + (size_t)a.K * GX2_CS * 4; };
        // enough workgroups for the chip, and two of them per CU (<= 76 KB each)
        while ((lds_bytes(TH, TW) > 76 * 1024 || (long)a.B * cdiv(a.H, TH) * cdiv(a.W, TW) * (a.C / GX2_CS) < 1024) && TH * TW > 64) {
            if (TW >= 2 * TH && TW > 8) TW = cdiv(TW, 2); else TH = cdiv(TH, 2);
        }
        const size_t lds = lds_bytes(TH, TW);
        if (lds > 150 * 1024) return DLKA_ERR_UNSUPPORTED;
#if !defined(HIPEMU)
        static std::atomic<uint64_t> attr_done{0};   // dynamic LDS above 64 KB: per function and per device
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return DLKA_ERR_LAUNCH;
        const uint64_t bit = 1ull << (dev & 63);
        if (!(attr_done.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(cl_ddw2d_gx_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(cl_ddw2d_gx_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return DLKA_ERR_LAUNCH;
            attr_done.fetch_or(bit, std::memory_order_release);
        }
#endif
        const int ntx = cdiv(a.W, TW), nty = cdiv(a.H, TH);
        dim3 ggrid(a.B * nty * ntx, a.C / GX2_CS);
        if (d.act_bf16) { auto k = cl_ddw2d_gx_kernel<bf16_t>; DLKA_LAUNCH(k, ggrid, dim3(256), lds, gst, a, TH, TW, ntx, nty, reach_y, reach_x); }
        else { auto k = cl_ddw2d_gx_kernel<float>; DLKA_LAUNCH(k, ggrid, dim3(256), lds, gst, a, TH, TW, ntx, nty, reach_y, reach_x); }
        DLKA_CHECK_LAUNCH();
    }
    return DLKA_OK;
}

}  // namespace dlka
