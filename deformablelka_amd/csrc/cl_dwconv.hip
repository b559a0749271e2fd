// Channels-last depthwise 3-D convolution (the large-kernel half of D-LKA): 5^3 pad 2 and 7^3 dilation 3 pad 9
// (3D/d_lka_former/network_architecture/synapse/transformerblock.py:637-638; cuDNN in the reference).
//
// Depthwise = no contraction over channels, so this is a register-tiled vector kernel, not a GEMM: lanes run over
// channels (every load / store is a contiguous 128-byte-per-32-lanes row piece of the [voxel][C] layout), each
// work-item owns TW consecutive outputs along W and slides the KW taps over one input row segment held in registers
// (TW + (KW-1)*DIL loads feed TW*KW FMAs).  Same kernel = forward and data gradient (flipped taps, no bias).
// (A row-tiled variant — TH output rows spaced DIL apart per work-item, 5 FMA per load instead of 2.15 — was measured SLOWER
// (72.9 vs 65.5 us at C=32 / 32^3, profiles/archive/r01o_dw_variants.txt): the kernel is bound by instruction issue with too few waves to
// hide latency, not by the L1 request rate, and bigger per-thread tiles leave ~1 wave per SIMD.)
#include <stdlib.h>

#include <atomic>

#include "cl_args.h"
#include "dlka_kernels.h"

namespace dlka {

// T: activation storage (float, or bf16_t with fp32 arithmetic — DLKA_BF16 token path); weights / bias are fp32
template <typename T, int KW, int DIL, int TW, int ABL = 0>   // ABL: ablation for profiling only (1 = no input loads, 2 = no FMAs)
__global__ __launch_bounds__(256) void cl_dwconv_kernel(DwArgs p)
{
    constexpr int SEG = TW + (KW - 1) * DIL;
    const T *inp = reinterpret_cast<const T *>(p.in), *gxp = reinterpret_cast<const T *>(p.gelu_x), *gap = reinterpret_cast<const T *>(p.gelu_add);
    T *outp = reinterpret_cast<T *>(p.out);
    const int cpb = p.C < 256 ? p.C : 256;           // channels per block
    const int rpb = 256 / cpb;                       // W-runs per block
    const int c = blockIdx.z * cpb + threadIdx.x % cpb;
    const int run = blockIdx.x * rpb + threadIdx.x / cpb;
    const int runs_per_row = cdiv(p.W, TW);
    const int rows = p.B * p.D * p.H;
    if (run >= rows * runs_per_row || c >= p.C) return;
    const int w0 = (run % runs_per_row) * TW;
    const int row = run / runs_per_row;
    const int h0 = row % p.H, d0 = (row / p.H) % p.D, b = row / (p.H * p.D);

    float acc[TW];
    const float bv = p.bias ? p.bias[c] : 0.f;
#pragma unroll
    for (int t = 0; t < TW; ++t) acc[t] = bv;

    for (int i = 0; i < p.kd; ++i) {
        const int zd = d0 + i * p.dd - p.pd;
        if (zd < 0 || zd >= p.D) continue;
        for (int j = 0; j < p.kh; ++j) {
            const int zh = h0 + j * p.dh - p.ph;
            if (zh < 0 || zh >= p.H) continue;
            const T *rowp = inp + (((long)(b * p.D + zd) * p.H + zh) * p.W) * p.C + c;
            float seg[SEG];
#pragma unroll
            for (int e = 0; e < SEG; ++e) {
                const int zw = w0 - p.pw + e;
                seg[e] = (ABL == 1) ? (float)(e + zh) : ((zw >= 0 && zw < p.W) ? act_load1(rowp, (long)zw * p.C) : 0.f);
            }
            const float *wrow = p.wp + (long)((i * p.kh + j) * KW) * p.C + c;
            if (ABL == 2) {
#pragma unroll
                for (int e = 0; e < SEG; ++e) acc[e % TW] += seg[e];
#pragma unroll
                for (int k = 0; k < KW; ++k) acc[k % TW] += wrow[(long)k * p.C];
                continue;
            }
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const float wv = wrow[(long)k * p.C];
#pragma unroll
                for (int t = 0; t < TW; ++t) acc[t] = fmaf(wv, seg[t + k * DIL], acc[t]);
            }
        }
    }
    const long obase = (((long)(b * p.D + d0) * p.H + h0) * p.W + w0) * p.C + c;
    if (p.gelu_x) {   // uniform: a = GELU(h) feeds dw 5^3 and the gate, so gh = (ga_gate + ga_dw5) * gelu'(h) closes here
#pragma unroll
        for (int t = 0; t < TW; ++t)
            if (w0 + t < p.W) act_store1(outp, obase + (long)t * p.C, (acc[t] + act_load1(gap, obase + (long)t * p.C)) * dgelu_f(act_load1(gxp, obase + (long)t * p.C)));
        return;
    }
#pragma unroll
    for (int t = 0; t < TW; ++t)
        if (w0 + t < p.W) act_store1(outp, obase + (long)t * p.C, acc[t]);
    if (sizeof(T) == 4 && p.out_lo) {   // uniform: the bf16 copy the gathers / the backward pass read (mixed-precision DLKA_BF16 block)
        bf16_t *lo = reinterpret_cast<bf16_t *>(p.out_lo);
#pragma unroll
        for (int t = 0; t < TW; ++t)
            if (w0 + t < p.W) act_store1(lo, obase + (long)t * p.C, acc[t]);
    }
}

// Second generation of the forward / data-gradient kernel.  Same tiling; what changes is how the input row segment is read.  The first
// version's `(zw >= 0 && zw < W) ? row[zw * C] : 0` compiled to a branch, a 64-bit address computation and a wait PER LOAD (~14
// instructions for each of the 26 loads of a 7-tap dil-3 row, against 56 FMAs).  Here the (b, d, h) of a wave's outputs is wave-uniform
// (the launcher checks it), so every input row gets its own buffer descriptor built by the scalar unit: base = the row, size = W*C*4
// bytes.  The per-lane offset ((w0 - pw + e)*C + c)*4 then needs no bounds code at all — left of the row it is "negative" (huge as
// unsigned), right of it >= the size, and the hardware range check returns 0 for both.  One v_add + one buffer_load per element; rows
// outside the volume are skipped by scalar branches; the tap weights come through the scalar offset.
template <typename T, int KW, int DIL, int TW>
__global__ __launch_bounds__(256) void cl_dwconv_rows_kernel(DwArgs p)
{
    constexpr int SEG = TW + (KW - 1) * DIL;
    constexpr int SB = sizeof(T);
    const T *inp = reinterpret_cast<const T *>(p.in), *gxp = reinterpret_cast<const T *>(p.gelu_x), *gap = reinterpret_cast<const T *>(p.gelu_add);
    T *outp = reinterpret_cast<T *>(p.out);
    const int cpb = p.C < 256 ? p.C : 256;
    const int rpb = 256 / cpb;
    const int c = blockIdx.z * cpb + threadIdx.x % cpb;
    const int bx = DLKA_XCD_BX(p.xcd_nx);   // XCD-aware: an XCD owns a contiguous slab of rows (neighbouring rows share their input rows)
    if (bx < 0) return;
    const int run = bx * rpb + threadIdx.x / cpb;
    const int runs_per_row = cdiv(p.W, TW);
    const int rows = p.B * p.D * p.H;
    if (run >= rows * runs_per_row || c >= p.C) return;
    const int w0 = (run % runs_per_row) * TW;
    const int row = wave_uniform(run / runs_per_row);
    const int h0 = row % p.H, d0 = (row / p.H) % p.D, b = row / (p.H * p.D);

    float acc[TW];
    const float bv = p.bias ? p.bias[c] : 0.f;
#pragma unroll
    for (int t = 0; t < TW; ++t) acc[t] = bv;

    const int cb = p.C * SB, cbw = p.C * 4;                  // bytes per voxel: activations / fp32 tap weights
    const unsigned rowbytes = (unsigned)(p.W * cb);
    const int vbase = (w0 - p.pw) * cb + c * SB;             // byte offset of segment element 0 inside the row (may be negative)
    const BufRsrc rwt = make_rsrc(p.wp, (size_t)p.kd * p.kh * KW * cbw);
    const unsigned cv = (unsigned)c * 4u;
    for (int i = 0; i < p.kd; ++i) {
        const int zd = d0 + i * p.dd - p.pd;
        if (zd < 0 || zd >= p.D) continue;                   // scalar
        for (int j = 0; j < p.kh; ++j) {
            const int zh = h0 + j * p.dh - p.ph;
            if (zh < 0 || zh >= p.H) continue;               // scalar
            const BufRsrc rr = make_rsrc(inp + ((long)(b * p.D + zd) * p.H + zh) * p.W * p.C, rowbytes);
            float seg[SEG];
#pragma unroll
            for (int e = 0; e < SEG; ++e) seg[e] = act_buf_load1<T>(rr, (unsigned)(vbase + e * cb));
            const unsigned ws = (unsigned)((i * p.kh + j) * KW) * (unsigned)cbw;
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const float wv = buf_load_f32_s(rwt, cv, ws + (unsigned)(k * cbw));
#pragma unroll
                for (int t = 0; t < TW; ++t) acc[t] = fmaf(wv, seg[t + k * DIL], acc[t]);
            }
        }
    }
    const long obase = (((long)(b * p.D + d0) * p.H + h0) * p.W + w0) * p.C + c;
    if (p.gelu_x) {   // uniform: a = GELU(h) feeds dw 5^3 and the gate, so gh = (ga_gate + ga_dw5) * gelu'(h) closes here
#pragma unroll
        for (int t = 0; t < TW; ++t)
            if (w0 + t < p.W) act_store1(outp, obase + (long)t * p.C, (acc[t] + act_load1(gap, obase + (long)t * p.C)) * dgelu_f(act_load1(gxp, obase + (long)t * p.C)));
        return;
    }
#pragma unroll
    for (int t = 0; t < TW; ++t)
        if (w0 + t < p.W) act_store1(outp, obase + (long)t * p.C, acc[t]);
    if (sizeof(T) == 4 && p.out_lo) {   // uniform: the bf16 copy the gathers / the backward pass read (mixed-precision DLKA_BF16 block)
        bf16_t *lo = reinterpret_cast<bf16_t *>(p.out_lo);
#pragma unroll
        for (int t = 0; t < TW; ++t)
            if (w0 + t < p.W) act_store1(lo, obase + (long)t * p.C, acc[t]);
    }
}

// Row groups of the TH-row kernels (shared by the kernel and its launcher): `regular` groups h0 = r + TH*DIL*q (r < DIL) over the full blocks of TH*DIL rows and, where 1 .. DIL
// rows remain behind them (TH == 2), tail groups of two consecutive rows from tail_row0 on; otherwise the remainder is one more (partly empty) block of regular groups.
struct DwRowGroups { int groups, regular, tail_row0; };
__host__ __device__ __forceinline__ DwRowGroups dw_row_groups(int H, int TH, int DIL)
{
    const int blk = TH * DIL, full = H / blk, rem = H - full * blk;
    DwRowGroups g;
    if (TH == 2 && full >= 1 && rem >= 1 && rem <= DIL) { g.regular = full * DIL; g.tail_row0 = full * blk; g.groups = g.regular + (rem + 1) / 2; }
    else { g.regular = DIL * ((H + blk - 1) / blk); g.tail_row0 = H; g.groups = g.regular; }
    return g;
}

// ... and with TH output rows per work-item, h0, h0 + DIL, ...: with the loads lean, the kernel sits on the L1 return path (one 256-byte
// wave load per segment element: 1.7 GB per launch at 32^3 for 7^3 dil 3), and output rows DIL apart share KH - 1 of their KH input
// rows — KH + TH - 1 segment loads per tap plane feed TH * KH row products.  The tap plane's KH*KW weights sit in registers for all.
// WL: the tap weights of the workgroup's channels sit in LDS (all kd*KH*KW taps x <= 32 channels, staged once) and are read where they are used, instead
// of a tap plane's KH*KW weights in registers: 7^3 then needs 125 instead of 174 registers — three waves per SIMD instead of two, and the stage-0 launch
// (2304 waves) fits the chip in ONE round (at two per SIMD: 2048 slots, the last 256 waves ran a second round alone).
template <typename T, int KW, int DIL, int TW, int TH, bool WL = false>
__global__ __launch_bounds__(256, WL ? 3 : 1) void cl_dwconv_rowsN_kernel(DwArgs p)
{
    constexpr int KH = KW;
    constexpr int SB = sizeof(T);
    const T *inp = reinterpret_cast<const T *>(p.in), *gxp = reinterpret_cast<const T *>(p.gelu_x), *gap = reinterpret_cast<const T *>(p.gelu_add);
    T *outp = reinterpret_cast<T *>(p.out);
    constexpr int SEG = TW + (KW - 1) * DIL;
    constexpr int NR = KH + TH - 1;                          // input rows per tap plane
    const int cpb = p.C < 256 ? p.C : 256;
    const int rpb = 256 / cpb;
    const int c = blockIdx.z * cpb + threadIdx.x % cpb;
    __shared__ __attribute__((aligned(16))) float Wl[WL ? KW * KH * KW * 32 : 4];   // [tap][channel of the workgroup]  (WL: cpb == 32, kd == KW)
    if (WL) {   // before any exit: every thread of the workgroup reaches the barrier
        const int n4 = p.kd * KH * KW * 8;   // 16-byte pieces: 8 per tap
        for (int e = threadIdx.x; e < n4; e += 256) {
            const int tap = e >> 3, q = e & 7;
            reinterpret_cast<f32x4 *>(Wl)[e] = *reinterpret_cast<const f32x4 *>(p.wp + (long)tap * p.C + blockIdx.z * 32 + 4 * q);
        }
        __syncthreads();
    }
    const int cl = threadIdx.x % cpb;
    const int bx = DLKA_XCD_BX(p.xcd_nx);   // XCD-aware: an XCD owns a contiguous slab of rows (neighbouring rows share their input rows)
    if (bx < 0) return;
    const int run = bx * rpb + threadIdx.x / cpb;
    const int runs_per_row = cdiv(p.W, TW);
    // row groups per (b, d) plane: h0 = r + TH*DIL*q, r < DIL — and (round 6) the 1 .. DIL rows that remain behind the last full block of TH*DIL rows as TAIL groups of two
    // rows each, walked one after the other by the same work-item: 32 rows at dilation 3 are 15 + 1 = 16 groups instead of 18 slots, i.e. 2048 waves for the stage-0 volume
    // instead of 2304 — two per SIMD instead of 2.25 (the SIMDs that took three set the time: scripts/time_dw_quant.py, 43.8 us at H = 32 and 48.3 at H = 48).
    const DwRowGroups rg = dw_row_groups(p.H, TH, DIL);
    const int groups = rg.groups;
    const long total = (long)p.B * p.D * groups * runs_per_row;
    if (run >= total || c >= p.C) return;
    const int w0 = (run % runs_per_row) * TW;
    const int gidx = wave_uniform(run / runs_per_row);
    const int grp = gidx % groups, d0 = (gidx / groups) % p.D, b = gidx / (groups * p.D);
    const bool tail = grp >= rg.regular;                     // scalar
    int h0 = tail ? rg.tail_row0 + 2 * (grp - rg.regular) : (grp % DIL) + (grp / DIL) * TH * DIL;
    if (h0 >= p.H) return;                                   // scalar
    const int npass = tail ? min(2, p.H - h0) : 1;

    const int cb = p.C * SB, cbw = p.C * 4;
    const unsigned rowbytes = (unsigned)(p.W * cb);
    const int vbase = (w0 - p.pw) * cb + c * SB;
    const BufRsrc rwt = make_rsrc(p.wp, (size_t)p.kd * KH * KW * cbw);
    const unsigned cv = (unsigned)c * 4u;
    const float bv = p.bias ? p.bias[c] : 0.f;
  for (int pass = 0; pass < npass; ++pass, ++h0) {           // (a tail work-item: its two rows, each as a single output row)
    float acc[TH][TW];
#pragma unroll
    for (int o = 0; o < TH; ++o)
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[o][t] = bv;

    for (int i = 0; i < p.kd; ++i) {
        const int zd = d0 + i * p.dd - p.pd;
        if (zd < 0 || zd >= p.D) continue;                   // scalar
        float wv[WL ? 1 : KH][KW];
        if (!WL) {
#pragma unroll
            for (int j = 0; j < KH; ++j)
#pragma unroll
                for (int k = 0; k < KW; ++k) wv[WL ? 0 : j][k] = buf_load_f32_s(rwt, cv, (unsigned)(((i * KH + j) * KW + k) * cbw));
        }
        const float *wl = Wl + i * KH * KW * 32 + cl;
        const T *plane = inp + ((long)(b * p.D + zd) * p.H) * p.W * p.C;
#pragma unroll
        for (int r = 0; r < NR; ++r) {                       // input row h0 - ph + r*DIL: tap row r - o of output row o
            const int zh = h0 - p.ph + r * DIL;
            if (zh < 0 || zh >= p.H || (tail && r >= KH)) continue;   // scalar (a tail pass has no second output row: the rows beyond KH - 1 feed nothing)
            const BufRsrc rr = make_rsrc(plane + (long)zh * p.W * p.C, rowbytes);
            float seg[SEG];
#pragma unroll
            for (int e = 0; e < SEG; ++e) seg[e] = act_buf_load1<T>(rr, (unsigned)(vbase + e * cb));
#pragma unroll
            for (int o = 0; o < TH; ++o) {
                if (r - o < 0 || r - o >= KH) continue;      // compile time
                if (o > 0 && tail) continue;                 // scalar
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    const float wk = WL ? wl[((r - o) * KW + k) * 32] : wv[WL ? 0 : r - o][k];
#pragma unroll
                    for (int t = 0; t < TW; ++t) acc[o][t] = fmaf(wk, seg[t + k * DIL], acc[o][t]);
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < TH; ++o) {
        if (h0 + o * DIL >= p.H || (o > 0 && tail)) break;   // scalar
        const long obase = (((long)(b * p.D + d0) * p.H + h0 + o * DIL) * p.W + w0) * p.C + c;
        if (p.gelu_x) {
#pragma unroll
            for (int t = 0; t < TW; ++t)
                if (w0 + t < p.W) act_store1(outp, obase + (long)t * p.C, (acc[o][t] + act_load1(gap, obase + (long)t * p.C)) * dgelu_f(act_load1(gxp, obase + (long)t * p.C)));
        } else {
#pragma unroll
            for (int t = 0; t < TW; ++t)
                if (w0 + t < p.W) act_store1(outp, obase + (long)t * p.C, acc[o][t]);
            if (sizeof(T) == 4 && p.out_lo) {   // uniform: bf16 copy (see cl_args.h: DwArgs::out_lo)
                bf16_t *lo = reinterpret_cast<bf16_t *>(p.out_lo);
#pragma unroll
                for (int t = 0; t < TW; ++t)
                    if (w0 + t < p.W) act_store1(lo, obase + (long)t * p.C, acc[o][t]);
            }
        }
    }
  }
}

// ... and with TWO output planes per work-item as well (d0 and d0 + DIL, on top of the two rows h0 and h0 + DIL): the kernel above sits exactly on the L1 return
// path (1.7 GB of wave loads per 7^3 launch at 32^3 = 42 us at 64 bytes / clock / CU, and it measures 42; the vector units are NOT the limit: independent fp32 FMAs
// issue at ~110 - 125 lanes per clock and CU, scripts/ubench/fma_rate.hip), and output planes DIL apart share KD - 1 of their KD input planes exactly as the rows do:
// KD + 1 planes of (KH + 1) segment rows feed 2 x 2 x KH x KW x TW products — 7.5 instead of 3.8 FMAs per loaded element at 7^3.
// Tap weights in LDS as in the WL variant above, plus ONE all-zero tap plane: the first / last input plane of a pair serves only one of the two outputs, and the
// other output multiplies it with zeros instead of branching (the FMAs are not what this kernel waits for).  C = 32 (one 32-channel group per workgroup).
#ifndef DLKA_TD2_OCC
#define DLKA_TD2_OCC 2   // (3: 168 registers with 14 - 19 spilled at 7^3; measured 60 against 51 us)
#endif
template <typename T, int KW, int DIL, int TW>
__global__ __launch_bounds__(256, DLKA_TD2_OCC) void cl_dwconv_rows2d_kernel(DwArgs p)
{
    constexpr int KH = KW, KD = KW, TH = 2, TD = 2;
    constexpr int SB = sizeof(T);
    const T *inp = reinterpret_cast<const T *>(p.in), *gxp = reinterpret_cast<const T *>(p.gelu_x), *gap = reinterpret_cast<const T *>(p.gelu_add);
    T *outp = reinterpret_cast<T *>(p.out);
    constexpr int SEG = TW + (KW - 1) * DIL;
    constexpr int NR = KH + TH - 1, NP = KD + TD - 1;        // input rows per plane, input planes per work-item
    constexpr int cpb = 32, rpb = 8;
    constexpr int PLANE = KH * KW * 32;                      // floats of one tap plane in LDS
    __shared__ __attribute__((aligned(16))) float Wl[(KD + 1) * PLANE];   // [tap plane (KD = zeros)][tap row][tap col][channel of the workgroup]
    {
        const int n4 = KD * KH * KW * 8;   // 16-byte pieces: 8 per tap
        for (int e = threadIdx.x; e < n4 + KH * KW * 8; e += 256) {
            const int tap = e >> 3, q = e & 7;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < n4) v = *reinterpret_cast<const f32x4 *>(p.wp + (long)tap * p.C + blockIdx.z * 32 + 4 * q);
            reinterpret_cast<f32x4 *>(Wl)[e] = v;
        }
        __syncthreads();
    }
    const int c = blockIdx.z * cpb + threadIdx.x % cpb, cl = threadIdx.x % cpb;
    const int bx = DLKA_XCD_BX(p.xcd_nx);
    if (bx < 0) return;
    const int run = bx * rpb + threadIdx.x / cpb;
    const int runs_per_row = cdiv(p.W, TW);
    const int hgroups = DIL * cdiv(p.H, TH * DIL), dgroups = DIL * cdiv(p.D, TD * DIL);   // h0 = r + TH*DIL*q, d0 = r' + TD*DIL*q'  (r, r' < DIL)
    const long total = (long)p.B * dgroups * hgroups * runs_per_row;
    if (run >= total || c >= p.C) return;
    const int w0 = (run % runs_per_row) * TW;
    const int gidx = wave_uniform(run / runs_per_row);
    const int hg = gidx % hgroups, dg = (gidx / hgroups) % dgroups, b = gidx / (hgroups * dgroups);
    const int h0 = (hg % DIL) + (hg / DIL) * TH * DIL, d0 = (dg % DIL) + (dg / DIL) * TD * DIL;
    if (h0 >= p.H || d0 >= p.D) return;                      // scalar
    const bool second = d0 + DIL < p.D;                      // the pair's second output plane exists

    float acc[TD][TH][TW];
    const float bv = p.bias ? p.bias[c] : 0.f;
#pragma unroll
    for (int od = 0; od < TD; ++od)
#pragma unroll
        for (int o = 0; o < TH; ++o)
#pragma unroll
            for (int t = 0; t < TW; ++t) acc[od][o][t] = bv;

    const int cb = p.C * SB;
    const unsigned rowbytes = (unsigned)(p.W * cb);
    const int vbase = (w0 - p.pw) * cb + c * SB;
#pragma unroll 1
    for (int ip = 0; ip < NP; ++ip) {
        const int zd = d0 - p.pd + ip * DIL;
        if (zd < 0 || zd >= p.D) continue;                   // scalar
        const float *wl0 = Wl + (ip < KD ? ip : KD) * PLANE + cl;                         // output plane d0: tap plane ip
        const float *wl1 = Wl + ((ip >= 1 && second) ? ip - 1 : KD) * PLANE + cl;         // output plane d0 + DIL: tap plane ip - 1
        const T *plane = inp + ((long)(b * p.D + zd) * p.H) * p.W * p.C;
#pragma unroll
        for (int r = 0; r < NR; ++r) {                       // input row h0 - ph + r*DIL: tap row r - o of output row o
            const int zh = h0 - p.ph + r * DIL;
            if (zh < 0 || zh >= p.H) continue;               // scalar
            const BufRsrc rr = make_rsrc(plane + (long)zh * p.W * p.C, rowbytes);
            float seg[SEG];
#pragma unroll
            for (int e = 0; e < SEG; ++e) seg[e] = act_buf_load1<T>(rr, (unsigned)(vbase + e * cb));
#pragma unroll
            for (int o = 0; o < TH; ++o) {
                if (r - o < 0 || r - o >= KH) continue;      // compile time
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    const float wk0 = wl0[((r - o) * KW + k) * 32], wk1 = wl1[((r - o) * KW + k) * 32];
#pragma unroll
                    for (int t = 0; t < TW; ++t) {
                        acc[0][o][t] = fmaf(wk0, seg[t + k * DIL], acc[0][o][t]);
                        acc[1][o][t] = fmaf(wk1, seg[t + k * DIL], acc[1][o][t]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int od = 0; od < TD; ++od) {
        if (d0 + od * DIL >= p.D) break;                     // scalar
#pragma unroll
        for (int o = 0; o < TH; ++o) {
            if (h0 + o * DIL >= p.H) break;                  // scalar
            const long obase = (((long)(b * p.D + d0 + od * DIL) * p.H + h0 + o * DIL) * p.W + w0) * p.C + c;
            if (p.gelu_x) {
#pragma unroll
                for (int t = 0; t < TW; ++t)
                    if (w0 + t < p.W) act_store1(outp, obase + (long)t * p.C, (acc[od][o][t] + act_load1(gap, obase + (long)t * p.C)) * dgelu_f(act_load1(gxp, obase + (long)t * p.C)));
            } else {
#pragma unroll
                for (int t = 0; t < TW; ++t)
                    if (w0 + t < p.W) act_store1(outp, obase + (long)t * p.C, acc[od][o][t]);
                if (sizeof(T) == 4 && p.out_lo) {   // uniform: bf16 copy (see cl_args.h: DwArgs::out_lo)
                    bf16_t *lo = reinterpret_cast<bf16_t *>(p.out_lo);
#pragma unroll
                    for (int t = 0; t < TW; ++t)
                        if (w0 + t < p.W) act_store1(lo, obase + (long)t * p.C, acc[od][o][t]);
                }
            }
        }
    }
}

// Third attempt at the family (round 6) — OPT-IN (DLKA_DW_2P=1 | 2), measured SLOWER than the row kernel: 7^3 dilation 3 49.5 against 43.4 us, 5^3 21.9 against
// 20.2 at (32 channels, 32^3); 30.7 against 19.1 / 19.2 against 16.9 at (64, 16^3)  (profiles/r10_notes.md).  What it does:
//  * input rows go through a RING of register segments; the loads of row r + RING - 1 are issued BEFORE row r computes.  Every load is unconditional (a row outside the
//    volume gets a zero-sized descriptor: the hardware range check returns 0, no memory is touched) so that the compiler's vmcnt bookkeeping stays exact across the scalar
//    branches that skip the FMAs of such rows; planes outside the volume are not visited at all.
//  * the two output ROWS h0 and h0 + DIL of a work-item are the two halves of v_pk_fma_f32: acc.xy += {w[r][k], w[r - 1][k]} * seg.xx — the broadcast is an op_sel of the
//    instruction and the weight pair is ONE ds_read2_b32 (LDS layout [tap plane][tap column][zero row, tap rows in DEscending order, zero row][channel]: the two tap rows
//    are 32 floats apart in register order; the zero rows serve the first / last input row), read one row AHEAD of its use: a tap costs one LDS read, one address add and
//    TW packed FMAs, no v_mov (the row kernels pack along W: 322 v_mov beside 392 v_pk_fma per tap plane at 7^3 dilation 3).  Per output the FMA chain is the row
//    kernel's (plus zero products): bit-identical to it.
//  * lane = channel, TW outputs along W, two rows x two planes (d0, d0 + DIL: input plane ip carries tap plane ip for the first, ip - 1 for the second).
// Why it loses (ablations of this kernel at 7^3, us: as is 49.5 / no weight staging 44.1 / no input loads 38.6 / no FMAs 27.8; scripts/ubench/fma_rate.hip, VGPR operands):
// ONE wave issues a vector instruction every ~5 - 6 clocks whatever it is (v_fma_f32 115 FMA lanes per ns and CU at one wave per SIMD, 203 at two, 241 at four; v_pk_fma_f32 with
// three VGPR sources 194 / 253 / 278), so the FMA pipe only fills at three or four waves per SIMD — and a 2 x 2 x 8 register tile with its ring is 232 registers and
// 1156 waves for the whole stage-0 volume: one or two waves per SIMD, the SIMDs that got two set the time.  4-wide work-items (twice the waves, 144 - 210 registers, 64 KB
// of LDS per workgroup: still two per SIMD) measured 56 - 60 us.  The family's lever is occupancy at a small instruction count, not reuse per load: the row kernel
// (three waves per SIMD, 40 % of its vector instructions are register moves and address adds) stays the default.
template <typename T, int KW, int DIL, int TW, int RING>
__global__ __launch_bounds__(256, 2) void cl_dwconv_rows2p_kernel(DwArgs p)
{
    constexpr int KH = KW, KD = KW, TH = 2;
    constexpr int SB = sizeof(T);
    constexpr int SEG = TW + (KW - 1) * DIL;
    constexpr int NR = KH + TH - 1, NP = KD + 1;             // input rows per plane, input planes per work-item
    constexpr int PF = RING - 1;                             // rows in flight ahead of the one that computes
    static_assert(NR % RING == 0, "the ring slot of a row must not depend on the plane");
    constexpr int KSTR = (KH + 2) * 32, PLANE = KW * KSTR;   // floats of one tap column (KH tap rows between two zero rows) / one tap plane in LDS
    const T *inp = reinterpret_cast<const T *>(p.in), *gxp = reinterpret_cast<const T *>(p.gelu_x), *gap = reinterpret_cast<const T *>(p.gelu_add);
    T *outp = reinterpret_cast<T *>(p.out);
    __shared__ __attribute__((aligned(16))) float Wl[KD * PLANE];   // [tap plane][tap column k][zero row, tap rows KH - 1 .. 0, zero row][channel of the workgroup]
    {
#pragma unroll 4
        for (int e = threadIdx.x; e < KD * PLANE / 4; e += 256) {
            const int q = e & 7, slot = (e >> 3) % (KH + 2), k = ((e >> 3) / (KH + 2)) % KW, i = (e >> 3) / ((KH + 2) * KW);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (slot >= 1 && slot <= KH) v = *reinterpret_cast<const f32x4 *>(p.wp + (long)((i * KH + KH - slot) * KW + k) * p.C + blockIdx.z * 32 + 4 * q);   // tap row KH - slot
            reinterpret_cast<f32x4 *>(Wl)[e] = v;
        }
        __syncthreads();
    }
    const int cl = threadIdx.x % 32, c = blockIdx.z * 32 + cl;
    const int bx = DLKA_XCD_BX(p.xcd_nx);
    if (bx < 0) return;
    const int run = bx * 8 + threadIdx.x / 32;
    const int runs_per_row = cdiv(p.W, TW);
    const int hgroups = DIL * cdiv(p.H, TH * DIL), dgroups = DIL * cdiv(p.D, 2 * DIL);   // h0 = r + 2*DIL*q, d0 = r' + 2*DIL*q'  (r, r' < DIL)
    const long total = (long)p.B * dgroups * hgroups * runs_per_row;
    if (run >= total) return;
    const int w0 = (run % runs_per_row) * TW;
    const int gidx = wave_uniform(run / runs_per_row);
    const int hg = gidx % hgroups, dg = (gidx / hgroups) % dgroups, b = gidx / (hgroups * dgroups);
    const int h0 = (hg % DIL) + (hg / DIL) * TH * DIL, d0 = (dg % DIL) + (dg / DIL) * 2 * DIL;
    if (h0 >= p.H || d0 >= p.D) return;                      // scalar

    f32x2 acc[2][TW];                                        // [output plane d0 + od*DIL][w]; .x: row h0, .y: row h0 + DIL
    const float bv = p.bias ? p.bias[c] : 0.f;
#pragma unroll
    for (int od = 0; od < 2; ++od)
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[od][t] = f32x2{bv, bv};

    const int cb = p.C * SB;
    const unsigned rowbytes = (unsigned)(p.W * cb);
    const int vbase = (w0 - p.pw) * cb + c * SB;
    // input planes zd = d0 - pd + ip*DIL inside the volume: ip in [ip_lo, ip_hi)
    const int ip_lo = d0 < p.pd ? cdiv(p.pd - d0, DIL) : 0;
    const int ip_hi = min(NP, cdiv(p.D - d0 + p.pd, DIL));
    float seg[RING][SEG];
    // descriptor of input row rr of plane ip (zero-sized outside the volume)
    auto row_rsrc = [&](int ip, int rr) {
        const int zd = d0 - p.pd + ip * DIL, zh = h0 - p.ph + rr * DIL;
        const bool ok = ip < ip_hi && zh >= 0 && zh < p.H;
        return make_rsrc(inp + ((long)(b * p.D + (ok ? zd : 0)) * p.H + (ok ? zh : 0)) * p.W * p.C, ok ? rowbytes : 0u);
    };
#pragma unroll
    for (int r = 0; r < PF; ++r) {
        const BufRsrc rr = row_rsrc(ip_lo, r);
#pragma unroll
        for (int e = 0; e < SEG; ++e) seg[r % RING][e] = act_buf_load1<T>(rr, (unsigned)(vbase + e * cb));
    }
    // tap weights of the row about to compute, read from LDS one row AHEAD (with one or two waves per SIMD nothing else hides the LDS latency: read at the point of
    // use, every tap cost ~60 stall clocks beside its 32 clocks of FMAs).  wr[od][k] = {w[i][r][k], w[i][r - 1][k]}, i = ip - od clamped into the staged planes (an
    // output plane that does not take this input plane skips its FMAs by a scalar branch; what was read for it is dropped).
    f32x2 wr[2][KW];
    auto load_w = [&](f32x2 (&w)[2][KW], int ip, int r) {
#pragma unroll
        for (int od = 0; od < 2; ++od) {
            const float *wl = Wl + min(max(ip - od, 0), KD - 1) * PLANE + (KH - r) * 32 + cl;
#pragma unroll
            for (int k = 0; k < KW; ++k) w[od][k] = f32x2{wl[k * KSTR], wl[k * KSTR + 32]};   // tap rows r (slot KH - r) and r - 1 (the next slot): one ds_read2_b32, in register order
        }
    };
    load_w(wr, ip_lo, 0);
#pragma unroll 1
    for (int ip = ip_lo; ip < ip_hi; ++ip) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {                       // input row h0 - ph + r*DIL: tap row r of output row h0, r - 1 of h0 + DIL
            {                                                // row r + PF (of this plane or the next one) into the slot row r - 1 left
                const BufRsrc rr = row_rsrc(ip + (r + PF) / NR, (r + PF) % NR);
#pragma unroll
                for (int e = 0; e < SEG; ++e) seg[(r + PF) % RING][e] = act_buf_load1<T>(rr, (unsigned)(vbase + e * cb));
            }
            f32x2 wn[2][KW];
            load_w(wn, ip + (r + 1) / NR, (r + 1) % NR);
            const int zh = h0 - p.ph + r * DIL;
            if (zh >= 0 && zh < p.H) {                       // scalar (else: the row's segment is zeros, nothing to add)
#pragma unroll
                for (int od = 0; od < 2; ++od) {
                    const int i = ip - od;                   // tap plane of output plane d0 + od*DIL
                    if (i < 0 || i >= KD) continue;          // scalar
#pragma unroll
                    for (int k = 0; k < KW; ++k) {
#pragma unroll
                        for (int t = 0; t < TW; ++t) {
                            const float sv = seg[r % RING][t + k * DIL];
                            acc[od][t] = pk_fma(wr[od][k], f32x2{sv, sv}, acc[od][t]);
                        }
                    }
                }
            }
#pragma unroll
            for (int od = 0; od < 2; ++od)
#pragma unroll
                for (int k = 0; k < KW; ++k) wr[od][k] = wn[od][k];
        }
    }
#pragma unroll
    for (int od = 0; od < 2; ++od) {
        if (d0 + od * DIL >= p.D) break;                     // scalar
#pragma unroll
        for (int o = 0; o < TH; ++o) {
            if (h0 + o * DIL >= p.H) break;                  // scalar
            const long obase = (((long)(b * p.D + d0 + od * DIL) * p.H + h0 + o * DIL) * p.W + w0) * p.C + c;
            if (p.gelu_x) {
#pragma unroll
                for (int t = 0; t < TW; ++t)
                    if (w0 + t < p.W) act_store1(outp, obase + (long)t * p.C, (acc[od][t][o] + act_load1(gap, obase + (long)t * p.C)) * dgelu_f(act_load1(gxp, obase + (long)t * p.C)));
            } else {
#pragma unroll
                for (int t = 0; t < TW; ++t)
                    if (w0 + t < p.W) act_store1(outp, obase + (long)t * p.C, acc[od][t][o]);
                if (sizeof(T) == 4 && p.out_lo) {   // uniform: bf16 copy (see cl_args.h: DwArgs::out_lo)
                    bf16_t *lo = reinterpret_cast<bf16_t *>(p.out_lo);
#pragma unroll
                    for (int t = 0; t < TW; ++t)
                        if (w0 + t < p.W) act_store1(lo, obase + (long)t * p.C, acc[od][t][o]);
                }
            }
        }
    }
}

// reference layout W[c][1][kd][kh][kw] -> Wp[tap][c]; flip = 1 reverses the taps (data gradient)
__global__ void cl_dw_prep_weight_kernel(const float *__restrict__ w, float *__restrict__ wp, int C, int K, int flip)
{
    const int n = C * K;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int c = e % C, tap = e / C;
        wp[e] = w[(long)c * K + (flip ? K - 1 - tap : tap)];
    }
}

int launch_cl_dw_prep_weight(const float *w, float *wp, int C, int K, int flip, hipStream_t st)
{
    DLKA_LAUNCH(cl_dw_prep_weight_kernel, dim3(cdiv(C * K, 256)), dim3(256), 0, st, w, wp, C, K, flip);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

static std::atomic<long> g_dw2p_launches{0};   // dlka_dwconv_2p_launch_count (include/dlka.h): diagnostics
#ifndef DLKA_DW_2P_MIN_WAVES
#define DLKA_DW_2P_MIN_WAVES 1000000000   // (no launch size at which it won: off unless DLKA_DW_2P asks for it)
#endif
// kw, dil_w select the instantiation; returns DLKA_ERR_UNSUPPORTED for other shapes (caller falls back to conv.hip)
template <typename T>
static int launch_cl_dwconv_t(const DwArgs &a, int kw, int dil_w, hipStream_t st)
{
    constexpr int TW = 8;
    constexpr bool F32 = sizeof(T) == 4;   // the bf16 storage path is instantiated for the two D-LKA shapes only (5^3, 7^3 dil 3)
    const int cpb = a.C < 256 ? a.C : 256, rpb = 256 / cpb;
    const long runs = (long)a.B * a.D * a.H * cdiv(a.W, TW);
    dim3 grid((unsigned)cdivl(runs, rpb), 1, cdiv(a.C, cpb)), block(256);
    if (256 % cpb != 0) return DLKA_ERR_UNSUPPORTED;
    if (!F32 && !((kw == 5 && dil_w == 1) || (kw == 7 && dil_w == 3))) return DLKA_ERR_UNSUPPORTED;
    constexpr bool v1 = false;
    constexpr int abl0 = 0;
    // the rows kernel needs (b, d, h) uniform per wave: the 64 / cpb runs of a wave must not straddle two rows
    const int wpr = cpb < 64 ? 64 / cpb : 1;
    if (!v1 && !abl0 && cdiv(a.W, TW) % wpr == 0 && (long)a.W * a.C * 4 < (1l << 31)) {
        DwArgs ax = a;
        ax.xcd_nx = 0;
        auto swz = [&](dim3 &g) { if (xcd_swizzle_enabled() && g.x >= (unsigned)xcd_min_blocks()) { ax.xcd_nx = (int)g.x; g.x = xcd_grid(ax.xcd_nx); } };
        constexpr int th_env = 0;
        const bool cubic = a.kd == kw && a.kh == kw && a.dd == dil_w && a.dh == dil_w;
        const int th = th_env ? th_env : 2;
        if ((th == 2 || (th == 3 && F32)) && cubic && a.H >= th * dil_w && ((kw == 7 && dil_w == 3) || (kw == 5 && dil_w == 1))) {
            // Software-pipelined kernel with two output rows on the halves of v_pk_fma_f32 and two output planes per work-item (cl_dwconv_rows2p_kernel): OPT-IN, measured
            // slower (see its header).  DLKA_DW_2P (read per launch): unset / 0 = never, 1 = wherever its geometry fits (5^3: two rows ahead), 2 = the same with one row ahead.
            {
                const char *e3 = getenv("DLKA_DW_2P");
                const int mode = e3 ? atoi(e3) : -1;
                const bool ok2p = th == 2 && a.C % 32 == 0 && a.D >= 2 * dil_w && a.pd == (kw - 1) / 2 * dil_w && a.ph == a.pd && !a.out_blk && cdiv(a.W, TW) % 2 == 0;
                const long runs3 = (long)a.B * dil_w * cdiv(a.D, 2 * dil_w) * dil_w * cdiv(a.H, th * dil_w) * cdiv(a.W, TW);
                const long waves3 = runs3 / 2 * (a.C / 32);
                if (ok2p && (mode > 0 || (mode < 0 && waves3 >= DLKA_DW_2P_MIN_WAVES))) {
                    dim3 grid3((unsigned)cdivl(runs3, 8), 1, a.C / 32);
                    swz(grid3);
                    if (kw == 7) { auto k = cl_dwconv_rows2p_kernel<T, 7, 3, TW, 2>; DLKA_LAUNCH(k, grid3, block, 0, st, ax); }   // (a ring of 4 spills: 262 registers' worth)
                    else {
                        if (mode == 2) { auto k = cl_dwconv_rows2p_kernel<T, 5, 1, TW, 2>; DLKA_LAUNCH(k, grid3, block, 0, st, ax); }
                        else { auto k = cl_dwconv_rows2p_kernel<T, 5, 1, TW, 3>; DLKA_LAUNCH(k, grid3, block, 0, st, ax); }
                    }
                    DLKA_CHECK_LAUNCH();
                    g_dw2p_launches.fetch_add(1, std::memory_order_relaxed);
                    return DLKA_OK;
                }
            }
            const long runs2 = (long)a.B * a.D * dw_row_groups(a.H, th, dil_w).groups * cdiv(a.W, TW);
            dim3 grid2((unsigned)cdivl(runs2, rpb), 1, cdiv(a.C, cpb));
            swz(grid2);
            // Small volumes: with 8 outputs along W per work-item the launch is a fraction of a wave per SIMD (16^3 x 64 channels: 576 waves) and the
            // kernel's time is ONE wave's serial instruction stream (~7.7 K instructions at 7^3); 4 outputs per work-item double the waves
            // and halve the stream.  Chosen where the 8-wide grid leaves SIMDs empty.
            const long waves8 = runs2 * cpb / 64 * cdiv(a.C, cpb);
            constexpr bool no_tw4 = false;
            // (measured at the stage shapes, us, 4 vs 8 wide: 5^3 13.3 / 17.9 at 16^3, 12.7 / 17.4 at 8^3, 11.0 / 12.4 at 4^3; 7^3 19.0 / 18.5 at 16^3, 8.8 / 9.5 at 8^3)
            #ifndef DLKA_DW_TW4_5
#define DLKA_DW_TW4_5 1024   // (waves of the 8-wide grid below which the 4-wide one is taken: 5^3 / 7^3; -D... for scripts/build_variant.sh)
#define DLKA_DW_TW4_7 384
#endif
            if (!no_tw4 && th == 2 && waves8 < (kw == 5 ? DLKA_DW_TW4_5 : DLKA_DW_TW4_7) && a.W % 4 == 0 && cdiv(a.W, 4) % wpr == 0) {
                const long runs4 = (long)a.B * a.D * dw_row_groups(a.H, th, dil_w).groups * cdiv(a.W, 4);
                dim3 grid4((unsigned)cdivl(runs4, rpb), 1, cdiv(a.C, cpb));
                swz(grid4);
                if (kw == 7) { auto k = cl_dwconv_rowsN_kernel<T, 7, 3, 4, 2>; DLKA_LAUNCH(k, grid4, block, 0, st, ax); }
                else { auto k = cl_dwconv_rowsN_kernel<T, 5, 1, 4, 2>; DLKA_LAUNCH(k, grid4, block, 0, st, ax); }
                DLKA_CHECK_LAUNCH();
                return DLKA_OK;
            }
            // two output planes per work-item (cl_dwconv_rows2d_kernel): OPT-IN, DLKA_DW_TD2=1 (read per launch).  Measured SLOWER at (32, 32^3): 7^3 60 us (51 at two waves
            // per SIMD, without its 14 spilled registers) against 45, 5^3 33 against 28; stage-0 stack 5.67 against 5.49 ms (profiles/r06_notes.md) — half the loads per
            // output did not pay for half the workgroups, twice the LDS weight reads and the longer per-wave stream: these convs are not simply L1-bound.
            {
                const char *e2 = getenv("DLKA_DW_TD2");
                const bool td2 = (e2 && e2[0] == '1') && th == 2 && cpb == 32 && a.C == 32 && a.D >= 4 * dil_w && a.pd == (kw - 1) / 2 * dil_w && !a.out_blk;
                if (td2) {
                    const long runs3 = (long)a.B * dil_w * cdiv(a.D, 2 * dil_w) * dil_w * cdiv(a.H, th * dil_w) * cdiv(a.W, TW);
                    dim3 grid3((unsigned)cdivl(runs3, rpb), 1, 1);
                    ax.xcd_nx = 0;   // (swz(grid2) above may have set it for the one-plane grid)
                    swz(grid3);
                    if (kw == 7) { auto k = cl_dwconv_rows2d_kernel<T, 7, 3, TW>; DLKA_LAUNCH(k, grid3, block, 0, st, ax); }
                    else { auto k = cl_dwconv_rows2d_kernel<T, 5, 1, TW>; DLKA_LAUNCH(k, grid3, block, 0, st, ax); }
                    DLKA_CHECK_LAUNCH();
                    return DLKA_OK;
                }
            }
            constexpr bool no_wl = false;   // (measured: 54.3 -> 45.0 us at 32 channels / 32^3, profiles/r04_notes.md)
            if (kw == 7 && th == 2 && cpb == 32 && a.kd == 7 && !no_wl) { auto k = cl_dwconv_rowsN_kernel<T, 7, 3, TW, 2, true>; DLKA_LAUNCH(k, grid2, block, 0, st, ax); }
            else if (kw == 7 && th == 2) { auto k = cl_dwconv_rowsN_kernel<T, 7, 3, TW, 2>; DLKA_LAUNCH(k, grid2, block, 0, st, ax); }
            else if (kw == 5 && th == 2) { auto k = cl_dwconv_rowsN_kernel<T, 5, 1, TW, 2>; DLKA_LAUNCH(k, grid2, block, 0, st, ax); }
            else if constexpr (F32) {
                if (kw == 7) { auto k = cl_dwconv_rowsN_kernel<float, 7, 3, TW, 3>; DLKA_LAUNCH(k, grid2, block, 0, st, ax); }
                else { auto k = cl_dwconv_rowsN_kernel<float, 5, 1, TW, 3>; DLKA_LAUNCH(k, grid2, block, 0, st, ax); }
            }
            DLKA_CHECK_LAUNCH();
            return DLKA_OK;
        }
        bool done = true;
        dim3 grid1 = grid;
        swz(grid1);
        if (kw == 5 && dil_w == 1) { auto k = cl_dwconv_rows_kernel<T, 5, 1, TW>; DLKA_LAUNCH(k, grid1, block, 0, st, ax); }
        else if (kw == 7 && dil_w == 3) { auto k = cl_dwconv_rows_kernel<T, 7, 3, TW>; DLKA_LAUNCH(k, grid1, block, 0, st, ax); }
        else if constexpr (F32) {
            if (kw == 3 && dil_w == 1) { auto k = cl_dwconv_rows_kernel<float, 3, 1, TW>; DLKA_LAUNCH(k, grid1, block, 0, st, ax); }
            else if (kw == 5 && dil_w == 3) { auto k = cl_dwconv_rows_kernel<float, 5, 3, TW>; DLKA_LAUNCH(k, grid1, block, 0, st, ax); }
            else if (kw == 7 && dil_w == 1) { auto k = cl_dwconv_rows_kernel<float, 7, 1, TW>; DLKA_LAUNCH(k, grid1, block, 0, st, ax); }
            else done = false;
        } else done = false;
        if (done) { DLKA_CHECK_LAUNCH(); return DLKA_OK; }
    }
    if (kw == 5 && dil_w == 1) { auto k = cl_dwconv_kernel<T, 5, 1, TW>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
    else if (kw == 7 && dil_w == 3) {
        constexpr int abl = 0;
        if (F32 && abl == 1) { auto k = cl_dwconv_kernel<float, 7, 3, TW, 1>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
        else if (F32 && abl == 2) { auto k = cl_dwconv_kernel<float, 7, 3, TW, 2>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
        else { auto k = cl_dwconv_kernel<T, 7, 3, TW>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
    }
    else if constexpr (F32) {
        if (kw == 3 && dil_w == 1) { auto k = cl_dwconv_kernel<float, 3, 1, TW>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
        else if (kw == 5 && dil_w == 3) { auto k = cl_dwconv_kernel<float, 5, 3, TW>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
        else if (kw == 7 && dil_w == 1) { auto k = cl_dwconv_kernel<float, 7, 1, TW>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
        else return DLKA_ERR_UNSUPPORTED;
    }
    else return DLKA_ERR_UNSUPPORTED;
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_dwconv(const DwArgs &a, int kw, int dil_w, hipStream_t st)
{
    const int e = launch_cl_dwconv_lds(a, kw, dil_w, st);   // the LDS-brick kernel where the volume is large enough for it
    if (e != DLKA_ERR_UNSUPPORTED) return e;
    return a.act_bf16 ? launch_cl_dwconv_t<bf16_t>(a, kw, dil_w, st) : launch_cl_dwconv_t<float>(a, kw, dil_w, st);
}

}   // namespace dlka
extern "C" long dlka_dwconv_2p_launch_count(void) { return dlka::g_dw2p_launches.load(std::memory_order_relaxed); }
namespace dlka {

// ---------------------------------------------------------------------------------------------
// depthwise weight gradient:  gW[c][tap] = sum_{b,d,h,w} G[b,d,h,w][c] * in[b, d+i*dd-pd, h+j*dh-ph, w+k*dw-pw][c]
// grid.y = (i, j) tap rows; a work-item owns channel c and strides over (b, d, h, W-run); KW accumulators.
// Work-items of a block that share c fold through LDS, then one fp32 atomic per (c, tap) per block into gWp[tap][c].
// ---------------------------------------------------------------------------------------------
template <typename T, int KW, int DIL, int TW>
__global__ __launch_bounds__(256) void cl_dwconv_wgrad_kernel(DwWgradArgs p)
{
    constexpr int SEG = TW + (KW - 1) * DIL;
    const T *inp = reinterpret_cast<const T *>(p.in), *gin = reinterpret_cast<const T *>(p.g);
    __shared__ float red[256 * KW];
    const int cpb = p.C < 256 ? p.C : 256, rpb = 256 / cpb;
    const int c = blockIdx.z * cpb + threadIdx.x % cpb;
    const int rsub = threadIdx.x / cpb;
    const int i = blockIdx.y / p.kh, j = blockIdx.y % p.kh;
    const int runs_per_row = cdiv(p.W, TW);
    const int rows = p.B * p.D * p.H;
    const long run_lo = (long)blockIdx.x * p.rows_per_block * runs_per_row;
    const long run_hi = min((long)rows * runs_per_row, run_lo + (long)p.rows_per_block * runs_per_row);
    float acc[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) acc[k] = 0.f;
    float bsum = 0.f;
    const bool want_bias = p.gb && blockIdx.y == 0;   // one (i, j) slice also folds the column sums of g
    if (c < p.C) {
        for (long run = run_lo + rsub; run < run_hi; run += rpb) {
            const int w0 = (int)(run % runs_per_row) * TW;
            const int row = (int)(run / runs_per_row);
            const int h0 = row % p.H, d0 = (row / p.H) % p.D, b = row / (p.H * p.D);
            const int zd = d0 + i * p.dd - p.pd, zh = h0 + j * p.dh - p.ph;
            const bool rows_ok = !(zd < 0 || zd >= p.D || zh < 0 || zh >= p.H);
            if (!rows_ok && !want_bias) continue;
            const T *gp = gin + (((long)(b * p.D + d0) * p.H + h0) * p.W + w0) * p.C + c;
            float gv[TW], seg[SEG];
#pragma unroll
            for (int t = 0; t < TW; ++t) gv[t] = (w0 + t < p.W) ? act_load1(gp, (long)t * p.C) : 0.f;
            if (want_bias) {
#pragma unroll
                for (int t = 0; t < TW; ++t) bsum += gv[t];
            }
            if (!rows_ok) continue;
            const T *rowp = inp + (((long)(b * p.D + zd) * p.H + zh) * p.W) * p.C + c;
#pragma unroll
            for (int e = 0; e < SEG; ++e) {
                const int zw = w0 - p.pw + e;
                seg[e] = (zw >= 0 && zw < p.W) ? act_load1(rowp, (long)zw * p.C) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < KW; ++k)
#pragma unroll
                for (int t = 0; t < TW; ++t) acc[k] = fmaf(gv[t], seg[t + k * DIL], acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < KW; ++k) red[k * 256 + threadIdx.x] = acc[k];
    __syncthreads();
    if (want_bias) {   // uniform per block; reuses the k = 0 slice after the weight sums are read below
        float a0 = 0.f;
        if (threadIdx.x < cpb && c < p.C)
            for (int r = 0; r < rpb; ++r) a0 += red[r * cpb + threadIdx.x];
        __syncthreads();
        red[threadIdx.x] = bsum;
        __syncthreads();
        if (threadIdx.x < cpb && c < p.C) {
            float bs = 0.f;
            for (int r = 0; r < rpb; ++r) bs += red[r * cpb + threadIdx.x];
            atomicAdd(p.gb + c, bs);
            atomicAdd(p.gwp + (long)((i * p.kh + j) * KW) * p.C + c, a0);
#pragma unroll
            for (int k = 1; k < KW; ++k) {
                float a = 0.f;
                for (int r = 0; r < rpb; ++r) a += red[k * 256 + r * cpb + threadIdx.x];
                atomicAdd(p.gwp + (long)((i * p.kh + j) * KW + k) * p.C + c, a);
            }
        }
        return;
    }
    if (threadIdx.x < cpb && c < p.C) {
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            float a = 0.f;
            for (int r = 0; r < rpb; ++r) a += red[k * 256 + r * cpb + threadIdx.x];
            atomicAdd(p.gwp + (long)((i * p.kh + j) * KW + k) * p.C + c, a);
        }
    }
}

// Second generation: the INPUT row segment is the stationary operand.  A work-item owns (channel c, W-run, kd index i) and walks
// over input rows (b, zd, zh); the row segment it loads (SEG values) meets the KH grad_output rows (zd - i*dd + pd, zh - j*dh + ph)
// that pair with it, KH*KW accumulators in registers: SEG + KH*TW loads feed KH*KW*TW FMAs (4.8 FMA per load for 7^3 dil 3, against
// 1.65 in the first version, which re-read the input segment for every (i, j) — that kernel was bound by the L1 request rate:
// 177 us at C=32 / 32^3, profiles/archive/r01n).  grid.y = kd.  The bias gradient rides on the (i, j) = centre-tap pairing, which visits every
// grad_output element exactly once.
template <typename T, int KW, int DIL, int TW>
__global__ __launch_bounds__(256) void cl_dwconv_wgrad2_kernel(DwWgradArgs p)
{
    constexpr int KH = KW;
    constexpr int SB = sizeof(T);
    const T *inp = reinterpret_cast<const T *>(p.in), *gin = reinterpret_cast<const T *>(p.g);
    constexpr int SEG = TW + (KW - 1) * DIL;
    __shared__ float red[256 * KW];
    const int cpb = p.C < 256 ? p.C : 256, rpb = 256 / cpb;
    const int c = blockIdx.z * cpb + threadIdx.x % cpb;
    const int rsub = threadIdx.x / cpb;
    const int i = blockIdx.y;
    const int runs_per_row = cdiv(p.W, TW);
    const int rows = p.B * p.D * p.H;
    const int bx = DLKA_XCD_BX(p.xcd_nx);   // XCD-aware: an XCD owns a contiguous range of row chunks
    if (bx < 0) return;
    const long run_lo = (long)bx * p.rows_per_block * runs_per_row;
    const long run_hi = min((long)rows * runs_per_row, run_lo + (long)p.rows_per_block * runs_per_row);
    float acc[KH][KW];
#pragma unroll
    for (int j = 0; j < KH; ++j)
#pragma unroll
        for (int k = 0; k < KW; ++k) acc[j][k] = 0.f;
    float bsum = 0.f;
    const int doff = i * p.dd - p.pd;
    const bool bias_slice = p.gb && doff == 0;   // uniform per block
    if (c < p.C) {
        // every load is a buffer load against a descriptor of ONE row (built by the scalar unit: the rows of a wave's runs are
        // wave-uniform, the launcher checks it): no bounds code per element, see cl_dwconv_rows_kernel
        const int cb = p.C * SB;
        const unsigned rowbytes = (unsigned)(p.W * cb);
        for (long run = run_lo + rsub; run < run_hi; run += rpb) {
            const int w0 = (int)(run % runs_per_row) * TW;
            const int row = wave_uniform((int)(run / runs_per_row));
            const int zh = row % p.H, zd = (row / p.H) % p.D, b = row / (p.H * p.D);
            const int d0 = zd - doff;
            if (d0 < 0 || d0 >= p.D) continue;   // scalar
            const BufRsrc rx = make_rsrc(inp + (((long)(b * p.D + zd) * p.H + zh) * p.W) * p.C, rowbytes);
            const int vbase = (w0 - p.pw) * cb + c * SB;
            float seg[SEG];
#pragma unroll
            for (int e = 0; e < SEG; ++e) seg[e] = act_buf_load1<T>(rx, (unsigned)(vbase + e * cb));
            const T *gplane = gin + ((long)(b * p.D + d0) * p.H * p.W) * p.C;
            const unsigned gbase = (unsigned)(w0 * cb + c * SB);
#pragma unroll
            for (int j = 0; j < KH; ++j) {
                const int hoff = j * p.dh - p.ph;
                const int h0 = zh - hoff;
                if (h0 < 0 || h0 >= p.H) continue;   // scalar
                const BufRsrc rg = make_rsrc(gplane + (long)h0 * p.W * p.C, rowbytes);
                float gv[TW];
#pragma unroll
                for (int t = 0; t < TW; ++t) gv[t] = act_buf_load1<T>(rg, gbase + (unsigned)(t * cb));   // w0 + t >= W: beyond the row -> 0
                if (bias_slice && hoff == 0) {
#pragma unroll
                    for (int t = 0; t < TW; ++t) bsum += gv[t];
                }
#pragma unroll
                for (int k = 0; k < KW; ++k)
#pragma unroll
                    for (int t = 0; t < TW; ++t) acc[j][k] = fmaf(gv[t], seg[t + k * DIL], acc[j][k]);
            }
        }
    }
    // work-items of the block that share c fold through LDS; one fp32 atomic per (c, tap) per block
#pragma unroll
    for (int j = 0; j < KH; ++j) {
#pragma unroll
        for (int k = 0; k < KW; ++k) red[k * 256 + threadIdx.x] = acc[j][k];
        __syncthreads();
        if (threadIdx.x < cpb && c < p.C) {
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                float a = 0.f;
                for (int r = 0; r < rpb; ++r) a += red[k * 256 + r * cpb + threadIdx.x];
                atomicAdd(p.gwp + (long)((i * KH + j) * KW + k) * p.C + c, a);
            }
        }
        __syncthreads();
    }
    if (bias_slice) {
        red[threadIdx.x] = bsum;
        __syncthreads();
        if (threadIdx.x < cpb && c < p.C) {
            float bs = 0.f;
            for (int r = 0; r < rpb; ++r) bs += red[r * cpb + threadIdx.x];
            atomicAdd(p.gb + c, bs);
        }
    }
}

static inline dim3 block256() { return dim3(256); }

template <typename T>
static int launch_cl_dwconv_wgrad_t(DwWgradArgs a, int kw, int dil_w, hipStream_t st, bool zero_init)
{
    constexpr int TW = 8;
    constexpr bool F32 = sizeof(T) == 4;   // bf16 storage: the two D-LKA shapes only
    if (!F32 && !((kw == 5 && dil_w == 1) || (kw == 7 && dil_w == 3))) return DLKA_ERR_UNSUPPORTED;
    const int cpb = a.C < 256 ? a.C : 256;
    if (256 % cpb != 0) return DLKA_ERR_UNSUPPORTED;
    const int rows = a.B * a.D * a.H;
    static int xb_env = -1;
    if (xb_env < 0) xb_env = 0;
    // row-chunks: bounded atomics, enough waves to hide latency.  64 chunks, 128 for the large volumes (measured on the block graph:
    // 1.515 -> 1.462 ms at 32^3 with 128, no change at 16^3, 256 worse at 16^3; profiles/archive/r01p)
    const int xb_want = xb_env ? xb_env : (rows >= 1024 ? 128 : 64);
    int xb = rows < xb_want ? rows : xb_want;
    a.rows_per_block = cdiv(rows, xb);
    xb = cdiv(rows, a.rows_per_block);
    if (zero_init) {   // (the fused block zeroes all of its accumulation targets with one memset)
        if (launch_zero(a.gwp, (size_t)a.kd * a.kh * kw * a.C * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
        if (a.gb && launch_zero(a.gb, (size_t)a.C * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
    }
    constexpr bool v1 = false;
    // the centre tap row must exist for the bias ride-along (odd kernels with "same" padding: always)
    const bool centre = (a.pd % a.dd == 0) && (a.ph % a.dh == 0) && a.pd / a.dd < a.kd && a.ph / a.dh < a.kh;
    // measured (profiles/archive/r01o_dw_variants.txt): 7^3 dil 3 178 -> 103 us at 32^3, 38 -> 36 us at 16^3; 5^3 72 -> 66 us at 32^3 but 23.5 -> 26 us at 16^3
    // with row descriptors (profiles/archive/r01y_dww_rows.txt, us, new vs first generation): 7^3 dil 3: 55 vs 188 at 32^3, 31 vs 39 at 16^3, 22 vs 18 at 8^3;
    // 5^3: 38 vs 69 at 32^3, equal below
    const bool pays = kw >= 7 ? rows >= 512 : rows >= 1024;
    const int wpr = cpb < 64 ? 64 / cpb : 1;                       // runs per wave: must not straddle rows (row descriptors)
    const bool uniform = cdiv(a.W, TW) % wpr == 0 && (long)a.W * a.C * 4 < (1l << 31);
    if (!v1 && pays && uniform && a.kd == kw && a.kh == kw && a.dd == dil_w && a.dh == dil_w && (centre || !a.gb)) {
        dim3 grid2(xb, a.kd, cdiv(a.C, cpb));
        a.xcd_nx = 0;
        if (xcd_swizzle_enabled() && xb >= (unsigned)xcd_min_blocks()) { a.xcd_nx = xb; grid2.x = xcd_grid(xb); }
        if (kw == 5 && dil_w == 1) { auto k = cl_dwconv_wgrad2_kernel<T, 5, 1, TW>; DLKA_LAUNCH(k, grid2, block256(), 0, st, a); }
        else if (kw == 7 && dil_w == 3) { auto k = cl_dwconv_wgrad2_kernel<T, 7, 3, TW>; DLKA_LAUNCH(k, grid2, block256(), 0, st, a); }
        else if (F32 && kw == 3 && dil_w == 1) { auto k = cl_dwconv_wgrad2_kernel<float, 3, 1, TW>; DLKA_LAUNCH(k, grid2, block256(), 0, st, a); }
        else if (F32 && kw == 5 && dil_w == 3) { auto k = cl_dwconv_wgrad2_kernel<float, 5, 3, TW>; DLKA_LAUNCH(k, grid2, block256(), 0, st, a); }
        else if (F32 && kw == 7 && dil_w == 1) { auto k = cl_dwconv_wgrad2_kernel<float, 7, 1, TW>; DLKA_LAUNCH(k, grid2, block256(), 0, st, a); }
        else return DLKA_ERR_UNSUPPORTED;
        DLKA_CHECK_LAUNCH();
        return DLKA_OK;
    }
    dim3 grid(xb, a.kd * a.kh, cdiv(a.C, cpb)), block(256);
    if (kw == 5 && dil_w == 1) { auto k = cl_dwconv_wgrad_kernel<T, 5, 1, TW>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
    else if (kw == 7 && dil_w == 3) { auto k = cl_dwconv_wgrad_kernel<T, 7, 3, TW>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
    else if (F32 && kw == 3 && dil_w == 1) { auto k = cl_dwconv_wgrad_kernel<float, 3, 1, TW>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
    else if (F32 && kw == 5 && dil_w == 3) { auto k = cl_dwconv_wgrad_kernel<float, 5, 3, TW>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
    else if (F32 && kw == 7 && dil_w == 1) { auto k = cl_dwconv_wgrad_kernel<float, 7, 1, TW>; DLKA_LAUNCH(k, grid, block, 0, st, a); }
    else return DLKA_ERR_UNSUPPORTED;
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_dwconv_wgrad(DwWgradArgs a, int kw, int dil_w, hipStream_t st, bool zero_init)
{
    return a.act_bf16 ? launch_cl_dwconv_wgrad_t<bf16_t>(a, kw, dil_w, st, zero_init) : launch_cl_dwconv_wgrad_t<float>(a, kw, dil_w, st, zero_init);
}

// gWp[tap][c] -> reference layout gW[c][1][taps] in storage type T
template <typename T>
__global__ void cl_dw_unprep_kernel(const float *__restrict__ gwp, T *__restrict__ gw, int C, int K)
{
    const int n = C * K;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int tap = e % K, c = e / K;
        stf(gw + e, gwp[(long)tap * C + c]);
    }
}

template <typename T>
int launch_cl_dw_unprep(const float *gwp, T *gw, int C, int K, hipStream_t st)
{
    auto k = cl_dw_unprep_kernel<T>;
    DLKA_LAUNCH(k, dim3(cdiv(C * K, 256)), dim3(256), 0, st, gwp, gw, C, K);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}
template int launch_cl_dw_unprep<float>(const float *, float *, int, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
// layout changes between the reference's planar NCDHW and channels-last (used by the *_ndhwc test entry points and by
// LKA_Attention3d_deform.forward_volume; the token entry point needs none).
// ---------------------------------------------------------------------------------------------
template <int TO_CL, typename T>
__global__ __launch_bounds__(256) void cl_transpose_kernel(const T *__restrict__ src, T *__restrict__ dst, int C, int N)
{
    // tile 32 (voxels) x 32 (channels) through LDS; grid = (ceil(N/32), ceil(C/32), B)
    __shared__ float tile[32][33];
    const int b = blockIdx.z, n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const T *s = src + (long)b * C * N;
    T *d = dst + (long)b * C * N;
    if (TO_CL) {  // src [C][N] -> dst [N][C]
        for (int r = ty; r < 32; r += 8)
            if (c0 + r < C && n0 + tx < N) tile[r][tx] = act_load1(s, (long)(c0 + r) * N + n0 + tx);
        __syncthreads();
        for (int r = ty; r < 32; r += 8)
            if (n0 + r < N && c0 + tx < C) act_store1(d, (long)(n0 + r) * C + c0 + tx, tile[tx][r]);
    } else {      // src [N][C] -> dst [C][N]
        for (int r = ty; r < 32; r += 8)
            if (n0 + r < N && c0 + tx < C) tile[r][tx] = act_load1(s, (long)(n0 + r) * C + c0 + tx);
        __syncthreads();
        for (int r = ty; r < 32; r += 8)
            if (c0 + r < C && n0 + tx < N) act_store1(d, (long)(c0 + r) * N + n0 + tx, tile[tx][r]);
    }
}

// bf16 = 1: src / dst are bf16 storage (the values pass through unchanged)
int launch_cl_transpose(const float *src, float *dst, int B, int C, int N, int to_cl, hipStream_t st, int bf16)
{
    dim3 grid(cdiv(N, 32), cdiv(C, 32), B), block(256);
    if (bf16) {
        const bf16_t *s16 = reinterpret_cast<const bf16_t *>(src);
        bf16_t *d16 = reinterpret_cast<bf16_t *>(dst);
        if (to_cl) { auto k = cl_transpose_kernel<1, bf16_t>; DLKA_LAUNCH(k, grid, block, 0, st, s16, d16, C, N); }
        else { auto k = cl_transpose_kernel<0, bf16_t>; DLKA_LAUNCH(k, grid, block, 0, st, s16, d16, C, N); }
    }
    else if (to_cl) { auto k = cl_transpose_kernel<1, float>; DLKA_LAUNCH(k, grid, block, 0, st, src, dst, C, N); }
    else { auto k = cl_transpose_kernel<0, float>; DLKA_LAUNCH(k, grid, block, 0, st, src, dst, C, N); }
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

}  // namespace dlka
