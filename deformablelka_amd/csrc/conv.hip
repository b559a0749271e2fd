// General grouped N-d convolution (forward, data gradient, weight gradient) — the nn.Conv3d / nn.Conv2d calls
// that sit on the D-LKA path: depthwise 5^3 and 7^3-dilation-3
// (3D/d_lka_former/network_architecture/synapse/transformerblock.py:637-638), the dense offset-predict conv
// C->81 (synapse/deform_conv.py:80-85), the 1x1x1 projections (transformerblock.py:641,659,662) and the 2-D
// offset nets (2D/deformable_LKA/deformable_LKA.py:10-16).  In the reference these run in cuDNN.
//
// This file is the GENERAL direct path (any geometry).  One work-item per output voxel, COB output channels in
// registers, weights wave-uniform through the scalar cache, lanes consecutive along W (coalesced).
// Shape-specialised LDS-tiled / MFMA kernels for the hot shapes are in conv_tiled.hip.
#include "dlka_kernels.h"

namespace dlka {

static int pick_pow2_upto32(int n)
{
    if (n >= 32) return 32;
    int c = 1;
    while (c < n) c <<= 1;
    return c;
}

int conv_fwd_wt_floats(const Geom &g) { return g.group * g.K * g.Cg * round_up(g.Og, pick_pow2_upto32(g.Og)); }
int conv_bwd_wb_floats(const Geom &g) { return g.group * g.K * g.Og * round_up(g.Cg, pick_pow2_upto32(g.Cg)); }

// W[co][cg][tap] -> Wb[g][tap][o][CgP]  (for the data gradient: the Cg weights multiplying one grad_out value are contiguous)
template <typename T>
__global__ void relayout_weight_bwd_kernel(const T *__restrict__ w, float *__restrict__ wb, int group, int Og, int Cg, int K, int CgP)
{
    const int n = group * K * Og * CgP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int cg = i % CgP, o = (i / CgP) % Og, tap = (i / CgP / Og) % K, g = i / CgP / Og / K;
        wb[i] = (cg < Cg) ? ldf(w + ((long)(g * Og + o) * Cg + cg) * K + tap) : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// forward: grid = (voxel tiles, group * OgP/COB, B)
// ---------------------------------------------------------------------------------------------
template <typename T, int COB>
__global__ __launch_bounds__(DLKA_THREADS) void conv_fwd_kernel(
    const T *__restrict__ x, const float *__restrict__ wt, const T *__restrict__ bias, T *__restrict__ out, Geom g, int OgP)
{
    const int v = blockIdx.x * DLKA_THREADS + threadIdx.x;
    const int chunks = OgP / COB;
    const int gi = blockIdx.y / chunks, co0 = (blockIdx.y % chunks) * COB;
    const int b = blockIdx.z;
    if (v >= g.No) return;
    const int ow = v % g.Wo, oh = (v / g.Wo) % g.Ho, od = v / (g.Wo * g.Ho);
    const int bd = od * g.sd - g.pd, bh = oh * g.sh - g.ph, bw = ow * g.sw - g.pw;
    float acc[COB];
#pragma unroll
    for (int j = 0; j < COB; ++j) acc[j] = 0.f;
    const T *xg = x + (long)(b * g.C + gi * g.Cg) * g.Ni;
    int tap = 0;
    for (int i = 0; i < g.kd; ++i) {
        const int zd = bd + i * g.dd;
        for (int jx = 0; jx < g.kh; ++jx) {
            const int zh = bh + jx * g.dh;
            for (int k = 0; k < g.kw; ++k, ++tap) {
                const int zw = bw + k * g.dw;
                const bool ok = zd >= 0 && zd < g.D && zh >= 0 && zh < g.H && zw >= 0 && zw < g.W;
                const int lin = ok ? (zd * g.H + zh) * g.W + zw : 0;
                const float *wrow = wt + ((long)(gi * g.K + tap) * g.Cg) * OgP + co0;
                for (int cg = 0; cg < g.Cg; ++cg) {
                    const float xv = ok ? ldf(xg + (long)cg * g.Ni + lin) : 0.f;
                    const float *wp = wrow + (long)cg * OgP;
#pragma unroll
                    for (int j = 0; j < COB; ++j) acc[j] = fmaf(xv, wp[j], acc[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < COB; ++j) {
        const int o = co0 + j;
        if (o < g.Og) {
            const int co = gi * g.Og + o;
            stf(out + (long)(b * g.Cout + co) * g.No + v, acc[j] + (bias ? ldf(bias + co) : 0.f));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 3^3 / stride 1 / padding 1 convolution with <= 16 input and <= 16 output channels on planar fp32 tensors — the full net's full-resolution
// plumbing convs (encoder1 / decoder2: 2 x 16 x 64 x 128 x 128, d_lka_former_synapse.py:89-133) — on the matrix cores; serves the forward
// pass and, with the taps flipped and the channel roles exchanged (FLIP), the data gradient.  The thread-per-voxel kernels take 0.51 / 0.78 ms there.
//   out[i][v] = sum_tap sum_k A_tap[i][k] in[k][v + tap],   v_mfma_f32_16x16x4_f32: D[i = out channel][j = voxel], k = in channel (4 steps of 4)
// Lane (j = lane & 15, kg = lane >> 4): the voxel w0 + j of a (b, d, h) row and input channels 4 s + kg.  Per (tap_d, tap_h) and k-step three dword
// loads (left, centre, right neighbour along w: the three tap_w operands); the 27 x 4 weight values a lane needs as A operand (row i = lane & 15,
// k = 4 s + kg) are taken from the weight tensor once and stay in registers.  A wave walks a run of rows.
// ---------------------------------------------------------------------------------------------
template <bool FLIP>
__global__ __launch_bounds__(256, 2) void conv3_mfma_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                                                           float *__restrict__ out, Geom g, int rows_per_wave)
{
    const int lane = threadIdx.x & 63, j = lane & 15, kg = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int CI = FLIP ? g.Cout : g.C, CO = FLIP ? g.C : g.Cout;   // channels of `in` / `out` of THIS op
    // A operand: forward A_tap[i = co][k = ci] = W[co][ci][tap]; data gradient A_tap[i = ci][k = co] = W[co][ci][26 - tap]
    float areg[27][4];
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int i = j, k = 4 * s4 + kg;
            float v = 0.f;
            if (i < CO && k < CI) v = FLIP ? w[((long)k * g.C + i) * 27 + (26 - t)] : w[((long)i * g.C + k) * 27 + t];
            areg[t][s4] = v;
        }
    const long nrows = (long)g.B * g.D * g.H;
    const long r0 = (long)wave * rows_per_wave, r1 = r0 + rows_per_wave < nrows ? r0 + rows_per_wave : nrows;
    const long plane = (long)g.D * g.H * g.W;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (!FLIP && bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = 4 * kg + r < CO ? bias[4 * kg + r] : 0.f;
    }
    for (long r = r0; r < r1; ++r) {
        const int h = (int)(r % g.H), d = (int)((r / g.H) % g.D), b = (int)(r / ((long)g.H * g.D));
        for (int w0 = 0; w0 < g.W; w0 += 16) {   // wave-uniform trip count (the MFMAs are wave-wide)
            const int wx = w0 + j;
            const bool in_c = wx < g.W, in_l = in_c && wx > 0, in_r = wx + 1 < g.W;
            f32x4 acc = {bv[0], bv[1], bv[2], bv[3]};
#pragma unroll
            for (int td = 0; td < 3; ++td) {
                const int zd = d + td - 1;
                if (zd < 0 || zd >= g.D) continue;   // uniform
#pragma unroll
                for (int th = 0; th < 3; ++th) {
                    const int zh = h + th - 1;
                    if (zh < 0 || zh >= g.H) continue;   // uniform
                    const int t0 = (td * 3 + th) * 3;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int k = 4 * s4 + kg;
                        const bool kok = k < CI;
                        const float *xr = in + (((long)b * CI + (kok ? k : 0)) * g.D + zd) * (long)g.H * g.W + (long)zh * g.W;
                        float l = xr[in_l ? wx - 1 : 0], c = xr[in_c ? wx : 0], rr = xr[in_r ? wx + 1 : 0];
                        if (!kok || !in_l) l = 0.f;
                        if (!kok || !in_c) c = 0.f;
                        if (!kok || !in_r) rr = 0.f;
                        acc = mfma_16x16x4(areg[t0][s4], l, acc);
                        acc = mfma_16x16x4(areg[t0 + 1][s4], c, acc);
                        acc = mfma_16x16x4(areg[t0 + 2][s4], rr, acc);
                    }
                }
            }
            if (in_c) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int i = 4 * kg + r4;
                    if (i < CO) out[((long)b * CO + i) * plane + ((long)d * g.H + h) * g.W + wx] = acc[r4];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same op for rows of W <= 128 voxels (W % 8 == 0: the net's 128-wide full-resolution rows), with the row held in registers: lane (j, kg) owns the
// EIGHT consecutive voxels 8 j .. 8 j + 7 of the row (two 16-byte loads per input row and k-step) and the wave runs eight 16x16x4 tiles, tile q = the voxels
// {8 j + q}.  The tap_w = -1 / +1 operands of tile q are the registers of tile q - 1 / q + 1 of the SAME lane; only tile 0's left and tile 7's right neighbour
// come from the adjacent lane — one DPP row shift each, whose out-of-row zero IS the padding.  Two 16-byte loads and two DPP moves feed 24 MFMAs (the kernel
// above: 24 dword loads); loads run one k-step ahead of the MFMAs; rows outside the volume are loaded as zeros (buffer range check), so the 36 steps of a row are
// straight-line code.  Per row 864 MFMAs = 27.6 k cycles of the SIMD's matrix pipe: 184 us for 2 x 16 x 64 x 128 x 128 with every pipe busy.
// ---------------------------------------------------------------------------------------------
template <bool FLIP>
__global__ __launch_bounds__(256, 2) void conv3_row_mfma_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                                                               float *__restrict__ out, Geom g, int rows_per_wave)
{
    const int lane = threadIdx.x & 63, j = lane & 15, kg = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int CI = FLIP ? g.Cout : g.C, CO = FLIP ? g.C : g.Cout;
    float areg[27][4];   // as conv3_mfma_kernel
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int i = j, k = 4 * s4 + kg;
            float v = 0.f;
            if (i < CO && k < CI) v = FLIP ? w[((long)k * g.C + i) * 27 + (26 - t)] : w[((long)i * g.C + k) * 27 + t];
            areg[t][s4] = v;
        }
    const long nrows = (long)g.B * g.D * g.H;
    const long r0 = (long)wave * rows_per_wave, r1 = r0 + rows_per_wave < nrows ? r0 + rows_per_wave : nrows;
    const unsigned plane = (unsigned)(g.D * g.H * g.W);
    const bool jok = 8 * j < g.W;
    const BufRsrc rin = make_rsrc(in, (size_t)g.B * CI * plane * 4), rout = make_rsrc(out, (size_t)g.B * CO * plane * 4);
    unsigned lane_off[4];   // channel plane + position in the row, bytes
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) lane_off[s4] = (jok && 4 * s4 + kg < CI) ? ((unsigned)(4 * s4 + kg) * plane + 8u * j) * 4u : DLKA_OOB;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (!FLIP && bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = 4 * kg + r < CO ? bias[4 * kg + r] : 0.f;
    }
    for (long r = r0; r < r1; ++r) {
        const int h = (int)(r % g.H), d = (int)((r / g.H) % g.D), b = (int)(r / ((long)g.H * g.D));
        const unsigned in_b = (unsigned)b * (unsigned)CI * plane * 4u;
        f32x4 acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = f32x4{bv[0], bv[1], bv[2], bv[3]};
        f32x4 lo, hi, nlo, nhi;
        auto request = [&](int step, f32x4 &a, f32x4 &c) {
            const int td = step / 12, th = (step / 4) % 3, s4 = step & 3;
            const int zd = d + td - 1, zh = h + th - 1;
            const bool ok = zd >= 0 && zd < g.D && zh >= 0 && zh < g.H;   // uniform
            const unsigned off = ok ? lane_off[s4] + in_b + (unsigned)((zd * g.H + zh) * g.W) * 4u : DLKA_OOB;
            a = buf_load_f32x4(rin, off);
            c = buf_load_f32x4(rin, off + 16u);
        };
        request(0, lo, hi);
#pragma unroll
        for (int step = 0; step < 36; ++step) {
#ifndef DLKA_ABLR   // -DDLKA_ABLR=bits: TIMING-ONLY ablations (wrong results): 1 no operand loads in the loop, 2 no MFMAs in the loop
#define DLKA_ABLR 0
#endif
            if (step + 1 < 36 && !(DLKA_ABLR & 1)) request(step + 1, nlo, nhi);
            const int t0 = (step / 4) * 3, s4 = step & 3;
            if (DLKA_ABLR & 2) { acc[0][0] += lo[0] + hi[3]; lo = nlo; hi = nhi; continue; }
            const float xv[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const float el = row_shr1(xv[7]), er = row_shl1(xv[0]);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = mfma_16x16x4(areg[t0][s4], q ? xv[q ? q - 1 : 0] : el, acc[q]);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = mfma_16x16x4(areg[t0 + 1][s4], xv[q], acc[q]);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = mfma_16x16x4(areg[t0 + 2][s4], q < 7 ? xv[q < 7 ? q + 1 : 7] : er, acc[q]);
            lo = nlo; hi = nhi;
        }
        const unsigned row = (unsigned)((d * g.H + h) * g.W);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int i = 4 * kg + r4;
            const unsigned off = (jok && i < CO) ? (((unsigned)b * CO + i) * plane + row + 8u * j) * 4u : DLKA_OOB;
            buf_store_f32x4(rout, off, f32x4{acc[0][r4], acc[1][r4], acc[2][r4], acc[3][r4]});
            buf_store_f32x4(rout, off == DLKA_OOB ? DLKA_OOB : off + 16u, f32x4{acc[4][r4], acc[5][r4], acc[6][r4], acc[7][r4]});
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Round 6 — the DATA gradient of the same conv on the bf16 matrix cores with both operands as two bf16 terms (the rule the D-LKA block's own gradients follow, DESIGN 4.1: ~1e-5
// against the 1e-3 gradient contract; the forward pass keeps the fp32-input MFMAs above).  Timing-only ablations of conv3_row_mfma_kernel at 2 x 16 x 64 x 128 x 128 said it is
// bound by its 864 fp32 MFMAs per row (288 us as built, 295 without the loads, 162 with the loads alone).  Here a lane's FOUR contraction channels 4 s + kg are half of the eight
// k-values it supplies to v_mfma_f32_16x16x32_bf16 and the other half carries a second term:  A = [w_hi | w_lo] (one 16-byte operand per tap, built once),
// B1 = [g_hi | g_hi], B2 = [g_lo | g_lo]:  A B1 + A B2 = (w_hi + w_lo)(g_hi + g_lo) — two instructions of 16 cycles per (tile, tap) instead of four of 32.  The B operands of
// position p (voxel 8 j + p, p = -1 .. 8) are built when the walk reaches it and serve the tiles p + 1, p, p - 1 (tap_w = 0, 1, 2).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void conv3_row_dgrad_b16_kernel(const float *__restrict__ in, const float *__restrict__ w, float *__restrict__ out, Geom g, int rows_per_wave)
{
    const int lane = threadIdx.x & 63, j = lane & 15, kg = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int CI = g.Cout, CO = g.C;   // (the data gradient contracts the conv's OUTPUT channels)
    bf16x8 areg[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        float v[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int i = j, k = 4 * s4 + kg;
            v[s4] = (i < CO && k < CI) ? w[((long)k * g.C + i) * 27 + (26 - t)] : 0.f;
        }
        // [w_hi | w_lo]: the high terms of (w, w - bf16(w))
        const float a8[8] = {v[0], v[1], v[2], v[3], v[0] - bf16_value(bf16_bits(v[0])), v[1] - bf16_value(bf16_bits(v[1])), v[2] - bf16_value(bf16_bits(v[2])),
                             v[3] - bf16_value(bf16_bits(v[3]))};
        bf16x8 unused;
        split_bf16x8(a8, areg[t], unused);
    }
    const long nrows = (long)g.B * g.D * g.H;
    const long r0 = (long)wave * rows_per_wave, r1 = r0 + rows_per_wave < nrows ? r0 + rows_per_wave : nrows;
    const unsigned plane = (unsigned)(g.D * g.H * g.W);
    const bool jok = 8 * j < g.W;
    const BufRsrc rin = make_rsrc(in, (size_t)g.B * CI * plane * 4), rout = make_rsrc(out, (size_t)g.B * CO * plane * 4);
    unsigned lane_off[4];   // channel plane + position in the row, bytes
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) lane_off[s4] = (jok && 4 * s4 + kg < CI) ? ((unsigned)(4 * s4 + kg) * plane + 8u * j) * 4u : DLKA_OOB;
    struct Raw { f32x4 lo[4], hi[4]; };   // the lane's eight voxels of its four channels
    for (long r = r0; r < r1; ++r) {
        const int h = (int)(r % g.H), d = (int)((r / g.H) % g.D), b = (int)(r / ((long)g.H * g.D));
        const unsigned in_b = (unsigned)b * (unsigned)CI * plane * 4u;
        f32x4 acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto request = [&](int step, Raw &o) {
            const int td = step / 3, th = step % 3;
            const int zd = d + td - 1, zh = h + th - 1;
            const bool ok = zd >= 0 && zd < g.D && zh >= 0 && zh < g.H;   // uniform
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const unsigned off = (ok && lane_off[s4] != DLKA_OOB) ? lane_off[s4] + in_b + (unsigned)((zd * g.H + zh) * g.W) * 4u : DLKA_OOB;
                o.lo[s4] = buf_load_f32x4(rin, off);
                o.hi[s4] = buf_load_f32x4(rin, off == DLKA_OOB ? DLKA_OOB : off + 16u);
            }
        };
        Raw cur, nxt;
        request(0, cur);
#pragma unroll
        for (int step = 0; step < 9; ++step) {
            if (step + 1 < 9) request(step + 1, nxt);
            const int t0 = step * 3;
            float el[4], er[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) { el[s4] = row_shr1(cur.hi[s4][3]); er[s4] = row_shl1(cur.lo[s4][0]); }
#pragma unroll
            for (int p = -1; p <= 8; ++p) {
                float v8[8];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const float v = p < 0 ? el[s4] : (p > 7 ? er[s4] : (p < 4 ? cur.lo[s4][p < 4 && p >= 0 ? p : 0] : cur.hi[s4][p >= 4 && p < 8 ? p - 4 : 0]));
                    v8[s4] = v; v8[4 + s4] = v;
                }
                bf16x8 b1, b2;   // [g_hi | g_hi], [g_lo | g_lo]  (A b2 = w_hi g_lo + w_lo g_lo: the fourth term of the product rides along)
                split_bf16x8(v8, b1, b2);
#pragma unroll
                for (int tw = 0; tw < 3; ++tw) {
                    const int q = p + 1 - tw;
                    if (q < 0 || q > 7) continue;
                    acc[q] = mfma_16x16x32_bf16(areg[t0 + tw], b1, acc[q]);
                    acc[q] = mfma_16x16x32_bf16(areg[t0 + tw], b2, acc[q]);
                }
            }
            cur = nxt;
        }
        const unsigned row = (unsigned)((d * g.H + h) * g.W);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int i = 4 * kg + r4;
            const unsigned off = (jok && i < CO) ? (((unsigned)b * CO + i) * plane + row + 8u * j) * 4u : DLKA_OOB;
            buf_store_f32x4(rout, off, f32x4{acc[0][r4], acc[1][r4], acc[2][r4], acc[3][r4]});
            buf_store_f32x4(rout, off == DLKA_OOB ? DLKA_OOB : off + 16u, f32x4{acc[4][r4], acc[5][r4], acc[6][r4], acc[7][r4]});
        }
    }
}

static bool conv3_mfma_shape(const Geom &g)
{
    return g.group == 1 && g.kd == 3 && g.kh == 3 && g.kw == 3 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 1 && g.ph == 1 && g.pw == 1 && g.dd == 1 &&
           g.dh == 1 && g.dw == 1 && g.C <= 16 && g.Cout <= 16;
}
// rows in registers: W <= 128 in whole groups of 8, at least 8 contraction channels (fewer: three quarters of the MFMA k-steps would multiply zeros), and
// 32-bit byte offsets in both tensors
static bool conv3_row_mfma_shape(const Geom &g, bool flip)
{
    const int CI = flip ? g.Cout : g.C;
    const size_t big = (size_t)g.B * 16 * g.D * g.H * g.W * 4;
    return conv3_mfma_shape(g) && g.W % 8 == 0 && g.W <= 128 && CI >= 8 && big < ((size_t)1 << 31);
}
static int conv3_mfma_rows_per_wave(const Geom &g, long &waves)
{
    const long nrows = (long)g.B * g.D * g.H;
    int rpw = (int)cdivl(nrows, 2048);   // ~2 waves per SIMD over the chip
    if (rpw < 1) rpw = 1;
    waves = cdivl(nrows, rpw);
    return rpw;
}

template <typename T>
int launch_conv_fwd(const T *x, const T *w, const T *bias, T *out, float *wt, const Geom &g, hipStream_t st)
{
    // (conv3_mfma_kernel<false> serves this shape too, but measured SLOWER than the kernel above at 2 x 16 x 64x128x128: 627 vs 513 us — per 16 voxels it
    //  issues 108 dword loads with their address arithmetic for 108 MFMAs; the data gradient, whose thread-per-voxel form is slower, gains: 617 vs 776 us)
    if constexpr (sizeof(T) == 4) {
        if (conv3_row_mfma_shape(g, false)) {
            long waves;
            const int rpw = conv3_mfma_rows_per_wave(g, waves);
            DLKA_LAUNCH(conv3_row_mfma_kernel<false>, dim3((unsigned)cdivl(waves, 4)), dim3(256), 0, st, reinterpret_cast<const float *>(x),
                        reinterpret_cast<const float *>(w), reinterpret_cast<const float *>(bias), reinterpret_cast<float *>(out), g, rpw);
            DLKA_CHECK_LAUNCH();
            return DLKA_OK;
        }
    }
    const int cob = pick_pow2_upto32(g.Og);
    const int OgP = round_up(g.Og, cob);
    int rc = launch_relayout_weight<T>(w, wt, g.group, g.Og, g.Cg, g.K, OgP, st);
    if (rc) return rc;
    dim3 grid(cdiv(g.No, DLKA_THREADS), g.group * (OgP / cob), g.B), block(DLKA_THREADS);
#define DLKA_L(COB)                                                                    \
    {                                                                                  \
        auto k = conv_fwd_kernel<T, COB>;                                              \
        DLKA_LAUNCH(k, grid, block, 0, st, x, (const float *)wt, bias, out, g, OgP); \
    }
    switch (cob) {
        case 1: DLKA_L(1) break;
        case 2: DLKA_L(2) break;
        case 4: DLKA_L(4) break;
        case 8: DLKA_L(8) break;
        case 16: DLKA_L(16) break;
        default: DLKA_L(32) break;
    }
#undef DLKA_L
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// ---------------------------------------------------------------------------------------------
// data gradient (gather form, any stride): one work-item per input voxel, CIB input channels in registers
//   gX[b][c][z] = sum_{o, tap : (z + p - tap*dil) % s == 0} gO[b][o][(z + p - tap*dil)/s] * W[o][c][tap]
// grid = (input voxel tiles, group * CgP/CIB, B)
// ---------------------------------------------------------------------------------------------
template <typename T, int CIB>
__global__ __launch_bounds__(DLKA_THREADS) void conv_bwd_data_kernel(
    const T *__restrict__ gout, const float *__restrict__ wb, T *__restrict__ gx, Geom g, int CgP)
{
    const int z = blockIdx.x * DLKA_THREADS + threadIdx.x;
    const int chunks = CgP / CIB;
    const int gi = blockIdx.y / chunks, ci0 = (blockIdx.y % chunks) * CIB;
    const int b = blockIdx.z;
    if (z >= g.Ni) return;
    const int zw = z % g.W, zh = (z / g.W) % g.H, zd = z / (g.W * g.H);
    float acc[CIB];
#pragma unroll
    for (int j = 0; j < CIB; ++j) acc[j] = 0.f;
    const T *gg = gout + (long)(b * g.Cout + gi * g.Og) * g.No;
    int tap = 0;
    for (int i = 0; i < g.kd; ++i) {
        const int nd = zd + g.pd - i * g.dd;
        const int od = nd / g.sd;
        const bool okd = nd >= 0 && (nd - od * g.sd) == 0 && od < g.Do;
        for (int jx = 0; jx < g.kh; ++jx) {
            const int nh = zh + g.ph - jx * g.dh;
            const int oh = nh / g.sh;
            const bool okh = nh >= 0 && (nh - oh * g.sh) == 0 && oh < g.Ho;
            for (int k = 0; k < g.kw; ++k, ++tap) {
                const int nw = zw + g.pw - k * g.dw;
                const int ow = nw / g.sw;
                const bool ok = okd && okh && nw >= 0 && (nw - ow * g.sw) == 0 && ow < g.Wo;
                const int lin = ok ? (od * g.Ho + oh) * g.Wo + ow : 0;
                const float *wrow = wb + ((long)(gi * g.K + tap) * g.Og) * CgP + ci0;
                for (int o = 0; o < g.Og; ++o) {
                    const float gv = ok ? ldf(gg + (long)o * g.No + lin) : 0.f;
                    const float *wp = wrow + (long)o * CgP;
#pragma unroll
                    for (int j = 0; j < CIB; ++j) acc[j] = fmaf(gv, wp[j], acc[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CIB; ++j) {
        const int cg = ci0 + j;
        if (cg < g.Cg) stf(gx + (long)(b * g.C + gi * g.Cg + cg) * g.Ni + z, acc[j]);
    }
}

template <typename T>
int launch_conv_bwd_data(const T *gout, const T *w, T *gx, float *wb, const Geom &g, hipStream_t st)
{
    if constexpr (sizeof(T) == 4) {
        if (conv3_row_mfma_shape(g, true)) {
            long waves;
            const int rpw = conv3_mfma_rows_per_wave(g, waves);
            static const bool exact = getenv("DLKA_EXACT_FP32") != nullptr;
            static const bool b16_off = [] { const char *e = getenv("DLKA_CONV3_DGRAD_B16"); return e && e[0] == '0'; }();   // (A/B: 0 = fp32-input MFMAs)
            if (!exact && !b16_off) {
                DLKA_LAUNCH(conv3_row_dgrad_b16_kernel, dim3((unsigned)cdivl(waves, 4)), dim3(256), 0, st, reinterpret_cast<const float *>(gout),
                            reinterpret_cast<const float *>(w), reinterpret_cast<float *>(gx), g, rpw);
            } else {
                DLKA_LAUNCH(conv3_row_mfma_kernel<true>, dim3((unsigned)cdivl(waves, 4)), dim3(256), 0, st, reinterpret_cast<const float *>(gout),
                            reinterpret_cast<const float *>(w), (const float *)nullptr, reinterpret_cast<float *>(gx), g, rpw);
            }
            DLKA_CHECK_LAUNCH();
            return DLKA_OK;
        }
        if (conv3_mfma_shape(g) && g.Cout >= 4) {
            long waves;
            const int rpw = conv3_mfma_rows_per_wave(g, waves);
            DLKA_LAUNCH(conv3_mfma_kernel<true>, dim3((unsigned)cdivl(waves, 4)), dim3(256), 0, st, reinterpret_cast<const float *>(gout),
                        reinterpret_cast<const float *>(w), (const float *)nullptr, reinterpret_cast<float *>(gx), g, rpw);
            DLKA_CHECK_LAUNCH();
            return DLKA_OK;
        }
    }
    const int cib = pick_pow2_upto32(g.Cg);
    const int CgP = round_up(g.Cg, cib);
    {
        const int n = g.group * g.K * g.Og * CgP;
        auto k = relayout_weight_bwd_kernel<T>;
        DLKA_LAUNCH(k, dim3(cdiv(n, 256)), dim3(256), 0, st, w, wb, g.group, g.Og, g.Cg, g.K, CgP);
        DLKA_CHECK_LAUNCH();
    }
    dim3 grid(cdiv(g.Ni, DLKA_THREADS), g.group * (CgP / cib), g.B), block(DLKA_THREADS);
#define DLKA_L(CIB)                                                               \
    {                                                                             \
        auto k = conv_bwd_data_kernel<T, CIB>;                                    \
        DLKA_LAUNCH(k, grid, block, 0, st, gout, (const float *)wb, gx, g, CgP); \
    }
    switch (cib) {
        case 1: DLKA_L(1) break;
        case 2: DLKA_L(2) break;
        case 4: DLKA_L(4) break;
        case 8: DLKA_L(8) break;
        case 16: DLKA_L(16) break;
        default: DLKA_L(32) break;
    }
#undef DLKA_L
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// ---------------------------------------------------------------------------------------------
// weight gradient: gW[co][cg][tap] = sum_{b,v} gO[b][co][v] * x[b][c][v*s - p + tap*dil]
// Same decomposition as deform_bwd_weight_kernel: block = (c, TPC taps, COB out-channels, voxel split).
// grid = (C, tapchunks * cochunks, VS); gw32 is fp32, zero-initialised, accumulated with atomics.
// ---------------------------------------------------------------------------------------------
template <typename T, int TPC, int COB>
__global__ __launch_bounds__(DLKA_THREADS) void conv_bwd_weight_kernel(
    const T *__restrict__ x, const T *__restrict__ gout, float *__restrict__ gw, Geom g, int cochunks)
{
    const int c = blockIdx.x;
    const int tchunk = blockIdx.y / cochunks, cchunk = blockIdx.y % cochunks;
    const int tap0 = tchunk * TPC, co0 = cchunk * COB;
    const int gi = c / g.Cg, cg = c - gi * g.Cg;
    const int VS = gridDim.z;
    float acc[TPC][COB];
#pragma unroll
    for (int t = 0; t < TPC; ++t)
#pragma unroll
        for (int j = 0; j < COB; ++j) acc[t][j] = 0.f;
    int ti[TPC], tj[TPC], tk[TPC];
#pragma unroll
    for (int t = 0; t < TPC; ++t) {
        const int tap = tap0 + t;
        tk[t] = (tap % g.kw) * g.dw; tj[t] = ((tap / g.kw) % g.kh) * g.dh; ti[t] = (tap / (g.kw * g.kh)) * g.dd;
    }
    const long total = (long)g.B * g.No;
    for (long n = (long)blockIdx.z * DLKA_THREADS + threadIdx.x; n < total; n += (long)DLKA_THREADS * VS) {
        const int b = (int)(n / g.No), v = (int)(n - (long)b * g.No);
        const int ow = v % g.Wo, oh = (v / g.Wo) % g.Ho, od = v / (g.Wo * g.Ho);
        const int bd = od * g.sd - g.pd, bh = oh * g.sh - g.ph, bw = ow * g.sw - g.pw;
        float G[COB];
#pragma unroll
        for (int j = 0; j < COB; ++j)
            G[j] = (co0 + j < g.Og) ? ldf(gout + (long)(b * g.Cout + gi * g.Og + co0 + j) * g.No + v) : 0.f;
        const T *xp = x + (long)(b * g.C + c) * g.Ni;
#pragma unroll
        for (int t = 0; t < TPC; ++t) {
            const int zd = bd + ti[t], zh = bh + tj[t], zw = bw + tk[t];
            const bool ok = (tap0 + t < g.K) && zd >= 0 && zd < g.D && zh >= 0 && zh < g.H && zw >= 0 && zw < g.W;
            const float xv = ok ? ldf(xp + (zd * g.H + zh) * g.W + zw) : 0.f;
#pragma unroll
            for (int j = 0; j < COB; ++j) acc[t][j] = fmaf(xv, G[j], acc[t][j]);
        }
    }
    __shared__ float red[DLKA_THREADS / 64][TPC * COB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < TPC; ++t)
#pragma unroll
        for (int j = 0; j < COB; ++j) {
            float a = acc[t][j];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m);
            if (lane == 0) red[wave][t * COB + j] = a;
        }
    __syncthreads();
    if (threadIdx.x < TPC * COB) {
        const int t = threadIdx.x / COB, j = threadIdx.x % COB;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < DLKA_THREADS / 64; ++w) a += red[w][threadIdx.x];
        if (tap0 + t < g.K && co0 + j < g.Og)
            atomicAdd(gw + ((long)(gi * g.Og + co0 + j) * g.Cg + cg) * g.K + tap0 + t, a);
    }
}

// Epilogue of the MFMA weight-gradient kernels: the four waves of the workgroup fold their 27 tiles through LDS, one atomic per element and WORKGROUP.
// With `part` the workgroup's folded tile goes to part[blockIdx.x][27][64][4] as plain 16-byte stores and conv3_wgrad_reduce_kernel adds the tiles up: 512
// workgroups x 6912 atomics on 6912 addresses took ~400 us of the 580 us kernel at 2 x 16 x 64 x 128 x 128.
__device__ __forceinline__ void conv3_wgrad_fold_and_add(f32x4 (&acc)[27], float *__restrict__ gw, const Geom &g, int lane, int i, int kg, bool ci_ok,
                                                         float *__restrict__ part = nullptr)
{
    // the four waves of the workgroup fold their tiles through LDS first (waves 2, 3 -> 0, 1; then 1 -> 0): every output element is hit by one
    // atomic per WORKGROUP — with one per wave, 2048 waves queued on the same 6912 addresses and the atomics were most of the kernel's time
    __shared__ __attribute__((aligned(16))) float red[2][27 * 64 * 4];
    const int wv = threadIdx.x >> 6;
    if (wv >= 2) {
#pragma unroll
        for (int t = 0; t < 27; ++t) *reinterpret_cast<f32x4 *>(&red[wv - 2][(t * 64 + lane) * 4]) = acc[t];
    }
    __syncthreads();
    if (wv < 2) {
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const f32x4 o = *reinterpret_cast<const f32x4 *>(&red[wv][(t * 64 + lane) * 4]);
            acc[t][0] += o[0]; acc[t][1] += o[1]; acc[t][2] += o[2]; acc[t][3] += o[3];
        }
    }
    __syncthreads();
    if (wv == 1) {
#pragma unroll
        for (int t = 0; t < 27; ++t) *reinterpret_cast<f32x4 *>(&red[0][(t * 64 + lane) * 4]) = acc[t];
    }
    __syncthreads();
    if (wv != 0) return;
    if (part) {
        float *dst = part + (size_t)blockIdx.x * (27 * 256);
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const f32x4 o = *reinterpret_cast<const f32x4 *>(&red[0][(t * 64 + lane) * 4]);
            *reinterpret_cast<f32x4 *>(dst + (t * 64 + lane) * 4) = f32x4{acc[t][0] + o[0], acc[t][1] + o[1], acc[t][2] + o[2], acc[t][3] + o[3]};
        }
        return;
    }
    // D layout: column j = lane & 15 = ci, rows 4 kg + r = co
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        const f32x4 o = *reinterpret_cast<const f32x4 *>(&red[0][(t * 64 + lane) * 4]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 4 * kg + r;
            if (ci_ok && co < g.Cout) atomicAdd(gw + ((long)co * g.C + i) * 27 + t, acc[t][r] + o[r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same weight gradient for the full net's full-resolution plumbing convs (3^3, stride 1, padding 1, <= 16 -> <= 16 channels at
// 2 x 64 x 128 x 128: encoder1 / decoder2, d_lka_former_synapse.py:89-133) on the matrix cores.  The thread-per-voxel kernel above needs 1.6 ms per
// conv there (18 TFLOP/s; 15 % of a trainer iteration).  Here the VOXEL axis is the contraction of v_mfma_f32_16x16x4_f32: per tap,
// A[i = co][k = voxel] = grad_out, B[k = voxel][j = ci] = x shifted by the tap.  Lane (i, kg = lane >> 4) owns voxels w0 + 4 kg .. + 3 of a (b, d, h) row:
// one 16-byte load of its grad_out row, and per (tap_d, tap_h) one 16-byte load of its x row plus the two neighbours x[w - 1], x[w + 4] — the three
// tap_w shifts are formed in registers.  27 accumulators (one 16 x 16 tile per tap) stay in registers over the wave's rows and leave as one atomic
// per element at the end (gw is zero-initialised by the caller, as for the kernel above).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void conv3_bwd_weight_mfma_kernel(const float *__restrict__ x, const float *__restrict__ gout, float *__restrict__ gw, Geom g,
                                                                      int rows_per_wave)
{
    const int lane = threadIdx.x & 63, i = lane & 15, kg = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nrows = (long)g.B * g.D * g.H;
    const long r0 = (long)wave * rows_per_wave, r1 = r0 + rows_per_wave < nrows ? r0 + rows_per_wave : nrows;
    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool co_ok = i < g.Cout, ci_ok = i < g.C;
    const long plane = (long)g.D * g.H * g.W;
    for (long r = r0; r < r1; ++r) {
        const int h = (int)(r % g.H), d = (int)((r / g.H) % g.D), b = (int)(r / ((long)g.H * g.D));
        const float *grow = gout + ((long)b * g.Cout + (co_ok ? i : 0)) * plane + ((long)d * g.H + h) * g.W;
        const float *xbase = x + ((long)b * g.C + (ci_ok ? i : 0)) * plane;
        for (int w0 = 0; w0 < g.W; w0 += 16) {   // wave-uniform trip count (the MFMAs are wave-wide)
            const int w = w0 + 4 * kg;
            const bool in = w < g.W;              // (W % 4 == 0: a lane's four voxels are inside the row or all outside)
            f32x4 q = *reinterpret_cast<const f32x4 *>(grow + (in ? w : 0));
            if (!in || !co_ok) q = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int td = 0; td < 3; ++td) {
                const int zd = d + td - 1;
                if (zd < 0 || zd >= g.D) continue;   // uniform
#pragma unroll
                for (int th = 0; th < 3; ++th) {
                    const int zh = h + th - 1;
                    if (zh < 0 || zh >= g.H) continue;   // uniform
                    const float *xr = xbase + ((long)zd * g.H + zh) * g.W;
                    f32x4 c = *reinterpret_cast<const f32x4 *>(xr + (in ? w : 0));
                    float lft = xr[(in && w > 0) ? w - 1 : 0], rgt = xr[(in && w + 4 < g.W) ? w + 4 : 0];
                    if (!in || !ci_ok) { c = f32x4{0.f, 0.f, 0.f, 0.f}; lft = 0.f; rgt = 0.f; }
                    if (w == 0) lft = 0.f;
                    if (w + 4 >= g.W) rgt = 0.f;
                    const int t0 = (td * 3 + th) * 3;
                    // tap_w = 0: x[w - 1 + s], 1: x[w + s], 2: x[w + 1 + s]
                    acc[t0] = mfma_16x16x4(q[0], lft, acc[t0]); acc[t0] = mfma_16x16x4(q[1], c[0], acc[t0]);
                    acc[t0] = mfma_16x16x4(q[2], c[1], acc[t0]); acc[t0] = mfma_16x16x4(q[3], c[2], acc[t0]);
                    acc[t0 + 1] = mfma_16x16x4(q[0], c[0], acc[t0 + 1]); acc[t0 + 1] = mfma_16x16x4(q[1], c[1], acc[t0 + 1]);
                    acc[t0 + 1] = mfma_16x16x4(q[2], c[2], acc[t0 + 1]); acc[t0 + 1] = mfma_16x16x4(q[3], c[3], acc[t0 + 1]);
                    acc[t0 + 2] = mfma_16x16x4(q[0], c[1], acc[t0 + 2]); acc[t0 + 2] = mfma_16x16x4(q[1], c[2], acc[t0 + 2]);
                    acc[t0 + 2] = mfma_16x16x4(q[2], c[3], acc[t0 + 2]); acc[t0 + 2] = mfma_16x16x4(q[3], rgt, acc[t0 + 2]);
                }
            }
        }
    }
    conv3_wgrad_fold_and_add(acc, gw, g, lane, i, kg, ci_ok);
}

// part[nwg][27][64][4] -> gw: grid (27, splits); thread = (lane, r) of tap blockIdx.x, sums its share of the workgroup tiles, one atomic per thread
__global__ __launch_bounds__(256) void conv3_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ gw, int nwg, int C, int Cout)
{
    const int t = blockIdx.x, e = threadIdx.x, lane = e >> 2, r = e & 3, i = lane & 15, kg = lane >> 4;
    const int per = (nwg + gridDim.y - 1) / gridDim.y, lo = blockIdx.y * per, hi = lo + per < nwg ? lo + per : nwg;
    float a = 0.f;
    for (int wg = lo; wg < hi; ++wg) a += part[(size_t)wg * (27 * 256) + t * 256 + e];
    const int co = 4 * kg + r;
    if (lo < hi && i < C && co < Cout) atomicAdd(gw + ((long)co * C + i) * 27 + t, a);
}

// ---------------------------------------------------------------------------------------------
// The same weight gradient with 32-voxel segments (lane (i, kg) owns the EIGHT voxels 32 seg + 8 kg .. + 7: two 16-byte loads + the two neighbours per
// (tap_d, tap_h) feed 24 MFMAs) and straight-line code: rows outside the volume are loaded as zeros through the buffer range check instead of being
// branched around, and every step's loads are issued one step ahead of its MFMAs — across segments and rows too (the kernel above waits for its loads
// at every (tap_d, tap_h): 730 us at 2 x 16 x 64 x 128 x 128 for 184 us of matrix-pipe time).  W % 8 == 0, byte offsets < 2^31.
// ---------------------------------------------------------------------------------------------
// B16 (round 6, default): the contraction on the bf16 matrix cores with both operands as two bf16 terms — a lane's EIGHT voxels are exactly the eight k-values it supplies to
// v_mfma_f32_16x16x32_bf16, so one instruction contracts the whole 32-voxel segment: gout_hi x_hi + gout_lo x_hi + gout_hi x_lo = 3 instructions of 16 cycles per tap where the
// fp32 form needs 8 of 32 (each product exact in the fp32 accumulator; the dropped lo x lo term is 2^-18 of the product: ~1e-5 on the gradient, the rule the D-LKA block's own
// weight gradients follow — DESIGN 4.1).  DLKA_EXACT_FP32 keeps the fp32-input MFMAs.
template <bool B16>
__global__ __launch_bounds__(256, 2) void conv3_bwd_weight_row_mfma_kernel(const float *__restrict__ x, const float *__restrict__ gout, float *__restrict__ gw,
                                                                           Geom g, int rows_per_wave, float *__restrict__ part)
{
    const int lane = threadIdx.x & 63, i = lane & 15, kg = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nrows = (long)g.B * g.D * g.H;
    const long r0 = (long)wave * rows_per_wave, r1 = r0 + rows_per_wave < nrows ? r0 + rows_per_wave : nrows;
    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool co_ok = i < g.Cout, ci_ok = i < g.C;
    const unsigned plane = (unsigned)(g.D * g.H * g.W);
    const int nseg = (g.W + 31) >> 5;
    const BufRsrc rx = make_rsrc(x, (size_t)g.B * g.C * plane * 4), rg = make_rsrc(gout, (size_t)g.B * g.Cout * plane * 4);
    struct XStep { f32x4 lo, hi; float lft, rgt; };
    // the x operands of step (tap_d, tap_h) of unit (row r, segment seg)
    auto request_x = [&](long r, int seg, int step, XStep &o) {
        const int td = step / 3, th = step % 3;
        const int h = (int)(r % g.H), d = (int)((r / g.H) % g.D), b = (int)(r / ((long)g.H * g.D));
        const int zd = d + td - 1, zh = h + th - 1, w0 = 32 * seg + 8 * kg;
        const bool ok = r < r1 && zd >= 0 && zd < g.D && zh >= 0 && zh < g.H && ci_ok && w0 < g.W;
        const unsigned off = ok ? (((unsigned)b * g.C + i) * plane + (unsigned)((zd * g.H + zh) * g.W + w0)) * 4u : DLKA_OOB;
        o.lo = buf_load_f32x4(rx, off);
        o.hi = buf_load_f32x4(rx, off + 16u);
        o.lft = buf_load_f32(rx, (ok && w0 > 0) ? off - 4u : DLKA_OOB);
        o.rgt = buf_load_f32(rx, (ok && w0 + 8 < g.W) ? off + 32u : DLKA_OOB);
    };
    auto request_g = [&](long r, int seg, f32x4 &lo, f32x4 &hi) {
        const int h = (int)(r % g.H), d = (int)((r / g.H) % g.D), b = (int)(r / ((long)g.H * g.D));
        const int w0 = 32 * seg + 8 * kg;
        const bool ok = r < r1 && co_ok && w0 < g.W;
        const unsigned off = ok ? (((unsigned)b * g.Cout + i) * plane + (unsigned)((d * g.H + h) * g.W + w0)) * 4u : DLKA_OOB;
        lo = buf_load_f32x4(rg, off);
        hi = buf_load_f32x4(rg, off + 16u);
    };
    // The operand loads run PD (tap_d, tap_h) steps ahead of their MFMAs, across segments and rows (9 % PD == 0: the ring slot of a step is known at compile time).  Timing-only
    // ablations of the one-step version (-DDLKA_ABLC): 361 us as built, 338 with the loads alone, 56 with the split + MFMAs alone — the kernel waits for its loads, two waves per
    // SIMD with four load instructions each in flight are too few for the ~2 us a line takes to arrive under load (profiles/r10_notes.md).
#ifndef DLKA_WGRAD_PD
#define DLKA_WGRAD_PD 3
#endif
    constexpr int PD = DLKA_WGRAD_PD;
    static_assert(9 % PD == 0 && PD < 9, "ring slots must be static");
    long r = r0;
    int seg = 0;
    f32x4 qcl, qch, qnl, qnh;
    XStep ring[PD];
    request_g(r, seg, qcl, qch);
#pragma unroll
    for (int s0 = 0; s0 < PD; ++s0) request_x(r, seg, s0, ring[s0]);
    while (r < r1) {   // wave-uniform
        const float q[8] = {qcl[0], qcl[1], qcl[2], qcl[3], qch[0], qch[1], qch[2], qch[3]};
        bf16x8 qh, ql;
        if (B16) split_bf16x8(q, qh, ql);
        int sn = seg + 1;
        long rn = r;
        if (sn == nseg) { sn = 0; rn = r + 1; }
#pragma unroll
        for (int step = 0; step < 9; ++step) {
            const XStep cx = ring[step % PD];
#ifndef DLKA_ABLC   // -DDLKA_ABLC=bits: TIMING-ONLY ablations (wrong results): 1 no operand loads in the loop, 2 no split / MFMAs in the loop
#define DLKA_ABLC 0
#endif
            if (!(DLKA_ABLC & 1)) {
                if (step + PD < 9) request_x(r, seg, step + PD, ring[step % PD]);
                else {
                    if (step + PD == 9) request_g(rn, sn, qnl, qnh);
                    request_x(rn, sn, step + PD - 9, ring[step % PD]);
                }
            }
            const float c[8] = {cx.lo[0], cx.lo[1], cx.lo[2], cx.lo[3], cx.hi[0], cx.hi[1], cx.hi[2], cx.hi[3]};
            const int t0 = step * 3;
            // tap_w = 0: x[w - 1], 1: x[w], 2: x[w + 1] for the lane's voxel w = w0 + e
            if (DLKA_ABLC & 2) { acc[t0][0] += c[0] + cx.lft + cx.rgt + c[7] + q[0]; }
            else if (B16) {
                const float xm[10] = {cx.lft, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], cx.rgt};   // x[w0 - 1 .. w0 + 8]
#pragma unroll
                for (int tw = 0; tw < 3; ++tw) {
                    bf16x8 bh, bl;
                    split_bf16x8(xm + tw, bh, bl);
                    acc[t0 + tw] = mfma_16x16x32_bf16(qh, bh, acc[t0 + tw]);
                    acc[t0 + tw] = mfma_16x16x32_bf16(ql, bh, acc[t0 + tw]);
                    acc[t0 + tw] = mfma_16x16x32_bf16(qh, bl, acc[t0 + tw]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    acc[t0] = mfma_16x16x4(q[e], e ? c[e ? e - 1 : 0] : cx.lft, acc[t0]);
                    acc[t0 + 1] = mfma_16x16x4(q[e], c[e], acc[t0 + 1]);
                    acc[t0 + 2] = mfma_16x16x4(q[e], e < 7 ? c[e < 7 ? e + 1 : 7] : cx.rgt, acc[t0 + 2]);
                }
            }
        }
        qcl = qnl; qch = qnh;
        r = rn; seg = sn;
    }
    conv3_wgrad_fold_and_add(acc, gw, g, lane, i, kg, ci_ok, part);
}

// ---------------------------------------------------------------------------------------------
// Round 6 — the same weight gradient, INPUT-row stationary.  The kernel above is bound by its operand loads (timing-only ablations at 2 x 16 x 64 x 128 x 128: 361 us as built, 338 with
// the loads alone, 56 with the split + MFMAs alone; a three-step prefetch ring changes nothing: 373) — every x row is requested nine times per output row, once per (tap_d, tap_h).
// Here a wave that owns `rows` consecutive h-rows of one (b, d) plane walks the rows + 2 INPUT rows zh of each of the three d-planes once per segment: x row (d + td - 1, zh) is the
// tap_h = 0 / 1 / 2 operand of the output rows zh + 1 / zh / zh - 1, whose grad_out rows sit in registers as bf16 pairs (a ring of three, split once per row): 3 (rows + 2) x-row
// requests per segment instead of 9 rows (30 instead of 72 at rows = 8), each feeding up to 27 MFMAs.  Same products and accumulators as the B16 form above (sums in another order).
// Needs H % rows == 0 (a wave's rows in one plane); the launcher falls back otherwise.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void conv3_bwd_weight_rows_b16_kernel(const float *__restrict__ x, const float *__restrict__ gout, float *__restrict__ gw, Geom g,
                                                                           int rows, float *__restrict__ part)
{
    const int lane = threadIdx.x & 63, i = lane & 15, kg = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nrows = (long)g.B * g.D * g.H;
    const long r0 = (long)wave * rows;
    const int nrw = r0 < nrows ? rows : 0;   // (H % rows == 0: whole groups only; a wave beyond the volume walks nothing and still joins the fold)
    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool co_ok = i < g.Cout, ci_ok = i < g.C;
    const unsigned plane = (unsigned)(g.D * g.H * g.W);
    const int nseg = (g.W + 31) >> 5;
    const BufRsrc rx = make_rsrc(x, (size_t)g.B * g.C * plane * 4), rg = make_rsrc(gout, (size_t)g.B * g.Cout * plane * 4);
    const int h0 = (int)(r0 % g.H), d = (int)((r0 / g.H) % g.D), b = nrw ? (int)(r0 / ((long)g.H * g.D)) : 0;
    struct XStep { f32x4 lo, hi; float lft, rgt; };
    // x row (d + td - 1, zh = h0 - 1 + zi) of segment seg: the lane's eight voxels and their two neighbours
    auto request_x = [&](int seg, int zi, int td, XStep &o) {
        const int zd = d + td - 1, zh = h0 - 1 + zi, w0 = 32 * seg + 8 * kg;
        const bool ok = zi < nrw + 2 && zd >= 0 && zd < g.D && zh >= 0 && zh < g.H && ci_ok && w0 < g.W;
        const unsigned off = ok ? (((unsigned)b * g.C + i) * plane + (unsigned)((zd * g.H + zh) * g.W + w0)) * 4u : DLKA_OOB;
        o.lo = buf_load_f32x4(rx, off);
        o.hi = buf_load_f32x4(rx, off + 16u);
        o.lft = buf_load_f32(rx, (ok && w0 > 0) ? off - 4u : DLKA_OOB);
        o.rgt = buf_load_f32(rx, (ok && w0 + 8 < g.W) ? off + 32u : DLKA_OOB);
    };
    // grad_out row h0 + k of the wave (zeros outside its range: those rows belong to other waves)
    auto request_g = [&](int seg, int k, f32x4 &lo, f32x4 &hi) {
        const int w0 = 32 * seg + 8 * kg;
        const bool ok = k >= 0 && k < nrw && co_ok && w0 < g.W;
        const unsigned off = ok ? (((unsigned)b * g.Cout + i) * plane + (unsigned)((d * g.H + h0 + k) * g.W + w0)) * 4u : DLKA_OOB;
        lo = buf_load_f32x4(rg, off);
        hi = buf_load_f32x4(rg, off + 16u);
    };
    auto split_row = [&](const f32x4 &lo, const f32x4 &hi, bf16x8 &qh, bf16x8 &ql) {
        const float q[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        split_bf16x8(q, qh, ql);
    };
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int seg = 0; seg < (nrw ? nseg : 0); ++seg) {   // wave-uniform
        // grad_out ring: rows zh - 1, zh, zh + 1 of the current input row zh = h0 - 1 + zi, i.e. wave rows zi - 2, zi - 1, zi
        bf16x8 gmh, gml, g0h, g0l, gph, gpl;
        split_row(z4, z4, gmh, gml);
        split_row(z4, z4, g0h, g0l);
        f32x4 nl, nh;
        request_g(seg, 0, nl, nh);
        XStep nx;
        request_x(seg, 0, 0, nx);
        split_row(nl, nh, gph, gpl);
        request_g(seg, 1, nl, nh);
        for (int zi = 0; zi < nrw + 2; ++zi) {
#pragma unroll
            for (int td = 0; td < 3; ++td) {
                const XStep cx = nx;
                if (td < 2) request_x(seg, zi, td + 1, nx);
                else request_x(seg, zi + 1, 0, nx);   // (zi + 1 == nrw + 2: zeros, no traffic)
                const float xm[10] = {cx.lft, cx.lo[0], cx.lo[1], cx.lo[2], cx.lo[3], cx.hi[0], cx.hi[1], cx.hi[2], cx.hi[3], cx.rgt};   // x[w0 - 1 .. w0 + 8]
                bf16x8 bh[3], bl[3];
#pragma unroll
                for (int tw = 0; tw < 3; ++tw) split_bf16x8(xm + tw, bh[tw], bl[tw]);
                // tap_h = th: output row zh + 1 - th = wave row zi - th
#pragma unroll
                for (int th = 0; th < 3; ++th) {
                    if ((unsigned)(zi - th) >= (unsigned)nrw) continue;   // (uniform) that output row is another wave's
                    const bf16x8 qh = th == 0 ? gph : (th == 1 ? g0h : gmh), ql = th == 0 ? gpl : (th == 1 ? g0l : gml);
                    const int t0 = (td * 3 + th) * 3;
#pragma unroll
                    for (int tw = 0; tw < 3; ++tw) {
                        acc[t0 + tw] = mfma_16x16x32_bf16(qh, bh[tw], acc[t0 + tw]);
                        acc[t0 + tw] = mfma_16x16x32_bf16(ql, bh[tw], acc[t0 + tw]);
                        acc[t0 + tw] = mfma_16x16x32_bf16(qh, bl[tw], acc[t0 + tw]);
                    }
                }
            }
            gmh = g0h; gml = g0l; g0h = gph; g0l = gpl;
            split_row(nl, nh, gph, gpl);          // wave row zi + 1 (zeros beyond the wave's rows)
            request_g(seg, zi + 2, nl, nh);
        }
    }
    conv3_wgrad_fold_and_add(acc, gw, g, lane, i, kg, ci_ok, part);
}

static bool conv3_wgrad_mfma_shape(const Geom &g)
{
    return g.group == 1 && g.kd == 3 && g.kh == 3 && g.kw == 3 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 1 && g.ph == 1 && g.pw == 1 && g.dd == 1 &&
           g.dh == 1 && g.dw == 1 && g.C <= 16 && g.Cout <= 16 && (g.W & 3) == 0;
}
static bool conv3_wgrad_row_shape(const Geom &g) { return conv3_wgrad_mfma_shape(g) && g.W % 8 == 0 && (size_t)g.B * 16 * g.D * g.H * g.W * 4 < ((size_t)1 << 31); }
static long conv3_wgrad_waves(const Geom &g, int &rpw)
{
    const long nrows = (long)g.B * g.D * g.H;
    rpw = (int)cdivl(nrows, 2048);   // ~2 waves per SIMD over the chip
    if (rpw < 8) rpw = 8;
    return cdivl(nrows, rpw);
}
size_t conv_bwd_weight_part_floats(const Geom &g, size_t elem_bytes)
{
    if (elem_bytes != 4 || !conv3_wgrad_row_shape(g)) return 0;
    int rpw;
    const long nwg = cdivl(conv3_wgrad_waves(g, rpw), 4);
    return nwg >= 8 ? (size_t)nwg * 27 * 256 : 0;   // (few workgroups: their atomics are cheap)
}

template <typename T>
int launch_conv_bwd_weight(const T *x, const T *gout, float *gw32, const Geom &g, hipStream_t st, float *part)
{
    if constexpr (sizeof(T) == 4) {
        if (conv3_wgrad_mfma_shape(g)) {
            int rpw;
            const long waves = conv3_wgrad_waves(g, rpw);
            const int nwg = (int)cdivl(waves, 4);
            if (conv3_wgrad_row_shape(g)) {
                if (part && conv_bwd_weight_part_floats(g, 4) == 0) part = nullptr;
                static const bool exact = getenv("DLKA_EXACT_FP32") != nullptr;
                static const bool rows_off = [] { const char *e = getenv("DLKA_CONV3_WGRAD_ROWS"); return e && e[0] == '0'; }();   // (A/B: 0 = the output-row form)
                if (!exact && !rows_off && g.H % rpw == 0) {
                    DLKA_LAUNCH(conv3_bwd_weight_rows_b16_kernel, dim3((unsigned)nwg), dim3(256), 0, st, reinterpret_cast<const float *>(x),
                                reinterpret_cast<const float *>(gout), gw32, g, rpw, part);
                } else if (exact) {
                    DLKA_LAUNCH(conv3_bwd_weight_row_mfma_kernel<false>, dim3((unsigned)nwg), dim3(256), 0, st, reinterpret_cast<const float *>(x),
                                reinterpret_cast<const float *>(gout), gw32, g, rpw, part);
                } else {
                    DLKA_LAUNCH(conv3_bwd_weight_row_mfma_kernel<true>, dim3((unsigned)nwg), dim3(256), 0, st, reinterpret_cast<const float *>(x),
                                reinterpret_cast<const float *>(gout), gw32, g, rpw, part);
                }
                DLKA_CHECK_LAUNCH();
                if (part) {
                    DLKA_LAUNCH(conv3_wgrad_reduce_kernel, dim3(27, (unsigned)(nwg >= 256 ? 16 : 4)), dim3(256), 0, st, (const float *)part, gw32, nwg, g.C, g.Cout);
                    DLKA_CHECK_LAUNCH();
                }
                return DLKA_OK;
            }
            DLKA_LAUNCH(conv3_bwd_weight_mfma_kernel, dim3((unsigned)nwg), dim3(256), 0, st, reinterpret_cast<const float *>(x),
                        reinterpret_cast<const float *>(gout), gw32, g, rpw);
            DLKA_CHECK_LAUNCH();
            return DLKA_OK;
        }
    }
    constexpr int TPC = 4;
    const int cob = g.Og >= 16 ? 16 : (g.Og >= 8 ? 8 : (g.Og >= 4 ? 4 : (g.Og >= 2 ? 2 : 1)));
    const int cochunks = cdiv(g.Og, cob), tchunks = cdiv(g.K, TPC);
    const long total = (long)g.B * g.No;
    long want = 2048 / ((long)g.C * tchunks * cochunks) + 1;
    long maxvs = cdivl(total, DLKA_THREADS);
    int VS = (int)(want < 1 ? 1 : (want > maxvs ? maxvs : want));
    if (VS > 64) VS = 64;
    dim3 grid(g.C, tchunks * cochunks, VS), block(DLKA_THREADS);
#define DLKA_L(COB)                                                    \
    {                                                                  \
        auto k = conv_bwd_weight_kernel<T, TPC, COB>;                  \
        DLKA_LAUNCH(k, grid, block, 0, st, x, gout, gw32, g, cochunks); \
    }
    switch (cob) {
        case 1: DLKA_L(1) break;
        case 2: DLKA_L(2) break;
        case 4: DLKA_L(4) break;
        case 8: DLKA_L(8) break;
        default: DLKA_L(16) break;
    }
#undef DLKA_L
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

#define DLKA_INST(T)                                                                                        \
    template int launch_conv_fwd<T>(const T *, const T *, const T *, T *, float *, const Geom &, hipStream_t); \
    template int launch_conv_bwd_data<T>(const T *, const T *, T *, float *, const Geom &, hipStream_t);      \
    template int launch_conv_bwd_weight<T>(const T *, const T *, float *, const Geom &, hipStream_t, float *);
DLKA_INST(float)
DLKA_INST(bf16_t)
#undef DLKA_INST

}  // namespace dlka
