// Deformable convolution (3-D D3D semantics, 2-D torchvision semantics) — direct "gather + contract" kernels.
//
// Replaces the reference's im2col -> cuBLAS pipeline (3D/dcn/src/cuda/deform_conv_cuda.cu:95-123,
// deform_im2col_cuda.cuh:192-265) without ever materialising the (27*C x B*N) column buffer
// (226 MB per call at C=32, 32^3, B=2 in the reference): one work-item owns one output voxel, forms each
// sample once per (tap, channel) and contracts it in registers against wave-uniform weights that the
// compiler fetches through the scalar cache.  Lanes of a wave are consecutive voxels along W, so offset
// reads / output writes are fully coalesced and the 8-corner gathers of neighbouring lanes land in the
// same or adjacent cache lines.
//
// Kernels in this file are the GENERAL path (any kernel size / stride / pad / dilation / groups /
// deformable groups).  The shape-specialised kernels of the D-LKA hot configuration are the channels-last ones
// (cl_*.hip), entered through dlka_capi_cl.hip.
#include "deform_sample.h"
#include "cl_gather.h"
#include "cl_ddw2d_describe.h"
#include "dlka_kernels.h"

namespace dlka {

// ---------------------------------------------------------------------------------------------
// weight re-layout:  W[co][cg][tap]  ->  Wt[g][tap][cg][OgP]   (OgP = Og rounded up, zero padded)
// so that the Og weights that multiply one sample are contiguous (scalar-cache friendly s_load_dwordxN).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void relayout_weight_kernel(const T *__restrict__ w, float *__restrict__ wt, int group, int Og, int Cg, int K, int OgP)
{
    const int n = group * K * Cg * OgP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int o = i % OgP, cg = (i / OgP) % Cg, tap = (i / OgP / Cg) % K, g = i / OgP / Cg / K;
        wt[i] = (o < Og) ? ldf(w + ((long)(g * Og + o) * Cg + cg) * K + tap) : 0.f;
    }
}

template <typename T>
int launch_relayout_weight(const T *w, float *wt, int group, int Og, int Cg, int K, int OgP, hipStream_t st)
{
    const int n = group * K * Cg * OgP;
    auto k = relayout_weight_kernel<T>;
    DLKA_LAUNCH(k, dim3(cdiv(n, 256)), dim3(256), 0, st, w, wt, group, Og, Cg, K, OgP);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// ---------------------------------------------------------------------------------------------
// forward
// grid = (voxel tiles of 256, group * OgP/COB, B)
// ---------------------------------------------------------------------------------------------
template <typename T, int NOFF, int COB>
__global__ __launch_bounds__(DLKA_THREADS) void deform_fwd_kernel(
    const T *__restrict__ x, const T *__restrict__ off, const float *__restrict__ wt, const T *__restrict__ bias,
    T *__restrict__ out, Geom g, int OgP)
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

    const int c_lo = gi * g.Cg;
    TapSample<NOFF> s;
    int tap = 0;
    for (int i = 0; i < g.kd; ++i)
        for (int jx = 0; jx < g.kh; ++jx)
            for (int k = 0; k < g.kw; ++k, ++tap) {
                int cur_dgi = -1;
                const float *wrow = wt + ((long)(gi * g.K + tap) * g.Cg) * OgP + co0;
                for (int cg = 0; cg < g.Cg; ++cg) {
                    const int c = c_lo + cg;
                    const int dgi = c / g.cpdg;
                    if (dgi != cur_dgi) {  // wave-uniform
                        cur_dgi = dgi;
                        const T *offp = off + ((long)(b * g.dg + dgi) * NOFF * g.K + NOFF * tap) * g.No + v;
                        setup_tap<NOFF>(s, offp, g.No, bd + i * g.dd, bh + jx * g.dh, bw + k * g.dw, g.D, g.H, g.W);
                    }
                    const float val = sample_value<NOFF>(s, x + (long)(b * g.C + c) * g.Ni);
                    const float *wp = wrow + (long)cg * OgP;
#pragma unroll
                    for (int j = 0; j < COB; ++j) acc[j] = fmaf(val, wp[j], acc[j]);
                }
            }
#pragma unroll
    for (int j = 0; j < COB; ++j) {
        const int o = co0 + j;
        if (o < g.Og) {
            const int co = gi * g.Og + o;
            const float bv = bias ? ldf(bias + co) : 0.f;
            stf(out + (long)(b * g.Cout + co) * g.No + v, acc[j] + bv);
        }
    }
}

static int pick_cob(int Og)
{
    if (Og >= 32) return 32;
    int c = 1;
    while (c < Og) c <<= 1;
    return c;
}

int deform_fwd_wt_floats(const Geom &g) { return g.group * g.K * g.Cg * round_up(g.Og, pick_cob(g.Og)); }

template <typename T, int NOFF>
int launch_deform_fwd(const T *x, const T *off, const T *w, const T *bias, T *out, float *wt, const Geom &g, hipStream_t st)
{
    const int cob = pick_cob(g.Og);
    const int OgP = round_up(g.Og, cob);
    int rc = launch_relayout_weight<T>(w, wt, g.group, g.Og, g.Cg, g.K, OgP, st);
    if (rc) return rc;
    dim3 grid(cdiv(g.No, DLKA_THREADS), g.group * (OgP / cob), g.B), block(DLKA_THREADS);
#define DLKA_LAUNCH_FWD(COB)                                                                 \
    {                                                                                        \
        auto k = deform_fwd_kernel<T, NOFF, COB>;                                            \
        DLKA_LAUNCH(k, grid, block, 0, st, x, off, (const float *)wt, bias, out, g, OgP); \
    }
    switch (cob) {
        case 1: DLKA_LAUNCH_FWD(1) break;
        case 2: DLKA_LAUNCH_FWD(2) break;
        case 4: DLKA_LAUNCH_FWD(4) break;
        case 8: DLKA_LAUNCH_FWD(8) break;
        case 16: DLKA_LAUNCH_FWD(16) break;
        default: DLKA_LAUNCH_FWD(32) break;
    }
#undef DLKA_LAUNCH_FWD
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// ---------------------------------------------------------------------------------------------
// backward w.r.t. input and offsets (fused): one work-item per (b, dg, output voxel).
//   col(c,tap)   = sum_o W[o][c][tap] * gO[b][o][v]                  (cu:226-231, never stored)
//   gOff(tap,a) += col * d(sample)/d(q_a)        summed over the channels of the dg group   (cuh:336-405)
//   gX[corner]  += col * w_corner                fp32 atomics, as the reference (cuh:327)    (cuh:267-334)
// OGR > 0: group == 1 and Og <= OGR: the work-item keeps its Og grad_out values in registers.
// grid = (voxel tiles, dg, B)
// ---------------------------------------------------------------------------------------------
template <typename T, int NOFF, int OGR>
__global__ __launch_bounds__(DLKA_THREADS) void deform_bwd_input_offset_kernel(
    const T *__restrict__ x, const T *__restrict__ off, const float *__restrict__ wt, const T *__restrict__ gout,
    float *__restrict__ gx /* fp32, zero-initialised, may be null */, T *__restrict__ goff /* may be null */, Geom g, int OgP)
{
    const int v = blockIdx.x * DLKA_THREADS + threadIdx.x;
    const int dgi = blockIdx.y, b = blockIdx.z;
    if (v >= g.No) return;
    const int ow = v % g.Wo, oh = (v / g.Wo) % g.Ho, od = v / (g.Wo * g.Ho);
    const int bd = od * g.sd - g.pd, bh = oh * g.sh - g.ph, bw = ow * g.sw - g.pw;

    float G[OGR > 0 ? OGR : 1];
    if (OGR > 0) {
#pragma unroll
        for (int o = 0; o < OGR; ++o) G[o] = (o < g.Og) ? ldf(gout + (long)(b * g.Cout + o) * g.No + v) : 0.f;
    }

    TapSample<NOFF> s;
    int tap = 0;
    for (int i = 0; i < g.kd; ++i)
        for (int jx = 0; jx < g.kh; ++jx)
            for (int k = 0; k < g.kw; ++k, ++tap) {
                const T *offp = off + ((long)(b * g.dg + dgi) * NOFF * g.K + NOFF * tap) * g.No + v;
                setup_tap<NOFF>(s, offp, g.No, bd + i * g.dd, bh + jx * g.dh, bw + k * g.dw, g.D, g.H, g.W);
                float go_d = 0.f, go_h = 0.f, go_w = 0.f;
                // D3D zeroes the coordinate gradient outside the guard (cuh:391-394, 116-120); torchvision's
                // get_coordinate_weight has no guard, only per-corner bounds.
                const unsigned dmask = (NOFF == 3) ? s.ok : s.cok;
                for (int cc = 0; cc < g.cpdg; ++cc) {
                    const int c = dgi * g.cpdg + cc;
                    const int gi = c / g.Cg, cg = c - gi * g.Cg;
                    const float *wp = wt + ((long)(gi * g.K + tap) * g.Cg + cg) * OgP;
                    float col = 0.f;
                    if (OGR > 0) {
#pragma unroll
                        for (int o = 0; o < OGR; ++o) col = fmaf(G[o], wp[o], col);
                    } else {
                        const T *gp = gout + (long)(b * g.Cout + gi * g.Og) * g.No + v;
                        for (int o = 0; o < g.Og; ++o) col = fmaf(ldf(gp + (long)o * g.No), wp[o], col);
                    }
                    const T *xp = x + (long)(b * g.C + c) * g.Ni;
                    float *gxp = gx ? gx + (long)(b * g.C + c) * g.Ni : nullptr;
                    float dd_ = 0.f, dh_ = 0.f, dw_ = 0.f;
#pragma unroll
                    for (int q = 0; q < TapSample<NOFF>::NC; ++q) {
                        const int cd = (NOFF == 3) ? (q >> 2) & 1 : 0, ch = (q >> 1) & 1, cw = q & 1;
                        const float xv = ((dmask >> q) & 1u) ? ldf(xp + s.idx[q]) : 0.f;
                        if (NOFF == 3) {
                            dd_ = fmaf((cd ? 1.f : -1.f) * s.fh[ch] * s.fw[cw], xv, dd_);
                            dh_ = fmaf((ch ? 1.f : -1.f) * s.fd[cd] * s.fw[cw], xv, dh_);
                            dw_ = fmaf((cw ? 1.f : -1.f) * s.fd[cd] * s.fh[ch], xv, dw_);
                        } else {
                            dh_ = fmaf((ch ? 1.f : -1.f) * s.fw[cw], xv, dh_);
                            dw_ = fmaf((cw ? 1.f : -1.f) * s.fh[ch], xv, dw_);
                        }
                        if (gxp && ((s.ok >> q) & 1u)) atomicAdd(gxp + s.idx[q], col * s.w[q]);
                    }
                    go_d = fmaf(col, dd_, go_d);
                    go_h = fmaf(col, dh_, go_h);
                    go_w = fmaf(col, dw_, go_w);
                }
                if (goff) {
                    T *gop = goff + ((long)(b * g.dg + dgi) * NOFF * g.K + NOFF * tap) * g.No + v;
                    if (NOFF == 3) {
                        stf(gop, go_d);
                        stf(gop + g.No, go_h);
                        stf(gop + 2 * (long)g.No, go_w);
                    } else {
                        stf(gop, go_h);
                        stf(gop + g.No, go_w);
                    }
                }
            }
}

template <typename T, int NOFF>
int launch_deform_bwd_input_offset(const T *x, const T *off, const float *wt, int OgP, const T *gout,
                                   float *gx32, T *goff, const Geom &g, hipStream_t st)
{
    dim3 grid(cdiv(g.No, DLKA_THREADS), g.dg, g.B), block(DLKA_THREADS);
#define DLKA_LAUNCH_BIO(OGR)                                                            \
    {                                                                                   \
        auto k = deform_bwd_input_offset_kernel<T, NOFF, OGR>;                          \
        DLKA_LAUNCH(k, grid, block, 0, st, x, off, wt, gout, gx32, goff, g, OgP); \
    }
    if (g.group == 1 && g.Og <= 8 && OgP >= 8) DLKA_LAUNCH_BIO(8)
    else if (g.group == 1 && g.Og <= 16 && OgP >= 16) DLKA_LAUNCH_BIO(16)
    else if (g.group == 1 && g.Og <= 32 && OgP >= 32) DLKA_LAUNCH_BIO(32)
    else DLKA_LAUNCH_BIO(0)
#undef DLKA_LAUNCH_BIO
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// ---------------------------------------------------------------------------------------------
// backward w.r.t. weight:  gW[co][cg][tap] = sum_{b,v} gO[b][co][v] * S(c,tap,b,v)     (cu:254-277)
// The sample is recomputed (the reference recomputes the whole im2col buffer, cu:254-261).
// Block = (input channel c, chunk of TPC taps, chunk of COB out-channels, voxel split); every work-item
// strides over (b, v), keeps TPC*COB partial sums, then wave-shuffle + LDS reduce -> one atomicAdd per value.
// grid = (C, tapchunks * cochunks, VS)
// ---------------------------------------------------------------------------------------------
template <typename T, int NOFF, int TPC, int COB>
__global__ __launch_bounds__(DLKA_THREADS) void deform_bwd_weight_kernel(
    const T *__restrict__ x, const T *__restrict__ off, const T *__restrict__ gout,
    float *__restrict__ gw /* fp32 [Cout][Cg][K], zero-initialised */, Geom g, int cochunks)
{
    const int c = blockIdx.x;
    const int tchunk = blockIdx.y / cochunks, cchunk = blockIdx.y % cochunks;
    const int tap0 = tchunk * TPC, co0 = cchunk * COB;
    const int gi = c / g.Cg, cg = c - gi * g.Cg, dgi = c / g.cpdg;
    const int VS = gridDim.z;

    float acc[TPC][COB];
#pragma unroll
    for (int t = 0; t < TPC; ++t)
#pragma unroll
        for (int j = 0; j < COB; ++j) acc[t][j] = 0.f;

    // decode the TPC taps once (uniform)
    int ti[TPC], tj[TPC], tk[TPC];
#pragma unroll
    for (int t = 0; t < TPC; ++t) {
        const int tap = tap0 + t;
        tk[t] = tap % g.kw; tj[t] = (tap / g.kw) % g.kh; ti[t] = tap / (g.kw * g.kh);
    }

    const long total = (long)g.B * g.No;
    TapSample<NOFF> s;
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
            if (tap0 + t < g.K) {  // uniform
                const T *offp = off + ((long)(b * g.dg + dgi) * NOFF * g.K + NOFF * (tap0 + t)) * g.No + v;
                setup_tap<NOFF>(s, offp, g.No, bd + ti[t] * g.dd, bh + tj[t] * g.dh, bw + tk[t] * g.dw, g.D, g.H, g.W);
                const float val = sample_value<NOFF>(s, xp);
#pragma unroll
                for (int j = 0; j < COB; ++j) acc[t][j] = fmaf(val, G[j], acc[t][j]);
            }
        }
    }

    // reduce: wave (64 lanes) by shuffles, then the 4 waves through LDS
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

template <typename T, int NOFF>
int launch_deform_bwd_weight(const T *x, const T *off, const T *gout, float *gw32, const Geom &g, hipStream_t st)
{
    constexpr int TPC = 4;
    const int cob = g.Og >= 16 ? 16 : (g.Og >= 8 ? 8 : (g.Og >= 4 ? 4 : (g.Og >= 2 ? 2 : 1)));
    const int cochunks = cdiv(g.Og, cob), tchunks = cdiv(g.K, TPC);
    const long total = (long)g.B * g.No;
    // enough blocks to fill 256 CUs a few times over, but at least 256 voxels' worth of work per block
    long want = 2048 / ((long)g.C * tchunks * cochunks) + 1;
    long maxvs = cdivl(total, DLKA_THREADS);
    int VS = (int)(want < 1 ? 1 : (want > maxvs ? maxvs : want));
    if (VS > 64) VS = 64;
    dim3 grid(g.C, tchunks * cochunks, VS), block(DLKA_THREADS);
#define DLKA_LAUNCH_BW(COB)                                                    \
    {                                                                          \
        auto k = deform_bwd_weight_kernel<T, NOFF, TPC, COB>;                  \
        DLKA_LAUNCH(k, grid, block, 0, st, x, off, gout, gw32, g, cochunks); \
    }
    switch (cob) {
        case 1: DLKA_LAUNCH_BW(1) break;
        case 2: DLKA_LAUNCH_BW(2) break;
        case 4: DLKA_LAUNCH_BW(4) break;
        case 8: DLKA_LAUNCH_BW(8) break;
        default: DLKA_LAUNCH_BW(16) break;
    }
#undef DLKA_LAUNCH_BW
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// ---------------------------------------------------------------------------------------------
// grad_bias[co] = sum_{b,v} gO[b][co][v]     (cu:223,277: addmv with a ones vector)
// one block per output channel
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(DLKA_THREADS) void bias_grad_kernel(const T *__restrict__ gout, T *__restrict__ gb, int B, int Cout, int No)
{
    const int co = blockIdx.x;
    float a = 0.f;
    for (int b = 0; b < B; ++b) {
        const T *p = gout + (long)(b * Cout + co) * No;
        for (int v = threadIdx.x; v < No; v += DLKA_THREADS) a += ldf(p + v);
    }
    __shared__ float red[DLKA_THREADS / 64];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < DLKA_THREADS / 64; ++w) t += red[w];
        stf(gb + co, t);
    }
}

template <typename T>
int launch_bias_grad(const T *gout, T *gb, int B, int Cout, int No, hipStream_t st)
{
    auto k = bias_grad_kernel<T>;
    DLKA_LAUNCH(k, dim3(Cout), dim3(DLKA_THREADS), 0, st, gout, gb, B, Cout, No);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// fp32 accumulation buffer -> storage type
template <typename T>
__global__ void cast_from_f32_kernel(const float *__restrict__ src, T *__restrict__ dst, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) stf(dst + i, src[i]);
}

template <typename T>
int launch_cast_from_f32(const float *src, T *dst, long n, hipStream_t st)
{
    if (n <= 0) return DLKA_OK;
    long blocks = cdivl(n, 256);
    if (blocks > 4096) blocks = 4096;
    auto k = cast_from_f32_kernel<T>;
    DLKA_LAUNCH(k, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, n);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// ---------------------------------------------------------------------------------------------
// debug: floor indices + guard mask (bit-exact index parity)
// ---------------------------------------------------------------------------------------------
// path 0: the rule written out on its own; path 1: through setup_tap<3> (what every general-path kernel calls);
// path 2: through gather_describe3 (what every channels-last fast-path kernel calls).  Paths 1/2 report the cell only where
// the guard holds (outside it the hot kernels never form an address).
template <typename T, int PATH>
__global__ void sample_index_kernel(const T *__restrict__ off, int32_t *__restrict__ idx, uint8_t *__restrict__ mask, Geom g)
{
    const long n = (long)g.B * g.dg * g.K * g.No;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int v = (int)(e % g.No), tap = (int)((e / g.No) % g.K);
        const long bg = e / g.No / g.K;  // b*dg + dgi
        const int ow = v % g.Wo, oh = (v / g.Wo) % g.Ho, od = v / (g.Wo * g.Ho);
        const int k = tap % g.kw, j = (tap / g.kw) % g.kh, i = tap / (g.kw * g.kh);
        const T *offp = off + (bg * 3 * g.K + 3 * tap) * g.No + v;
        const int bd = od * g.sd - g.pd + i * g.dd, bh = oh * g.sh - g.ph + j * g.dh, bw = ow * g.sw - g.pw + k * g.dw;
        if (PATH == 0) {   // the rule itself (also what cl_deform_gx_fx2_kernel calls directly)
            int zd, zh, zw;
            float ld, lh, lw;
            const bool in = sample_cell3(ldf(offp), ldf(offp + g.No), ldf(offp + 2 * (long)g.No), bd, bh, bw, g.D, g.H, g.W, zd, zh, zw, ld, lh, lw);
            idx[e * 3 + 0] = zd;
            idx[e * 3 + 1] = zh;
            idx[e * 3 + 2] = zw;
            mask[e] = in ? 1 : 0;
        } else if (PATH == 1) {
            TapSample<3> s;
            setup_tap<3, T>(s, offp, g.No, bd, bh, bw, g.D, g.H, g.W);
            idx[e * 3 + 0] = s.inside ? s.z0[0] : 0;
            idx[e * 3 + 1] = s.inside ? s.z0[1] : 0;
            idx[e * 3 + 2] = s.inside ? s.z0[2] : 0;
            mask[e] = s.inside ? 1 : 0;
        } else if (PATH == 2) {
            const RowDesc r = gather_describe3(ldf(offp), ldf(offp + g.No), ldf(offp + 2 * (long)g.No), (long)g.Ni, 0, bd, bh, bw, g.D, g.H, g.W);
            idx[e * 3 + 0] = r.zd;
            idx[e * 3 + 1] = r.zh;
            idx[e * 3 + 2] = r.zw;
            mask[e] = r.okm ? 1 : 0;   // inside the guard <=> at least one corner contributes
        } else {   // PATH 3: lane_tap of the grad_input window kernels (cl_deform_bwd2.hip)
            LaneTap lt;
            lane_tap(lt, ldf(offp), ldf(offp + g.No), ldf(offp + 2 * (long)g.No), bd, bh, bw, g.D, g.H, g.W);
            idx[e * 3 + 0] = lt.zd;
            idx[e * 3 + 1] = lt.zh;
            idx[e * 3 + 2] = lt.zw;
            mask[e] = lt.okm ? 1 : 0;
        }
    }
}

// 2-D (torchvision layout: (dy, dx) per tap).  idx [.][2] = the floor cell where `reach`, else 0; mask bit 0 = sample inside the guard,
// bit 1 = reach (some corner can lie inside the image: the coordinate weight's domain).  PATH 0 = sample_cell2, 1 = setup_tap<2> (general
// kernels), 2 = describe2 (cl_ddw2d.hip; also what its window scatter calls through sample_cell2).
template <typename T, int PATH>
__global__ void sample_index2_kernel(const T *__restrict__ off, int32_t *__restrict__ idx, uint8_t *__restrict__ mask, Geom g)
{
    const long n = (long)g.B * g.dg * g.K * g.No;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int v = (int)(e % g.No), tap = (int)((e / g.No) % g.K);
        const long bg = e / g.No / g.K;
        const int ow = v % g.Wo, oh = v / g.Wo;
        const int k = tap % g.kw, j = tap / g.kw;
        const T *offp = off + (bg * 2 * g.K + 2 * tap) * g.No + v;
        const int bh = oh * g.sh - g.ph + j * g.dh, bw = ow * g.sw - g.pw + k * g.dw;
        if (PATH == 0) {
            int y0, x0;
            float ly, lx;
            bool reach;
            const bool in = sample_cell2(ldf(offp), ldf(offp + g.No), bh, bw, g.H, g.W, y0, x0, ly, lx, reach);
            idx[e * 2 + 0] = y0;
            idx[e * 2 + 1] = x0;
            mask[e] = (in ? 1 : 0) | (reach ? 2 : 0);
        } else if (PATH == 1) {
            TapSample<2> s;
            setup_tap<2, T>(s, offp, g.No, 0, bh, bw, 1, g.H, g.W);
            idx[e * 2 + 0] = s.z0[1];
            idx[e * 2 + 1] = s.z0[2];
            mask[e] = (s.inside ? 1 : 0) | (s.reach ? 2 : 0);
        } else {
            // describe2 does not keep the cell: it is recovered from the row offset of the first corner inside the image (rowbytes = 1 -> offset = pixel)
            Tap2 t2;
            describe2(t2, ldf(offp), ldf(offp + g.No), 0, bh, bw, g.H, g.W, g.H * g.W, 1);
            int y0 = 0, x0 = 0;
            bool any = false;
#pragma unroll
            for (int q = 3; q >= 0; --q)
                if (t2.off[q] != DLKA_OOB) { const int px = (int)t2.off[q]; y0 = px / g.W - (q >> 1); x0 = px % g.W - (q & 1); any = true; }
            idx[e * 2 + 0] = any ? y0 : 0;
            idx[e * 2 + 1] = any ? x0 : 0;
            mask[e] = (t2.okm ? 1 : 0) | (any ? 2 : 0);
        }
    }
}

template <typename T>
int launch_sample_index2(const T *off, int32_t *idx, uint8_t *mask, const Geom &g, int path, hipStream_t st)
{
    const long n = (long)g.B * g.dg * g.K * g.No;
    long blocks = cdivl(n, 256);
    if (blocks > 8192) blocks = 8192;
    if (path == 0) DLKA_LAUNCH((sample_index2_kernel<T, 0>), dim3((unsigned)blocks), dim3(256), 0, st, off, idx, mask, g);
    else if (path == 1) DLKA_LAUNCH((sample_index2_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, st, off, idx, mask, g);
    else if (path == 2) DLKA_LAUNCH((sample_index2_kernel<T, 2>), dim3((unsigned)blocks), dim3(256), 0, st, off, idx, mask, g);
    else return DLKA_ERR_UNSUPPORTED;
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

template <typename T>
int launch_sample_index(const T *off, int32_t *idx, uint8_t *mask, const Geom &g, int path, hipStream_t st)
{
    const long n = (long)g.B * g.dg * g.K * g.No;
    long blocks = cdivl(n, 256);
    if (blocks > 8192) blocks = 8192;
    if (path == 0) DLKA_LAUNCH((sample_index_kernel<T, 0>), dim3((unsigned)blocks), dim3(256), 0, st, off, idx, mask, g);
    else if (path == 1) DLKA_LAUNCH((sample_index_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, st, off, idx, mask, g);
    else if (path == 2) DLKA_LAUNCH((sample_index_kernel<T, 2>), dim3((unsigned)blocks), dim3(256), 0, st, off, idx, mask, g);
    else if (path == 3) DLKA_LAUNCH((sample_index_kernel<T, 3>), dim3((unsigned)blocks), dim3(256), 0, st, off, idx, mask, g);
    else return DLKA_ERR_UNSUPPORTED;
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// explicit instantiations used by dlka_capi.hip
#define DLKA_INST(T)                                                                                                          \
    template int launch_relayout_weight<T>(const T *, float *, int, int, int, int, int, hipStream_t);                          \
    template int launch_deform_fwd<T, 3>(const T *, const T *, const T *, const T *, T *, float *, const Geom &, hipStream_t); \
    template int launch_deform_fwd<T, 2>(const T *, const T *, const T *, const T *, T *, float *, const Geom &, hipStream_t); \
    template int launch_deform_bwd_input_offset<T, 3>(const T *, const T *, const float *, int, const T *, float *, T *, const Geom &, hipStream_t); \
    template int launch_deform_bwd_input_offset<T, 2>(const T *, const T *, const float *, int, const T *, float *, T *, const Geom &, hipStream_t); \
    template int launch_deform_bwd_weight<T, 3>(const T *, const T *, const T *, float *, const Geom &, hipStream_t);          \
    template int launch_deform_bwd_weight<T, 2>(const T *, const T *, const T *, float *, const Geom &, hipStream_t);          \
    template int launch_bias_grad<T>(const T *, T *, int, int, int, hipStream_t);                                              \
    template int launch_cast_from_f32<T>(const float *, T *, long, hipStream_t);                                               \
    template int launch_sample_index<T>(const T *, int32_t *, uint8_t *, const Geom &, int, hipStream_t);                      \
    template int launch_sample_index2<T>(const T *, int32_t *, uint8_t *, const Geom &, int, hipStream_t);
DLKA_INST(float)
DLKA_INST(bf16_t)
#undef DLKA_INST

int deform_pick_cob(int Og) { return pick_cob(Og); }

}  // namespace dlka
