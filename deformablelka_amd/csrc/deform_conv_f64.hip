// DLKA_F64: the deformable conv (3-D D3D semantics, 2-D torchvision semantics) and the plain grouped conv in DOUBLE precision — the second half of the
// reference's dispatch (AT_DISPATCH_FLOATING_TYPES = float, double: 3D/dcn/src/cuda/deform_conv_cuda.cu:96,233; the reference imports gradcheck, 3D/dcn/test.py:9).
// General NCDHW layout only, any kernel size / stride / padding / dilation / groups / deformable groups; every value, weight, coordinate and accumulator is a double.
// Not a fast path — one work-item per output element, the contraction in a loop — its purpose is numerical: torch.autograd.gradcheck runs THROUGH the product, and the
// reference's own op compiled for double is matched to ~1e-12 (tests/test_f64_gpu.py).
//
// The sampling rule is deform_sample.h's, restated in double (the coordinate is formed as double(int base) + offset: the reference's `scalar_t` arithmetic,
// deform_im2col_cuda.cuh:244-247,26-72 with scalar_t = double):
//   guard  q > -1 && q < size on every axis;  corners floor(q) + {0,1}, low valid iff >= 0, high iff <= size - 1;  weights (1 - l | l) per axis;
//   2-D (torchvision 0.12): the sample is guarded the same way, the coordinate weight only by per-corner bounds (`reach`: q >= -1 && q < size).
#include "dlka_kernels.h"

namespace dlka {

template <int NOFF>
struct Tap64 {
    static constexpr int NC = (NOFF == 3) ? 8 : 4;
    int idx[NC];
    double w[NC];
    unsigned ok, cok;
    double fd[2], fh[2], fw[2];
};

template <int NOFF>
__device__ __forceinline__ void setup_tap64(Tap64<NOFF> &s, const double *__restrict__ offp, int No, int bd, int bh, int bw, int D, int H, int W)
{
    double qd = 0., qh, qw;
    bool inside, reach;
    if (NOFF == 3) {
        qd = (double)bd + offp[0]; qh = (double)bh + offp[No]; qw = (double)bw + offp[2 * (long)No];
        inside = (qd > -1.) & (qh > -1.) & (qw > -1.) & (qd < (double)D) & (qh < (double)H) & (qw < (double)W);
        reach = inside;
    } else {
        qh = (double)bh + offp[0]; qw = (double)bw + offp[No];
        reach = (qh >= -1.) & (qw >= -1.) & (qh < (double)H) & (qw < (double)W);
        inside = reach & (qh > -1.) & (qw > -1.);
    }
    if (!reach) { qd = 0.; qh = 0.; qw = 0.; }
    const double fd_ = floor(qd), fh_ = floor(qh), fw_ = floor(qw);
    const int d0 = (int)fd_, h0 = (int)fh_, w0 = (int)fw_;
    const double ld = qd - fd_, lh = qh - fh_, lw = qw - fw_;
    s.fd[0] = 1. - ld; s.fd[1] = ld; s.fh[0] = 1. - lh; s.fh[1] = lh; s.fw[0] = 1. - lw; s.fw[1] = lw;
    unsigned ok = 0, cok = 0;
#pragma unroll
    for (int q = 0; q < Tap64<NOFF>::NC; ++q) {
        const int cd = (NOFF == 3) ? (q >> 2) & 1 : 0, ch = (q >> 1) & 1, cw = q & 1;
        const int zd = d0 + cd, zh = h0 + ch, zw = w0 + cw;
        const bool v = reach && zd >= 0 && zd <= D - 1 && zh >= 0 && zh <= H - 1 && zw >= 0 && zw <= W - 1;
        const bool use = v && inside;
        cok |= (v ? 1u : 0u) << q;
        ok |= (use ? 1u : 0u) << q;
        s.idx[q] = v ? (zd * H + zh) * W + zw : 0;
        s.w[q] = use ? ((NOFF == 3) ? s.fd[cd] * s.fh[ch] * s.fw[cw] : s.fh[ch] * s.fw[cw]) : 0.;
    }
    s.ok = ok;
    s.cok = cok;
}

template <int NOFF>
__device__ __forceinline__ double sample64(const Tap64<NOFF> &s, const double *__restrict__ xp)
{
    double val = 0.;
#pragma unroll
    for (int q = 0; q < Tap64<NOFF>::NC; ++q)
        if ((s.ok >> q) & 1u) val = fma(s.w[q], xp[s.idx[q]], val);
    return val;
}

__device__ __forceinline__ void vox_base(const Geom &g, int v, int &bd, int &bh, int &bw)
{
    const int ow = v % g.Wo, oh = (v / g.Wo) % g.Ho, od = v / (g.Wo * g.Ho);
    bd = od * g.sd - g.pd; bh = oh * g.sh - g.ph; bw = ow * g.sw - g.pw;
}

// out[b][co][v] = bias[co] + sum_{tap, cg} S(b, c, tap, v) * W[co][cg][tap]          one work-item per output element
template <int NOFF>
__global__ __launch_bounds__(256) void deform_fwd_f64_kernel(const double *__restrict__ x, const double *__restrict__ off, const double *__restrict__ w,
                                                             const double *__restrict__ bias, double *__restrict__ out, Geom g)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)g.B * g.Cout * g.No) return;
    const int v = (int)(e % g.No), co = (int)((e / g.No) % g.Cout), b = (int)(e / ((long)g.No * g.Cout));
    const int gi = co / g.Og;
    int bd, bh, bw;
    vox_base(g, v, bd, bh, bw);
    double acc = bias ? bias[co] : 0.;
    Tap64<NOFF> s;
    int tap = 0;
    for (int i = 0; i < g.kd; ++i)
        for (int j = 0; j < g.kh; ++j)
            for (int k = 0; k < g.kw; ++k, ++tap) {
                int cur = -1;
                for (int cg = 0; cg < g.Cg; ++cg) {
                    const int c = gi * g.Cg + cg, dgi = c / g.cpdg;
                    if (dgi != cur) {
                        cur = dgi;
                        setup_tap64<NOFF>(s, off + ((long)(b * g.dg + dgi) * NOFF * g.K + NOFF * tap) * g.No + v, g.No, bd + i * g.dd, bh + j * g.dh, bw + k * g.dw,
                                          g.D, g.H, g.W);
                    }
                    acc = fma(sample64<NOFF>(s, x + (long)(b * g.C + c) * g.Ni), w[((long)co * g.Cg + cg) * g.K + tap], acc);
                }
            }
    out[e] = acc;
}

// one work-item per (b, dg, tap, v): col(c) = sum_o W[o][c][tap] gO[b][o][v];  gOff written (no other work-item owns it), gX by double atomics (cuh:267-405)
template <int NOFF>
__global__ __launch_bounds__(256) void deform_bwd_io_f64_kernel(const double *__restrict__ x, const double *__restrict__ off, const double *__restrict__ w,
                                                                const double *__restrict__ gout, double *__restrict__ gx, double *__restrict__ goff, Geom g)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)g.B * g.dg * g.K * g.No) return;
    const int v = (int)(e % g.No), tap = (int)((e / g.No) % g.K), dgi = (int)((e / ((long)g.No * g.K)) % g.dg), b = (int)(e / ((long)g.No * g.K * g.dg));
    const int k = tap % g.kw, j = (tap / g.kw) % g.kh, i = tap / (g.kw * g.kh);
    int bd, bh, bw;
    vox_base(g, v, bd, bh, bw);
    Tap64<NOFF> s;
    setup_tap64<NOFF>(s, off + ((long)(b * g.dg + dgi) * NOFF * g.K + NOFF * tap) * g.No + v, g.No, bd + i * g.dd, bh + j * g.dh, bw + k * g.dw, g.D, g.H, g.W);
    const unsigned dmask = (NOFF == 3) ? s.ok : s.cok;   // D3D zeroes the coordinate gradient outside the guard (cuh:391-394); torchvision bounds each corner only
    double go_d = 0., go_h = 0., go_w = 0.;
    for (int cc = 0; cc < g.cpdg; ++cc) {
        const int c = dgi * g.cpdg + cc, gi = c / g.Cg, cg = c - gi * g.Cg;
        double col = 0.;
        for (int o = 0; o < g.Og; ++o) col = fma(gout[((long)b * g.Cout + gi * g.Og + o) * g.No + v], w[((long)(gi * g.Og + o) * g.Cg + cg) * g.K + tap], col);
        const double *xp = x + (long)(b * g.C + c) * g.Ni;
        double dd_ = 0., dh_ = 0., dw_ = 0.;
#pragma unroll
        for (int q = 0; q < Tap64<NOFF>::NC; ++q) {
            const int cd = (NOFF == 3) ? (q >> 2) & 1 : 0, ch = (q >> 1) & 1, cw = q & 1;
            const double xv = ((dmask >> q) & 1u) ? xp[s.idx[q]] : 0.;
            if (NOFF == 3) {
                dd_ = fma((cd ? 1. : -1.) * s.fh[ch] * s.fw[cw], xv, dd_);
                dh_ = fma((ch ? 1. : -1.) * s.fd[cd] * s.fw[cw], xv, dh_);
                dw_ = fma((cw ? 1. : -1.) * s.fd[cd] * s.fh[ch], xv, dw_);
            } else {
                dh_ = fma((ch ? 1. : -1.) * s.fw[cw], xv, dh_);
                dw_ = fma((cw ? 1. : -1.) * s.fh[ch], xv, dw_);
            }
            if (gx && ((s.ok >> q) & 1u)) atomicAdd(gx + (long)(b * g.C + c) * g.Ni + s.idx[q], col * s.w[q]);
        }
        go_d = fma(col, dd_, go_d); go_h = fma(col, dh_, go_h); go_w = fma(col, dw_, go_w);
    }
    if (goff) {
        double *gop = goff + ((long)(b * g.dg + dgi) * NOFF * g.K + NOFF * tap) * g.No + v;
        if (NOFF == 3) { gop[0] = go_d; gop[g.No] = go_h; gop[2 * (long)g.No] = go_w; }
        else { gop[0] = go_h; gop[g.No] = go_w; }
    }
}

// gW[co][cg][tap] = sum_{b, v} gO[b][co][v] S(b, c, tap, v): one WORKGROUP per weight element, its 256 work-items stride over (b, v) and meet in LDS in a fixed order
template <int NOFF>
__global__ __launch_bounds__(256) void deform_bwd_w_f64_kernel(const double *__restrict__ x, const double *__restrict__ off, const double *__restrict__ gout,
                                                               double *__restrict__ gw, Geom g)
{
    __shared__ double red[256];
    const long e = blockIdx.x;
    const int tap = (int)(e % g.K), cg = (int)((e / g.K) % g.Cg), co = (int)(e / ((long)g.K * g.Cg));
    const int gi = co / g.Og, c = gi * g.Cg + cg, dgi = c / g.cpdg;
    const int k = tap % g.kw, j = (tap / g.kw) % g.kh, i = tap / (g.kw * g.kh);
    double acc = 0.;
    Tap64<NOFF> s;
    for (long t = threadIdx.x; t < (long)g.B * g.No; t += 256) {
        const int b = (int)(t / g.No), v = (int)(t - (long)b * g.No);
        int bd, bh, bw;
        vox_base(g, v, bd, bh, bw);
        setup_tap64<NOFF>(s, off + ((long)(b * g.dg + dgi) * NOFF * g.K + NOFF * tap) * g.No + v, g.No, bd + i * g.dd, bh + j * g.dh, bw + k * g.dw, g.D, g.H, g.W);
        acc = fma(gout[((long)b * g.Cout + co) * g.No + v], sample64<NOFF>(s, x + (long)(b * g.C + c) * g.Ni), acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) gw[e] = red[0];
}

// gb[co] = sum_{b, v} gO[b][co][v]   (also the plain conv's bias gradient)
__global__ __launch_bounds__(256) void bias_grad_f64_kernel(const double *__restrict__ gout, double *__restrict__ gb, int B, int Cout, int No)
{
    __shared__ double red[256];
    const int co = blockIdx.x;
    double acc = 0.;
    for (long t = threadIdx.x; t < (long)B * No; t += 256) {
        const int b = (int)(t / No), v = (int)(t - (long)b * No);
        acc += gout[((long)b * Cout + co) * No + v];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) gb[co] = red[0];
}

// ---- plain grouped conv, double --------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_fwd_f64_kernel(const double *__restrict__ x, const double *__restrict__ w, const double *__restrict__ bias,
                                                           double *__restrict__ out, Geom g)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)g.B * g.Cout * g.No) return;
    const int v = (int)(e % g.No), co = (int)((e / g.No) % g.Cout), b = (int)(e / ((long)g.No * g.Cout));
    const int gi = co / g.Og;
    int bd, bh, bw;
    vox_base(g, v, bd, bh, bw);
    double acc = bias ? bias[co] : 0.;
    int tap = 0;
    for (int i = 0; i < g.kd; ++i)
        for (int j = 0; j < g.kh; ++j)
            for (int k = 0; k < g.kw; ++k, ++tap) {
                const int zd = bd + i * g.dd, zh = bh + j * g.dh, zw = bw + k * g.dw;
                if ((unsigned)zd >= (unsigned)g.D || (unsigned)zh >= (unsigned)g.H || (unsigned)zw >= (unsigned)g.W) continue;
                const long lin = ((long)zd * g.H + zh) * g.W + zw;
                for (int cg = 0; cg < g.Cg; ++cg) acc = fma(x[(long)(b * g.C + gi * g.Cg + cg) * g.Ni + lin], w[((long)co * g.Cg + cg) * g.K + tap], acc);
            }
    out[e] = acc;
}

// gX[b][c][u] = sum over (co of c's group, tap, output voxel v that reads u through tap) gO[b][co][v] W[co][cg][tap]      gather form, no atomics
__global__ __launch_bounds__(256) void conv_bwd_x_f64_kernel(const double *__restrict__ gout, const double *__restrict__ w, double *__restrict__ gx, Geom g)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)g.B * g.C * g.Ni) return;
    const int u = (int)(e % g.Ni), c = (int)((e / g.Ni) % g.C), b = (int)(e / ((long)g.Ni * g.C));
    const int gi = c / g.Cg, cg = c - gi * g.Cg;
    const int uw = u % g.W, uh = (u / g.W) % g.H, ud = u / (g.W * g.H);
    double acc = 0.;
    int tap = 0;
    for (int i = 0; i < g.kd; ++i)
        for (int j = 0; j < g.kh; ++j)
            for (int k = 0; k < g.kw; ++k, ++tap) {
                const int nd = ud + g.pd - i * g.dd, nh = uh + g.ph - j * g.dh, nw = uw + g.pw - k * g.dw;
                if (nd < 0 || nh < 0 || nw < 0 || nd % g.sd || nh % g.sh || nw % g.sw) continue;
                const int od = nd / g.sd, oh = nh / g.sh, ow = nw / g.sw;
                if (od >= g.Do || oh >= g.Ho || ow >= g.Wo) continue;
                const long v = ((long)od * g.Ho + oh) * g.Wo + ow;
                for (int o = 0; o < g.Og; ++o) acc = fma(gout[((long)b * g.Cout + gi * g.Og + o) * g.No + v], w[((long)(gi * g.Og + o) * g.Cg + cg) * g.K + tap], acc);
            }
    gx[e] = acc;
}

__global__ __launch_bounds__(256) void conv_bwd_w_f64_kernel(const double *__restrict__ x, const double *__restrict__ gout, double *__restrict__ gw, Geom g)
{
    __shared__ double red[256];
    const long e = blockIdx.x;
    const int tap = (int)(e % g.K), cg = (int)((e / g.K) % g.Cg), co = (int)(e / ((long)g.K * g.Cg));
    const int gi = co / g.Og, c = gi * g.Cg + cg;
    const int k = tap % g.kw, j = (tap / g.kw) % g.kh, i = tap / (g.kw * g.kh);
    double acc = 0.;
    for (long t = threadIdx.x; t < (long)g.B * g.No; t += 256) {
        const int b = (int)(t / g.No), v = (int)(t - (long)b * g.No);
        int bd, bh, bw;
        vox_base(g, v, bd, bh, bw);
        const int zd = bd + i * g.dd, zh = bh + j * g.dh, zw = bw + k * g.dw;
        if ((unsigned)zd >= (unsigned)g.D || (unsigned)zh >= (unsigned)g.H || (unsigned)zw >= (unsigned)g.W) continue;
        acc = fma(gout[((long)b * g.Cout + co) * g.No + v], x[(long)(b * g.C + c) * g.Ni + ((long)zd * g.H + zh) * g.W + zw], acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) gw[e] = red[0];
}

// ---- launchers ------------------------------------------------------------------------------------------------------------------------------------
static unsigned blocks_for(long n) { return (unsigned)((n + 255) / 256); }

template <int NOFF>
int launch_deform_fwd_f64(const double *x, const double *off, const double *w, const double *bias, double *out, const Geom &g, hipStream_t st)
{
    const long n = (long)g.B * g.Cout * g.No;
    if (n >= (1l << 39)) return DLKA_ERR_SHAPE;
    auto k = deform_fwd_f64_kernel<NOFF>;
    DLKA_LAUNCH(k, dim3(blocks_for(n)), dim3(256), 0, st, x, off, w, bias, out, g);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

template <int NOFF>
int launch_deform_bwd_f64(const double *x, const double *off, const double *w, const double *gout, double *gx, double *goff, double *gw, double *gb, const Geom &g,
                          hipStream_t st)
{
    if (gx || goff) {
        if (gx && launch_zero(gx, (size_t)g.B * g.C * g.Ni * 8, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
        const long n = (long)g.B * g.dg * g.K * g.No;
        auto k = deform_bwd_io_f64_kernel<NOFF>;
        DLKA_LAUNCH(k, dim3(blocks_for(n)), dim3(256), 0, st, x, off, w, gout, gx, goff, g);
        DLKA_CHECK_LAUNCH();
    }
    if (gw) {
        auto k = deform_bwd_w_f64_kernel<NOFF>;
        DLKA_LAUNCH(k, dim3((unsigned)((long)g.Cout * g.Cg * g.K)), dim3(256), 0, st, x, off, gout, gw, g);
        DLKA_CHECK_LAUNCH();
    }
    if (gb) {
        DLKA_LAUNCH(bias_grad_f64_kernel, dim3(g.Cout), dim3(256), 0, st, gout, gb, g.B, g.Cout, g.No);
        DLKA_CHECK_LAUNCH();
    }
    return DLKA_OK;
}

int launch_conv_fwd_f64(const double *x, const double *w, const double *bias, double *out, const Geom &g, hipStream_t st)
{
    DLKA_LAUNCH(conv_fwd_f64_kernel, dim3(blocks_for((long)g.B * g.Cout * g.No)), dim3(256), 0, st, x, w, bias, out, g);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_conv_bwd_f64(const double *x, const double *w, const double *gout, double *gx, double *gw, double *gb, const Geom &g, hipStream_t st)
{
    if (gx) {
        DLKA_LAUNCH(conv_bwd_x_f64_kernel, dim3(blocks_for((long)g.B * g.C * g.Ni)), dim3(256), 0, st, gout, w, gx, g);
        DLKA_CHECK_LAUNCH();
    }
    if (gw) {
        DLKA_LAUNCH(conv_bwd_w_f64_kernel, dim3((unsigned)((long)g.Cout * g.Cg * g.K)), dim3(256), 0, st, x, gout, gw, g);
        DLKA_CHECK_LAUNCH();
    }
    if (gb) {
        DLKA_LAUNCH(bias_grad_f64_kernel, dim3(g.Cout), dim3(256), 0, st, gout, gb, g.B, g.Cout, g.No);
        DLKA_CHECK_LAUNCH();
    }
    return DLKA_OK;
}

template int launch_deform_fwd_f64<3>(const double *, const double *, const double *, const double *, double *, const Geom &, hipStream_t);
template int launch_deform_fwd_f64<2>(const double *, const double *, const double *, const double *, double *, const Geom &, hipStream_t);
template int launch_deform_bwd_f64<3>(const double *, const double *, const double *, const double *, double *, double *, double *, double *, const Geom &, hipStream_t);
template int launch_deform_bwd_f64<2>(const double *, const double *, const double *, const double *, double *, double *, double *, double *, const Geom &, hipStream_t);

}  // namespace dlka
