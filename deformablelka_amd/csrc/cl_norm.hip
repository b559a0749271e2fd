// The non-convolutional pieces of TransformerBlock_3D_single_deform_LKA (3D/d_lka_former/network_architecture/synapse/
// transformerblock.py:570-630) around the D-LKA block, in token / channels-last layout [M = B*N][C], fp32:
//   tokens + pos_embed + LayerNorm (:620-624)            cl_layernorm_fwd_kernel / cl_layernorm_bwd_kernel
//   x + gamma * epa_block(...) (:624)                    cl_scale_residual_fwd_kernel / _bwd_kernel
//   BatchNorm3d (+ residual) + LeakyReLU of UnetResBlock (dynunet_block.py:66-79)   cl_bn_* kernels
//   Dropout3d's per-(sample, channel) mask (:611)        cl_channel_scale_kernel
// All of them are HBM-bound streaming / reduction kernels: lanes run over channels (contiguous 128-byte row pieces), rows
// are strided over the grid; per-channel reductions fold in LDS and finish with one fp32 atomic per channel and workgroup.
#include "dlka_kernels.h"

namespace dlka {

#define DLKA_TRY_LAUNCH(expr)           \
    do {                                \
        int rc_ = (expr);               \
        if (rc_ != DLKA_OK) return rc_; \
    } while (0)

namespace {
constexpr int NT = 256;
constexpr int KMAX = 4;   // channels per lane: C <= 256 (the D-LKA stage widths are 32 .. 256)
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// sum over the 32-lane half this lane belongs to (C <= 32: a wave carries two token rows)
__device__ __forceinline__ float half_sum(float v)
{
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
}  // namespace

// The tensors that cross into / out of the D-LKA attention block (xn, e, g_e, g_xn) are bf16 storage when the wrapper block runs its attention in
// DLKA_BF16 (`lo` != 0; include/dlka.h: dlka_tblock3d_*, dtype = DLKA_BF16); everything else of the wrapper is fp32.
__device__ __forceinline__ float ld_lo(const float *p, long i, int lo) { return lo ? act_load1(reinterpret_cast<const bf16_t *>(p), i) : p[i]; }
__device__ __forceinline__ void st_lo(float *p, long i, float v, int lo)
{
    if (lo) act_store1(reinterpret_cast<bf16_t *>(p), i, v);
    else p[i] = v;
}

// One wave per token row: x (planar [B][C][N] — the NCDHW tensor the block receives — or channels-last [M][C]) (+ pos[N][C])
// -> xt[M][C]; xn = (xt - mean) * rstd * w + b; stats[m] = {mean, rstd}.   Biased variance, eps inside the sqrt (nn.LayerNorm).
__global__ __launch_bounds__(NT) void cl_layernorm_fwd_kernel(const float *__restrict__ x, int x_planar, const float *__restrict__ pos,
                                                              const float *__restrict__ w, const float *__restrict__ b, float *__restrict__ xt,
                                                              float *__restrict__ xn, float *__restrict__ stats, int B, int N, int C, float eps, int lo,
                                                              float *__restrict__ xn32)
{   // xn32 (optional, with lo): the UNROUNDED LayerNorm output, fp32 — what the mixed mode's offset-determining chain starts from (dlka_tblock3d_forward_v)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long M = (long)B * N;
    if (C <= 32) {   // two token rows per wave: lanes 0-31 / 32-63 (a 32-channel row would leave half the wave idle)
        const int c = lane & 31, hh = lane >> 5;
        for (long m0 = ((long)blockIdx.x * (NT / 64) + wave) * 2; m0 < M; m0 += (long)gridDim.x * (NT / 64) * 2) {
            const long m = m0 + hh;
            const bool ok = m < M && c < C;
            const int bb = ok ? (int)(m / N) : 0, v = ok ? (int)(m - (long)bb * N) : 0;
            float val = 0.f;
            if (ok) {
                val = x_planar ? x[((long)bb * C + c) * N + v] : x[m * C + c];
                if (pos) val += pos[(long)v * C + c];
                xt[m * C + c] = val;
            }
            const float mean = half_sum(val) / C;
            const float var = fmaxf(half_sum(val * val) / C - mean * mean, 0.f);
            const float rstd = 1.f / sqrtf(var + eps);
            if (ok) {
                const float o = (val - mean) * rstd * w[c] + b[c];
                st_lo(xn, m * C + c, o, lo);
                if (xn32) xn32[m * C + c] = o;
                if (c == 0) { stats[2 * m] = mean; stats[2 * m + 1] = rstd; }
            }
        }
        return;
    }
    for (long m = (long)blockIdx.x * (NT / 64) + wave; m < M; m += (long)gridDim.x * (NT / 64)) {
        const int bb = (int)(m / N), v = (int)(m - (long)bb * N);
        float s = 0.f, s2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            float val = x_planar ? x[((long)bb * C + c) * N + v] : x[m * C + c];
            if (pos) val += pos[(long)v * C + c];
            xt[m * C + c] = val;
            s += val;
            s2 = fmaf(val, val, s2);
        }
        s = wave_sum(s);
        s2 = wave_sum(s2);
        const float mean = s / C;
        const float var = fmaxf(s2 / C - mean * mean, 0.f);
        const float rstd = 1.f / sqrtf(var + eps);
        for (int c = lane; c < C; c += 64) {
            const float o = (xt[m * C + c] - mean) * rstd * w[c] + b[c];
            st_lo(xn, m * C + c, o, lo);
            if (xn32) xn32[m * C + c] = o;
        }
        if (lane == 0) { stats[2 * m] = mean; stats[2 * m + 1] = rstd; }
    }
}

// gxt[m][c] = (g_res ? g_res : 0) + rstd * (dxhat - mean_c(dxhat) - xhat * mean_c(dxhat * xhat)),  dxhat = g * w
// gw[c] += sum_m g * xhat, gb[c] += sum_m g   (zero-initialised; one atomic per channel and workgroup)
// gpos[v][c] += gxt[m][c]                     (zero-initialised; optional)
__global__ __launch_bounds__(NT) void cl_layernorm_bwd_kernel(const float *__restrict__ g, const float *__restrict__ g_res, const float *__restrict__ xt,
                                                              const float *__restrict__ stats, const float *__restrict__ w, float *__restrict__ gxt,
                                                              float *__restrict__ gw, float *__restrict__ gb, float *__restrict__ gpos, int B, int N, int C, int lo)
{
    DLKA_DYN_SMEM(float, red);   // [waves][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long M = (long)B * N;
    float aw[KMAX], ab[KMAX];   // this lane's channels lane + 64k: partial sums over the rows of this wave (registers, no LDS atomics)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { aw[k] = 0.f; ab[k] = 0.f; }
    if (C <= 32) {   // two token rows per wave (see the forward kernel); the two halves' channel sums meet in LDS below
        const int c = lane & 31, hh = lane >> 5;
        float a_w = 0.f, a_b = 0.f;
        for (long m0 = ((long)blockIdx.x * (NT / 64) + wave) * 2; m0 < M; m0 += (long)gridDim.x * (NT / 64) * 2) {
            const long m = m0 + hh;
            const bool ok = m < M && c < C;
            float xh = 0.f, dxh = 0.f, rstd = 0.f;
            if (ok) {
                const float gv = ld_lo(g, m * C + c, lo);
                rstd = stats[2 * m + 1];
                xh = (xt[m * C + c] - stats[2 * m]) * rstd;
                dxh = gv * w[c];
                a_w = fmaf(gv, xh, a_w);
                a_b += gv;
            }
            const float s1 = half_sum(dxh) / C, s2 = half_sum(dxh * xh) / C;
            if (ok) {
                float val = rstd * (dxh - s1 - xh * s2);
                if (g_res) val += g_res[m * C + c];
                gxt[m * C + c] = val;
                if (gpos) atomicAdd(gpos + (long)(m % N) * C + c, val);
            }
        }
        a_w += __shfl_xor(a_w, 32);
        a_b += __shfl_xor(a_b, 32);
        if (lane < 32 && c < C) { red[(wave * 2 + 0) * C + c] = a_w; red[(wave * 2 + 1) * C + c] = a_b; }
        __syncthreads();
        for (int cc = threadIdx.x; cc < C; cc += NT) {
            float sw = 0.f, sb = 0.f;
            for (int wv = 0; wv < NT / 64; ++wv) { sw += red[(wv * 2 + 0) * C + cc]; sb += red[(wv * 2 + 1) * C + cc]; }
            atomicAdd(gw + cc, sw);
            atomicAdd(gb + cc, sb);
        }
        return;
    }
    for (long m = (long)blockIdx.x * (NT / 64) + wave; m < M; m += (long)gridDim.x * (NT / 64)) {
        const float mean = stats[2 * m], rstd = stats[2 * m + 1];
        float s1 = 0.f, s2 = 0.f, xh[KMAX], dxh[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = lane + 64 * k;
            xh[k] = 0.f; dxh[k] = 0.f;
            if (c < C) {
                const float gv = ld_lo(g, m * C + c, lo);
                xh[k] = (xt[m * C + c] - mean) * rstd;
                dxh[k] = gv * w[c];
                s1 += dxh[k];
                s2 = fmaf(dxh[k], xh[k], s2);
                aw[k] = fmaf(gv, xh[k], aw[k]);
                ab[k] += gv;
            }
        }
        s1 = wave_sum(s1) / C;
        s2 = wave_sum(s2) / C;
        const int v = (int)(m % N);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = lane + 64 * k;
            if (c < C) {
                float val = rstd * (dxh[k] - s1 - xh[k] * s2);
                if (g_res) val += g_res[m * C + c];
                gxt[m * C + c] = val;
                if (gpos) atomicAdd(gpos + (long)v * C + c, val);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int c = lane + 64 * k;
        if (c < C) { red[(wave * 2 + 0) * C + c] = aw[k]; red[(wave * 2 + 1) * C + c] = ab[k]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += NT) {
        float sw = 0.f, sb = 0.f;
        for (int wv = 0; wv < NT / 64; ++wv) { sw += red[(wv * 2 + 0) * C + c]; sb += red[(wv * 2 + 1) * C + c]; }
        atomicAdd(gw + c, sw);
        atomicAdd(gb + c, sb);
    }
}

// out = xt + gamma[c] * e
__global__ __launch_bounds__(NT) void cl_scale_residual_fwd_kernel(const float *__restrict__ xt, const float *__restrict__ e, const float *__restrict__ gamma,
                                                                   float *__restrict__ out, long M, int C, int lo)
{
    const long n = M * C;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) out[i] = fmaf(gamma[i % C], ld_lo(e, i, lo), xt[i]);
}

// ge = gamma[c] * g;  ggamma[c] += sum_m g * e   (zero-initialised)
__global__ __launch_bounds__(NT) void cl_scale_residual_bwd_kernel(const float *__restrict__ g, const float *__restrict__ e, const float *__restrict__ gamma,
                                                                   float *__restrict__ ge, float *__restrict__ ggamma, long M, int C, int lo)
{
    DLKA_DYN_SMEM(float, red);   // [C]
    for (int c = threadIdx.x; c < C; c += NT) red[c] = 0.f;
    __syncthreads();
    const int cpb = C < NT ? C : NT, rpb = NT / cpb;   // C is a multiple of 32 and <= 1024; rows in flight per pass
    const int c_in = threadIdx.x % cpb, r_in = threadIdx.x / cpb;
    for (int cb = 0; cb < C; cb += cpb) {
        const int c = cb + c_in;
        float acc = 0.f;
        if (r_in < rpb && c < C) {
            const float gm = gamma[c];
            for (long m = (long)blockIdx.x * rpb + r_in; m < M; m += (long)gridDim.x * rpb) {
                const float gv = g[m * C + c];
                st_lo(ge, m * C + c, gm * gv, lo);
                acc = fmaf(gv, ld_lo(e, m * C + c, lo), acc);
            }
            atomicAdd(&red[c], acc);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += NT) atomicAdd(ggamma + c, red[c]);
}

// sums[c] += sum_m (x[m][c] - p_c), sums[C + c] += sum_m (x[m][c] - p_c)^2     (zero-initialised)
// p_c = x[0][c], the first row, is a per-channel pivot: the one-pass E[x^2] - mean^2 form loses all its digits when |mean| >> std
// (fp32: 24 % variance error at mean/std = 500); around a pivot that is itself a sample the two sums stay O(std), so the
// subtraction in cl_bn_finish_stats_kernel cancels nothing that matters (matches torch's Welford BatchNorm to ~1e-6 rel).
__global__ __launch_bounds__(NT) void cl_bn_stats_kernel(const float *__restrict__ x, float *__restrict__ sums, long M, int C)
{
    DLKA_DYN_SMEM(float, red);   // [2][C]
    for (int c = threadIdx.x; c < 2 * C; c += NT) red[c] = 0.f;
    __syncthreads();
    const int cpb = C < NT ? C : NT, rpb = NT / cpb;
    const int c_in = threadIdx.x % cpb, r_in = threadIdx.x / cpb;
    for (int cb = 0; cb < C; cb += cpb) {
        const int c = cb + c_in;
        if (r_in < rpb && c < C) {
            float s = 0.f, s2 = 0.f;
            const float pv = x[c];
            const long step = (long)gridDim.x * rpb;
            long m = (long)blockIdx.x * rpb + r_in;
            for (; m + 3 * step < M; m += 4 * step) {   // four rows in flight per work-item (the loop is pure load latency otherwise)
                const float v0 = x[m * C + c] - pv, v1 = x[(m + step) * C + c] - pv, v2 = x[(m + 2 * step) * C + c] - pv, v3 = x[(m + 3 * step) * C + c] - pv;
                s += (v0 + v1) + (v2 + v3);
                s2 = fmaf(v0, v0, fmaf(v1, v1, fmaf(v2, v2, fmaf(v3, v3, s2))));
            }
            for (; m < M; m += step) {
                const float v = x[m * C + c] - pv;
                s += v;
                s2 = fmaf(v, v, s2);
            }
            atomicAdd(&red[c], s);
            atomicAdd(&red[C + c], s2);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += NT) atomicAdd(sums + c, red[c]);
}

// stats[c] = mean, stats[C + c] = rstd, stats[2C + c] = unbiased variance (for the running estimate)
__global__ void cl_bn_finish_stats_kernel(const float *__restrict__ x, const float *__restrict__ sums, float *__restrict__ stats, long M, int C, float eps)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float dm = sums[c] / (float)M;               // mean - pivot
    const float var = fmaxf(sums[C + c] / (float)M - dm * dm, 0.f);
    const float mean = x[c] + dm;
    stats[c] = mean;
    stats[C + c] = 1.f / sqrtf(var + eps);
    stats[2 * C + c] = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
}

// y = lrelu((x - mean) * rstd * w + b (+ res))
__global__ __launch_bounds__(NT) void cl_bn_apply_kernel(const float *__restrict__ x, const float *__restrict__ res, const float *__restrict__ w,
                                                         const float *__restrict__ b, const float *__restrict__ stats, const float *__restrict__ mask,
                                                         float *__restrict__ y, long M, long N, int C, float slope)
{
    const long n = M * C;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const int c = (int)(i % C);
        float v = (x[i] - stats[c]) * stats[C + c] * w[c] + b[c];
        if (res) v += res[i];
        v = v > 0.f ? v : slope * v;
        if (mask) v *= mask[(i / (N * C)) * C + c];   // Dropout3d of the block output, folded in (mask >= 0: the sign survives where it matters)
        y[i] = v;
    }
}

// gpre = g * lrelu'(y);  sums[c] += sum_m gpre, sums[C + c] += sum_m gpre * xhat   (zero-initialised);  gres = gpre (optional)
__global__ __launch_bounds__(NT) void cl_bn_bwd_reduce_kernel(const float *__restrict__ g, const float *__restrict__ gmask, const float *__restrict__ x,
                                                              const float *__restrict__ y, const float *__restrict__ stats, float *__restrict__ sums,
                                                              float *__restrict__ gres, const float *__restrict__ gres_add, long M, long N, int C, float slope)
{
    DLKA_DYN_SMEM(float, red);   // [2][C]
    for (int c = threadIdx.x; c < 2 * C; c += NT) red[c] = 0.f;
    __syncthreads();
    const int cpb = C < NT ? C : NT, rpb = NT / cpb;
    const int c_in = threadIdx.x % cpb, r_in = threadIdx.x / cpb;
    for (int cb = 0; cb < C; cb += cpb) {
        const int c = cb + c_in;
        if (r_in < rpb && c < C) {
            const float mean = stats[c], rstd = stats[C + c];
            float s = 0.f, s2 = 0.f;
            const long step = (long)gridDim.x * rpb;
            long m = (long)blockIdx.x * rpb + r_in;
            for (; m + step < M; m += 2 * step) {   // two rows (six to eight loads) in flight per work-item
                const long i0 = m * C + c, i1 = (m + step) * C + c;
                const float g0 = g[i0], g1 = g[i1], y0 = y[i0], y1 = y[i1], x0 = x[i0], x1 = x[i1];
                float gp0 = g0 * (y0 > 0.f ? 1.f : slope), gp1 = g1 * (y1 > 0.f ? 1.f : slope);
                if (gmask) { gp0 *= gmask[(m / N) * C + c]; gp1 *= gmask[((m + step) / N) * C + c]; }
                if (gres) {
                    gres[i0] = gres_add ? gp0 + gres_add[i0] : gp0;
                    gres[i1] = gres_add ? gp1 + gres_add[i1] : gp1;
                }
                s += gp0 + gp1;
                s2 = fmaf(gp0, (x0 - mean) * rstd, fmaf(gp1, (x1 - mean) * rstd, s2));
            }
            for (; m < M; m += step) {
                const long i = m * C + c;
                float gp = g[i] * (y[i] > 0.f ? 1.f : slope);
                if (gmask) gp *= gmask[(m / N) * C + c];
                if (gres) gres[i] = gres_add ? gp + gres_add[i] : gp;
                s += gp;
                s2 = fmaf(gp, (x[i] - mean) * rstd, s2);
            }
            atomicAdd(&red[c], s);
            atomicAdd(&red[C + c], s2);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += NT) atomicAdd(sums + c, red[c]);
}

// training: gx = rstd * w * (gpre - sums[c]/M - xhat * sums[C+c]/M);  eval (running statistics): gx = rstd * w * gpre
// gw[c] = sums[C + c], gb[c] = sums[c]   (written by workgroup 0)
__global__ __launch_bounds__(NT) void cl_bn_bwd_apply_kernel(const float *__restrict__ g, const float *__restrict__ gmask, const float *__restrict__ x,
                                                             const float *__restrict__ y, const float *__restrict__ w, const float *__restrict__ stats,
                                                             const float *__restrict__ sums, float *__restrict__ gx, float *__restrict__ gw, float *__restrict__ gb,
                                                             long M, long N, int C, float slope, int training)
{
    const long n = M * C;
    const float invM = 1.f / (float)M;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const int c = (int)(i % C);
        const float rstd = stats[C + c];
        float gp = g[i] * (y[i] > 0.f ? 1.f : slope);
        if (gmask) gp *= gmask[(i / (N * C)) * C + c];
        float v = gp;
        if (training) v -= sums[c] * invM + (x[i] - stats[c]) * rstd * sums[C + c] * invM;
        gx[i] = rstd * w[c] * v;
    }
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += NT) { gw[c] = sums[C + c]; gb[c] = sums[c]; }
}

// y[m][c] = x[m][c] * mask[b][c]
__global__ __launch_bounds__(NT) void cl_channel_scale_kernel(const float *__restrict__ x, const float *__restrict__ mask, float *__restrict__ y, int B, long N,
                                                              int C)
{
    const long n = (long)B * N * C;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const int c = (int)(i % C), b = (int)(i / ((long)N * C));
        y[i] = x[i] * mask[(long)b * C + c];
    }
}

static unsigned grid_for(long work_items, long per_block, long cap = 2048)
{
    long g = cdivl(work_items, per_block);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

int launch_cl_layernorm_fwd(const float *x, int x_planar, const float *pos, const float *w, const float *b, float *xt, float *xn, float *stats, int B, int N,
                            int C, float eps, hipStream_t st, int lo, float *xn32)
{
    if (C > 64 * KMAX) return DLKA_ERR_UNSUPPORTED;
    DLKA_LAUNCH(cl_layernorm_fwd_kernel, dim3(grid_for((long)B * N, NT / 64, 4096)), dim3(NT), 0, st, x, x_planar, pos, w, b, xt, xn, stats, B, N, C, eps, lo, xn32);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_layernorm_bwd(const float *g, const float *g_res, const float *xt, const float *stats, const float *w, float *gxt, float *gw, float *gb,
                            float *gpos, int B, int N, int C, hipStream_t st, bool zeroed, int lo)
{
    if (C > 64 * KMAX) return DLKA_ERR_UNSUPPORTED;
    if (!zeroed) {   // (the fused block zero-fills every accumulation target of a direction with one launch)
        DLKA_TRY_LAUNCH(launch_zero(gw, (size_t)C * 4, st));
        DLKA_TRY_LAUNCH(launch_zero(gb, (size_t)C * 4, st));
        if (gpos) DLKA_TRY_LAUNCH(launch_zero(gpos, (size_t)N * C * 4, st));
    }
    DLKA_LAUNCH(cl_layernorm_bwd_kernel, dim3(grid_for((long)B * N, NT / 64, 1024)), dim3(NT), (NT / 64) * 2 * C * sizeof(float), st, g, g_res, xt, stats, w, gxt, gw, gb,
                       gpos, B, N, C, lo);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_scale_residual_fwd(const float *xt, const float *e, const float *gamma, float *out, long M, int C, hipStream_t st, int lo)
{
    DLKA_LAUNCH(cl_scale_residual_fwd_kernel, dim3(grid_for(M * C, NT)), dim3(NT), 0, st, xt, e, gamma, out, M, C, lo);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_scale_residual_bwd(const float *g, const float *e, const float *gamma, float *ge, float *ggamma, long M, int C, hipStream_t st, bool zeroed, int lo)
{
    if (!zeroed) DLKA_TRY_LAUNCH(launch_zero(ggamma, (size_t)C * 4, st));
    const int rpb = NT / (C < NT ? C : NT);
    DLKA_LAUNCH(cl_scale_residual_bwd_kernel, dim3(grid_for(M, rpb * 16, 1024)), dim3(NT), C * sizeof(float), st, g, e, gamma, ge, ggamma, M, C, lo);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// sums: 2C floats of scratch; stats: 3C floats {mean, rstd, unbiased var}
int launch_cl_bn_stats(const float *x, float *sums, float *stats, long M, int C, float eps, hipStream_t st, bool zeroed)
{
    if (!zeroed) DLKA_TRY_LAUNCH(launch_zero(sums, (size_t)2 * C * 4, st));
    const int rpb = NT / (C < NT ? C : NT);
    DLKA_LAUNCH(cl_bn_stats_kernel, dim3(grid_for(M, rpb * 16, 1024)), dim3(NT), 2 * C * sizeof(float), st, x, sums, M, C);
    DLKA_CHECK_LAUNCH();
    DLKA_LAUNCH(cl_bn_finish_stats_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, x, (const float *)sums, stats, M, C, eps);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_bn_apply(const float *x, const float *res, const float *w, const float *b, const float *stats, const float *mask, float *y, long M, long N, int C,
                       float slope, hipStream_t st)
{
    DLKA_LAUNCH(cl_bn_apply_kernel, dim3(grid_for(M * C, NT)), dim3(NT), 0, st, x, res, w, b, stats, mask, y, M, N, C, slope);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_bn_bwd(const float *g, const float *gmask, const float *x, const float *y, const float *w, const float *stats, float *sums, float *gx, float *gres,
                     const float *gres_add, float *gw, float *gb, long M, long N, int C, float slope, int training, hipStream_t st, bool zeroed)
{
    if (!zeroed) DLKA_TRY_LAUNCH(launch_zero(sums, (size_t)2 * C * 4, st));
    const int rpb = NT / (C < NT ? C : NT);
    DLKA_LAUNCH(cl_bn_bwd_reduce_kernel, dim3(grid_for(M, rpb * 16, 1024)), dim3(NT), 2 * C * sizeof(float), st, g, gmask, x, y, stats, sums, gres, gres_add, M, N, C, slope);
    DLKA_CHECK_LAUNCH();
    DLKA_LAUNCH(cl_bn_bwd_apply_kernel, dim3(grid_for(M * C, NT)), dim3(NT), 0, st, g, gmask, x, y, w, stats, (const float *)sums, gx, gw, gb, M, N, C, slope, training);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_channel_scale(const float *x, const float *mask, float *y, int B, long N, int C, hipStream_t st)
{
    DLKA_LAUNCH(cl_channel_scale_kernel, dim3(grid_for((long)B * N * C, NT)), dim3(NT), 0, st, x, mask, y, B, N, C);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

}  // namespace dlka
