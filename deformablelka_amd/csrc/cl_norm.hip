// The non-convolutional pieces of TransformerBlock_3D_single_deform_LKA (3D/d_lka_former/network_architecture/synapse/
// transformerblock.py:570-630) around the D-LKA block, in token / channels-last layout [M = B*N][C], fp32:
//   tokens + pos_embed + LayerNorm (:620-624)            cl_layernorm_fwd_kernel / cl_layernorm_bwd_kernel
//   x + gamma * epa_block(...) (:624)                    cl_scale_residual_fwd_kernel / _bwd_kernel
//   BatchNorm3d (+ residual) + LeakyReLU of UnetResBlock (dynunet_block.py:66-79)   cl_bn_* kernels
//   Dropout3d's per-(sample, channel) mask (:611)        cl_channel_scale_kernel
// All of them are HBM-bound streaming / reduction kernels: lanes run over channels (contiguous 128-byte row pieces), rows
// are strided over the grid; per-channel reductions fold in LDS and finish with one fp32 atomic per channel and workgroup.
#include <stdint.h>

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

// =====================================================================================================================================
// Round 5 — the same operators on 16-BYTE accesses.  The kernels above give a lane ONE channel (dword loads, a modulo per element, 0.5 - 1.6 TB/s at
// the 32^3 stage: profiles/r05k_tblock_stage0_kernel_stats.csv: cl_bn_stats 16.4 us for 8.4 MB, cl_bn_bwd_reduce 21.4 us, cl_layernorm_bwd 28.5 us);
// here a lane owns a QUAD of consecutive channels of a row: LPR = C / 4 lanes per row (8 .. 64 for the block's widths 32 .. 256), NT / LPR rows per
// workgroup pass, the channel quad of a thread never changes (per-channel parameters live in registers, no modulo anywhere), row sums of LayerNorm are
// xor-shuffles over the LPR lanes of a row, per-channel sums stay in registers over the thread's rows and meet in LDS once per workgroup.
// Taken for C in {32, 64, 128, 256} and channels-last input; everything else keeps the kernels above.
// =====================================================================================================================================
namespace {
__device__ __forceinline__ bool quad_shape_ok_dev(int C) { return C == 32 || C == 64 || C == 128 || C == 256; }
__device__ __forceinline__ f32x4 ldq_lo(const float *p, long i, int lo) { return lo ? act_load4(reinterpret_cast<const bf16_t *>(p), i) : act_load4(p, i); }
__device__ __forceinline__ void stq_lo(float *p, long i, f32x4 v, int lo)
{
    if (lo) act_store4(reinterpret_cast<bf16_t *>(p), i, v);
    else act_store4(p, i, v);
}
template <int LPR> __device__ __forceinline__ float row_sum(float v)
{
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// per-channel partial sums of a workgroup's threads -> one global atomic per channel: red = [NS][C] floats of LDS (zeroed here)
template <int NS> __device__ __forceinline__ void quad_fold(float *red, const f32x4 *acc, int q, int C, float *const *dst)
{
    for (int c = threadIdx.x; c < NS * C; c += NT) red[c] = 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&red[k * C + 4 * q + e], acc[k][e]);
    __syncthreads();
    for (int c = threadIdx.x; c < NS * C; c += NT) atomicAdd(dst[c / C] + (c % C), red[c]);
}
}  // namespace

template <int LPR>
__global__ __launch_bounds__(NT) void cl_layernorm_fwd_q_kernel(const float *__restrict__ x, const float *__restrict__ pos, const float *__restrict__ w,
                                                                const float *__restrict__ b, float *__restrict__ xt, float *__restrict__ xn,
                                                                float *__restrict__ stats, long M, int N, float eps, int lo, float *__restrict__ xn32)
{
    constexpr int C = 4 * LPR, RPB = NT / LPR;
    const int q = threadIdx.x % LPR, r = threadIdx.x / LPR;
    const f32x4 wq = act_load4(w, 4 * q), bq = act_load4(b, 4 * q);
    for (int m0 = blockIdx.x * RPB; m0 < (int)M; m0 += gridDim.x * RPB) {   // uniform trip count per workgroup: the row sums are wave collectives (32-bit row
        const int m = m0 + r;                                               // counters: the launchers take these kernels for M < 2^30 only)
        const bool ok = m < (int)M;
        const long i = (long)(ok ? m : 0) * C + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            v = act_load4(x, i);
            if (pos) { const f32x4 pv = act_load4(pos, (long)(m % N) * C + 4 * q); v[0] += pv[0]; v[1] += pv[1]; v[2] += pv[2]; v[3] += pv[3]; }
            act_store4(xt, i, v);
        }
        const float s = row_sum<LPR>((v[0] + v[1]) + (v[2] + v[3]));
        const float s2 = row_sum<LPR>(fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3]))));
        const float mean = s / C;
        const float var = fmaxf(s2 / C - mean * mean, 0.f);
        const float rstd = 1.f / sqrtf(var + eps);
        if (!ok) continue;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * wq[e] + bq[e];
        stq_lo(xn, i, o, lo);
        if (xn32) act_store4(xn32, i, o);
        if (q == 0) { stats[2 * m] = mean; stats[2 * m + 1] = rstd; }
    }
}

template <int LPR>
__global__ __launch_bounds__(NT) void cl_layernorm_bwd_q_kernel(const float *__restrict__ g, const float *__restrict__ g_res, const float *__restrict__ xt,
                                                                const float *__restrict__ stats, const float *__restrict__ w, float *__restrict__ gxt,
                                                                float *__restrict__ gw, float *__restrict__ gb, float *__restrict__ gpos, long M, int N, int lo)
{
    constexpr int C = 4 * LPR, RPB = NT / LPR;
    __shared__ float red[2 * C];
    const int q = threadIdx.x % LPR, r = threadIdx.x / LPR;
    const f32x4 wq = act_load4(w, 4 * q);
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // gw, gb of this thread's quad over its rows
    // gpos[v][c] = sum over the batch of gxt[b][v][c].  Round 6: the work-item that owns voxel row v walks the B batch elements itself and STORES the sum — one fp32 atomic per
    // element of grad_x made this kernel 38.8 us at (2, 32^3, 32) against 8.5 us for the forward kernel; gpos needs no zero fill on this path and its sum has a fixed order.
    const int rows = gpos ? N : (int)M, nb = gpos ? (int)(M / N) : 1;
    for (int m0 = blockIdx.x * RPB; m0 < rows; m0 += gridDim.x * RPB) {   // (uniform trip count: see the forward kernel)
        const int v = m0 + r;
        const bool ok = v < rows;
        f32x4 pv = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < nb; ++b) {
            const int m = b * rows + (ok ? v : 0);
            const long i = (long)m * C + 4 * q;
            f32x4 gv = {0.f, 0.f, 0.f, 0.f}, xv = gv;
            float mean = 0.f, rstd = 0.f;
            if (ok) { gv = ldq_lo(g, i, lo); xv = act_load4(xt, i); mean = stats[2 * m]; rstd = stats[2 * m + 1]; }
            f32x4 xh, dxh;
            float p1 = 0.f, p2 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[e] = (xv[e] - mean) * rstd;
                dxh[e] = gv[e] * wq[e];
                p1 += dxh[e];
                p2 = fmaf(dxh[e], xh[e], p2);
                acc[0][e] = fmaf(gv[e], xh[e], acc[0][e]);
                acc[1][e] += gv[e];
            }
            const float s1 = row_sum<LPR>(p1) / C, s2 = row_sum<LPR>(p2) / C;
            if (!ok) continue;
            f32x4 val;
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] = rstd * (dxh[e] - s1 - xh[e] * s2);
            if (g_res) { const f32x4 rv = act_load4(g_res, i); val[0] += rv[0]; val[1] += rv[1]; val[2] += rv[2]; val[3] += rv[3]; }
            act_store4(gxt, i, val);
            pv[0] += val[0]; pv[1] += val[1]; pv[2] += val[2]; pv[3] += val[3];
        }
        if (gpos && ok) act_store4(gpos, (long)v * C + 4 * q, pv);
    }
    float *const dst[2] = {gw, gb};
    quad_fold<2>(red, acc, q, C, dst);
}

// out = xt + gamma[c] * e
template <int LPR>
__global__ __launch_bounds__(NT) void cl_scale_residual_fwd_q_kernel(const float *__restrict__ xt, const float *__restrict__ e, const float *__restrict__ gamma,
                                                                     float *__restrict__ out, long M, int lo)
{
    constexpr int C = 4 * LPR, RPB = NT / LPR;
    const int q = threadIdx.x % LPR, r = threadIdx.x / LPR;
    const f32x4 gq = act_load4(gamma, 4 * q);
    for (int m = blockIdx.x * RPB + r; m < (int)M; m += gridDim.x * RPB) {   // (32-bit row counters: the launchers take these kernels for M < 2^31 only)
        const long i = (long)m * C + 4 * q;
        const f32x4 ev = ldq_lo(e, i, lo), xv = act_load4(xt, i);
        act_store4(out, i, f32x4{fmaf(gq[0], ev[0], xv[0]), fmaf(gq[1], ev[1], xv[1]), fmaf(gq[2], ev[2], xv[2]), fmaf(gq[3], ev[3], xv[3])});
    }
}

// ge = gamma[c] * g;  ggamma[c] += sum_m g * e
template <int LPR>
__global__ __launch_bounds__(NT) void cl_scale_residual_bwd_q_kernel(const float *__restrict__ g, const float *__restrict__ e, const float *__restrict__ gamma,
                                                                     float *__restrict__ ge, float *__restrict__ ggamma, long M, int lo)
{
    constexpr int C = 4 * LPR, RPB = NT / LPR;
    __shared__ float red[C];
    const int q = threadIdx.x % LPR, r = threadIdx.x / LPR;
    const f32x4 gq = act_load4(gamma, 4 * q);
    f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
    for (int m = blockIdx.x * RPB + r; m < (int)M; m += gridDim.x * RPB) {   // (32-bit row counters: the launchers take these kernels for M < 2^31 only)
        const long i = (long)m * C + 4 * q;
        const f32x4 gv = act_load4(g, i), ev = ldq_lo(e, i, lo);
        stq_lo(ge, i, f32x4{gq[0] * gv[0], gq[1] * gv[1], gq[2] * gv[2], gq[3] * gv[3]}, lo);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[0][k] = fmaf(gv[k], ev[k], acc[0][k]);
    }
    float *const dst[1] = {ggamma};
    quad_fold<1>(red, acc, q, C, dst);
}

// sums[c] += sum_m (x[m][c] - p_c), sums[C + c] += sum_m (x[m][c] - p_c)^2, p_c = x[0][c]   (see cl_bn_stats_kernel)
template <int LPR>
__global__ __launch_bounds__(NT) void cl_bn_stats_q_kernel(const float *__restrict__ x, float *__restrict__ sums, long M)
{
    constexpr int C = 4 * LPR, RPB = NT / LPR;
    __shared__ float red[2 * C];
    const int q = threadIdx.x % LPR, r = threadIdx.x / LPR;
    const f32x4 pv = act_load4(x, 4 * q);
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int step = gridDim.x * RPB, Mi = (int)M;
    int m = blockIdx.x * RPB + r;
    for (; (long)m + 3l * step < M; m += 4 * step) {   // four rows (64 bytes) in flight per work-item
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = act_load4(x, (long)(m + u * step) * C + 4 * q);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[u][e] - pv[e]; acc[0][e] += d; acc[1][e] = fmaf(d, d, acc[1][e]); }
    }
    for (; m < Mi; m += step) {
        const f32x4 v = act_load4(x, (long)m * C + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - pv[e]; acc[0][e] += d; acc[1][e] = fmaf(d, d, acc[1][e]); }
    }
    float *const dst[2] = {sums, sums + C};
    quad_fold<2>(red, acc, q, C, dst);
}

// DETERMINISTIC batch statistics (round 6): the same row walk, but the threads' partial sums meet in LDS in ROW-GROUP ORDER (no LDS atomics) and the workgroup writes its 2C sums
// to part[blockIdx.x][2C] (no global atomics, no zero fill); cl_bn_finish_stats_det_kernel adds the workgroups' partials in workgroup order.  The wrapper block's training-mode forward
// is then bitwise reproducible from run to run, as its eval-mode forward and the D-LKA block inside it already are.
template <int LPR>
__global__ __launch_bounds__(NT) void cl_bn_stats_det_q_kernel(const float *__restrict__ x, float *__restrict__ part, long M)
{
    constexpr int C = 4 * LPR, RPB = NT / LPR;
    __shared__ __attribute__((aligned(16))) float red[RPB][2 * C];
    const int q = threadIdx.x % LPR, r = threadIdx.x / LPR;
    const f32x4 pv = act_load4(x, 4 * q);
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int step = gridDim.x * RPB, Mi = (int)M;
    int m = blockIdx.x * RPB + r;
    for (; (long)m + 3l * step < M; m += 4 * step) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = act_load4(x, (long)(m + u * step) * C + 4 * q);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[u][e] - pv[e]; acc[0][e] += d; acc[1][e] = fmaf(d, d, acc[1][e]); }
    }
    for (; m < Mi; m += step) {
        const f32x4 v = act_load4(x, (long)m * C + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - pv[e]; acc[0][e] += d; acc[1][e] = fmaf(d, d, acc[1][e]); }
    }
    *reinterpret_cast<f32x4 *>(&red[r][4 * q]) = acc[0];
    *reinterpret_cast<f32x4 *>(&red[r][C + 4 * q]) = acc[1];
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += NT) {
        float t = 0.f;
        for (int rr = 0; rr < RPB; ++rr) t += red[rr][c];   // fixed order
        part[(long)blockIdx.x * 2 * C + c] = t;
    }
}

// 256 threads = 32 channels x 8 partial groups: group g adds the workgroup partials g, g + 8, ... in that order, the eight group sums meet in LDS and are added in group order —
// a fixed summation tree, eight loads in flight per channel instead of one dependent chain over all partials (a single thread per channel measured 2.5 % on the wrapper-block stack)
__global__ __launch_bounds__(256) void cl_bn_finish_stats_det_kernel(const float *__restrict__ x, const float *__restrict__ part, int nparts, float *__restrict__ stats, long M, int C,
                                                                     float eps)
{
    __shared__ float red[8][64];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
#pragma unroll 8   // (eight partial pairs in flight: the one workgroup of this launch is pure load latency — 9.6 us at 256 partials with the rolled loop; same order of additions)
        for (int w = g; w < nparts; w += 8) { s1 += part[(long)w * 2 * C + c]; s2 += part[(long)w * 2 * C + C + c]; }
    }
    red[g][cl] = s1;
    red[g][32 + cl] = s2;
    __syncthreads();
    if (g != 0 || c >= C) return;
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1 += red[k][cl]; s2 += red[k][32 + cl]; }
    const float dm = s1 / (float)M;
    const float var = fmaxf(s2 / (float)M - dm * dm, 0.f);
    stats[c] = x[c] + dm;
    stats[C + c] = 1.f / sqrtf(var + eps);
    stats[2 * C + c] = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
}

// y = lrelu((x - mean) * rstd * w + b (+ res)) (* mask[b][c])
template <int LPR>
__global__ __launch_bounds__(NT) void cl_bn_apply_q_kernel(const float *__restrict__ x, const float *__restrict__ res, const float *__restrict__ w,
                                                           const float *__restrict__ b, const float *__restrict__ stats, const float *__restrict__ mask,
                                                           float *__restrict__ y, long M, long N, float slope)
{
    constexpr int C = 4 * LPR, RPB = NT / LPR;
    const int q = threadIdx.x % LPR, r = threadIdx.x / LPR;
    const f32x4 mean = act_load4(stats, 4 * q), rstd = act_load4(stats, C + 4 * q), wq = act_load4(w, 4 * q), bq = act_load4(b, 4 * q);
    for (int m = blockIdx.x * RPB + r; m < (int)M; m += gridDim.x * RPB) {   // (32-bit row counters: the launchers take these kernels for M < 2^31 only)
        const long i = (long)m * C + 4 * q;
        const f32x4 xv = act_load4(x, i);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (xv[e] - mean[e]) * rstd[e] * wq[e] + bq[e];   // (the dword kernel's rounding sequence)
        if (res) { const f32x4 rv = act_load4(res, i); v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : slope * v[e];
        if (mask) { const f32x4 mk = act_load4(mask, (long)(m / (int)N) * C + 4 * q); v[0] *= mk[0]; v[1] *= mk[1]; v[2] *= mk[2]; v[3] *= mk[3]; }
        act_store4(y, i, v);
    }
}

// gpre = g * lrelu'(y) (* mask);  sums[c] += sum_m gpre, sums[C + c] += sum_m gpre * xhat;  gres = gpre (+ gres_add)   (optional)
template <int LPR>
__global__ __launch_bounds__(NT) void cl_bn_bwd_reduce_q_kernel(const float *__restrict__ g, const float *__restrict__ gmask, const float *__restrict__ x,
                                                                const float *__restrict__ y, const float *__restrict__ stats, float *__restrict__ sums,
                                                                float *__restrict__ gres, const float *__restrict__ gres_add, long M, long N, float slope)
{
    constexpr int C = 4 * LPR, RPB = NT / LPR;
    __shared__ float red[2 * C];
    const int q = threadIdx.x % LPR, r = threadIdx.x / LPR;
    const f32x4 mean = act_load4(stats, 4 * q), rstd = act_load4(stats, C + 4 * q);
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int step = gridDim.x * RPB, Mi = (int)M;
    int m = blockIdx.x * RPB + r;
    auto one = [&](int mm, const f32x4 &gv, const f32x4 &yv, const f32x4 &xv) {
        f32x4 gp;
#pragma unroll
        for (int e = 0; e < 4; ++e) gp[e] = gv[e] * (yv[e] > 0.f ? 1.f : slope);
        if (gmask) { const f32x4 mk = act_load4(gmask, (long)(mm / (int)N) * C + 4 * q); gp[0] *= mk[0]; gp[1] *= mk[1]; gp[2] *= mk[2]; gp[3] *= mk[3]; }
        if (gres) {
            f32x4 o = gp;
            if (gres_add) { const f32x4 av = act_load4(gres_add, (long)mm * C + 4 * q); o[0] += av[0]; o[1] += av[1]; o[2] += av[2]; o[3] += av[3]; }
            act_store4(gres, (long)mm * C + 4 * q, o);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[0][e] += gp[e]; acc[1][e] = fmaf(gp[e], (xv[e] - mean[e]) * rstd[e], acc[1][e]); }
    };
    for (; (long)m + step < M; m += 2 * step) {   // two rows (six 16-byte loads) in flight per work-item
        const long i0 = (long)m * C + 4 * q, i1 = (long)(m + step) * C + 4 * q;
        const f32x4 g0 = act_load4(g, i0), g1 = act_load4(g, i1), y0 = act_load4(y, i0), y1 = act_load4(y, i1), x0 = act_load4(x, i0), x1 = act_load4(x, i1);
        one(m, g0, y0, x0);
        one(m + step, g1, y1, x1);
    }
    for (; m < Mi; m += step) {
        const long i = (long)m * C + 4 * q;
        one(m, act_load4(g, i), act_load4(y, i), act_load4(x, i));
    }
    float *const dst[2] = {sums, sums + C};
    quad_fold<2>(red, acc, q, C, dst);
}

// training: gx = rstd * w * (gpre - sums[c]/M - xhat * sums[C+c]/M);  eval: gx = rstd * w * gpre;  gw[c] = sums[C + c], gb[c] = sums[c] (workgroup 0)
template <int LPR>
__global__ __launch_bounds__(NT) void cl_bn_bwd_apply_q_kernel(const float *__restrict__ g, const float *__restrict__ gmask, const float *__restrict__ x,
                                                               const float *__restrict__ y, const float *__restrict__ w, const float *__restrict__ stats,
                                                               const float *__restrict__ sums, float *__restrict__ gx, float *__restrict__ gw, float *__restrict__ gb,
                                                               long M, long N, float slope, int training)
{
    constexpr int C = 4 * LPR, RPB = NT / LPR;
    const int q = threadIdx.x % LPR, r = threadIdx.x / LPR;
    const float invM = 1.f / (float)M;
    const f32x4 mean = act_load4(stats, 4 * q), rstd = act_load4(stats, C + 4 * q), wq = act_load4(w, 4 * q);
    const f32x4 s1 = act_load4(sums, 4 * q), s2 = act_load4(sums, C + 4 * q);
    for (int m = blockIdx.x * RPB + r; m < (int)M; m += gridDim.x * RPB) {   // (32-bit row counters: the launchers take these kernels for M < 2^31 only)
        const long i = (long)m * C + 4 * q;
        const f32x4 gv = act_load4(g, i), yv = act_load4(y, i);
        f32x4 gp;
#pragma unroll
        for (int e = 0; e < 4; ++e) gp[e] = gv[e] * (yv[e] > 0.f ? 1.f : slope);
        if (gmask) { const f32x4 mk = act_load4(gmask, (long)(m / (int)N) * C + 4 * q); gp[0] *= mk[0]; gp[1] *= mk[1]; gp[2] *= mk[2]; gp[3] *= mk[3]; }
        f32x4 v = gp;
        if (training) {
            const f32x4 xv = act_load4(x, i);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] -= s1[e] * invM + (xv[e] - mean[e]) * rstd[e] * s2[e] * invM;
        }
        act_store4(gx, i, f32x4{rstd[0] * wq[0] * v[0], rstd[1] * wq[1] * v[1], rstd[2] * wq[2] * v[2], rstd[3] * wq[3] * v[3]});
    }
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += NT) { gw[c] = sums[C + c]; gb[c] = sums[c]; }
}

static bool quad_shape_ok(int C, long M = 0) { return (C == 32 || C == 64 || C == 128 || C == 256) && M < (1l << 30); }
static bool quad_aligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// rows per workgroup pass at width C; grids of the streaming kernels: ~4 passes per workgroup, at most `cap` workgroups
constexpr long FOLD_WGS = 256;   // grid cap of the quad kernels that end in a workgroup fold + atomics on per-channel sums (see launch_cl_bn_bwd)
static unsigned quad_grid(long M, int C, int passes, long cap)
{
    const long rpb = NT / (C / 4);
    long gsz = (M + rpb * passes - 1) / (rpb * passes);
    if (gsz > cap) gsz = cap;
    if (gsz < 1) gsz = 1;
    return (unsigned)gsz;
}
#define DLKA_QUAD_DISPATCH(C_, KERNEL, GRID, ...)                                                              \
    switch (C_) {                                                                                              \
    case 32: { auto k_ = KERNEL<8>; DLKA_LAUNCH(k_, dim3(GRID), dim3(NT), 0, st, __VA_ARGS__); } break;        \
    case 64: { auto k_ = KERNEL<16>; DLKA_LAUNCH(k_, dim3(GRID), dim3(NT), 0, st, __VA_ARGS__); } break;       \
    case 128: { auto k_ = KERNEL<32>; DLKA_LAUNCH(k_, dim3(GRID), dim3(NT), 0, st, __VA_ARGS__); } break;      \
    default: { auto k_ = KERNEL<64>; DLKA_LAUNCH(k_, dim3(GRID), dim3(NT), 0, st, __VA_ARGS__); } break;       \
    }

int launch_cl_layernorm_fwd(const float *x, int x_planar, const float *pos, const float *w, const float *b, float *xt, float *xn, float *stats, int B, int N,
                            int C, float eps, hipStream_t st, int lo, float *xn32)
{
    if (C > 64 * KMAX) return DLKA_ERR_UNSUPPORTED;
    if (!x_planar && quad_shape_ok(C, (long)B * N) && quad_aligned(x) && quad_aligned(xt) && quad_aligned(xn) && quad_aligned(pos) && quad_aligned(xn32) && quad_aligned(w) && quad_aligned(b)) {
        const long M = (long)B * N;
        DLKA_QUAD_DISPATCH(C, cl_layernorm_fwd_q_kernel, quad_grid(M, C, 2, 4096), x, pos, w, b, xt, xn, stats, M, N, eps, lo, xn32)
        DLKA_CHECK_LAUNCH();
        return DLKA_OK;
    }
    DLKA_LAUNCH(cl_layernorm_fwd_kernel, dim3(grid_for((long)B * N, NT / 64, 4096)), dim3(NT), 0, st, x, x_planar, pos, w, b, xt, xn, stats, B, N, C, eps, lo, xn32);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_layernorm_bwd(const float *g, const float *g_res, const float *xt, const float *stats, const float *w, float *gxt, float *gw, float *gb,
                            float *gpos, int B, int N, int C, hipStream_t st, bool zeroed, int lo)
{
    if (C > 64 * KMAX) return DLKA_ERR_UNSUPPORTED;
    const bool quad = quad_shape_ok(C, (long)B * N) && quad_aligned(g) && quad_aligned(g_res) && quad_aligned(xt) && quad_aligned(gxt) && quad_aligned(w) && quad_aligned(gpos);
    if (!zeroed) {   // (the fused block zero-fills every accumulation target of a direction with one launch)
        DLKA_TRY_LAUNCH(launch_zero(gw, (size_t)C * 4, st));
        DLKA_TRY_LAUNCH(launch_zero(gb, (size_t)C * 4, st));
        if (gpos && !quad) DLKA_TRY_LAUNCH(launch_zero(gpos, (size_t)N * C * 4, st));   // (the quad kernel stores gpos)
    }
    if (quad) {
        const long M = (long)B * N;
        DLKA_QUAD_DISPATCH(C, cl_layernorm_bwd_q_kernel, quad_grid(gpos ? (long)N : M, C, gpos ? 2 : 4, FOLD_WGS), g, g_res, xt, stats, w, gxt, gw, gb, gpos, M, N, lo)
        DLKA_CHECK_LAUNCH();
        return DLKA_OK;
    }
    DLKA_LAUNCH(cl_layernorm_bwd_kernel, dim3(grid_for((long)B * N, NT / 64, 1024)), dim3(NT), (NT / 64) * 2 * C * sizeof(float), st, g, g_res, xt, stats, w, gxt, gw, gb,
                       gpos, B, N, C, lo);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_scale_residual_fwd(const float *xt, const float *e, const float *gamma, float *out, long M, int C, hipStream_t st, int lo)
{
    if (quad_shape_ok(C, M) && quad_aligned(xt) && quad_aligned(e) && quad_aligned(out) && quad_aligned(gamma)) {
        DLKA_QUAD_DISPATCH(C, cl_scale_residual_fwd_q_kernel, quad_grid(M, C, 2, 4096), xt, e, gamma, out, M, lo)
        DLKA_CHECK_LAUNCH();
        return DLKA_OK;
    }
    DLKA_LAUNCH(cl_scale_residual_fwd_kernel, dim3(grid_for(M * C, NT)), dim3(NT), 0, st, xt, e, gamma, out, M, C, lo);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_scale_residual_bwd(const float *g, const float *e, const float *gamma, float *ge, float *ggamma, long M, int C, hipStream_t st, bool zeroed, int lo)
{
    if (!zeroed) DLKA_TRY_LAUNCH(launch_zero(ggamma, (size_t)C * 4, st));
    if (quad_shape_ok(C, M) && quad_aligned(g) && quad_aligned(e) && quad_aligned(ge) && quad_aligned(gamma)) {
        DLKA_QUAD_DISPATCH(C, cl_scale_residual_bwd_q_kernel, quad_grid(M, C, 4, FOLD_WGS), g, e, gamma, ge, ggamma, M, lo)
        DLKA_CHECK_LAUNCH();
        return DLKA_OK;
    }
    const int rpb = NT / (C < NT ? C : NT);
    DLKA_LAUNCH(cl_scale_residual_bwd_kernel, dim3(grid_for(M, rpb * 16, 1024)), dim3(NT), C * sizeof(float), st, g, e, gamma, ge, ggamma, M, C, lo);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// sums: 2C floats of scratch; stats: 3C floats {mean, rstd, unbiased var}
int launch_cl_bn_stats(const float *x, float *sums, float *stats, long M, int C, float eps, hipStream_t st, bool zeroed, float *det_part, size_t det_floats)
{
    // det_part (optional scratch, det_floats floats): the deterministic form — per-workgroup partial sums added in fixed order, no atomics (round 6)
    if (det_part && quad_shape_ok(C, M) && quad_aligned(x) && quad_aligned(det_part)) {
        unsigned nwg = quad_grid(M, C, 8, 256);
        if ((size_t)nwg * 2 * C <= det_floats) {
            DLKA_QUAD_DISPATCH(C, cl_bn_stats_det_q_kernel, nwg, x, det_part, M)
            DLKA_CHECK_LAUNCH();
            DLKA_LAUNCH(cl_bn_finish_stats_det_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, x, (const float *)det_part, (int)nwg, stats, M, C, eps);
            DLKA_CHECK_LAUNCH();
            return DLKA_OK;
        }
    }
    if (!zeroed) DLKA_TRY_LAUNCH(launch_zero(sums, (size_t)2 * C * 4, st));
    const int rpb = NT / (C < NT ? C : NT);
    if (quad_shape_ok(C, M) && quad_aligned(x)) {
        DLKA_QUAD_DISPATCH(C, cl_bn_stats_q_kernel, quad_grid(M, C, 8, 1024), x, sums, M)
    } else
        DLKA_LAUNCH(cl_bn_stats_kernel, dim3(grid_for(M, rpb * 16, 1024)), dim3(NT), 2 * C * sizeof(float), st, x, sums, M, C);
    DLKA_CHECK_LAUNCH();
    DLKA_LAUNCH(cl_bn_finish_stats_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, x, (const float *)sums, stats, M, C, eps);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_bn_apply(const float *x, const float *res, const float *w, const float *b, const float *stats, const float *mask, float *y, long M, long N, int C,
                       float slope, hipStream_t st)
{
    if (quad_shape_ok(C, M) && quad_aligned(x) && quad_aligned(res) && quad_aligned(y) && quad_aligned(stats) && quad_aligned(w) && quad_aligned(b) && quad_aligned(mask)) {
        DLKA_QUAD_DISPATCH(C, cl_bn_apply_q_kernel, quad_grid(M, C, 2, 4096), x, res, w, b, stats, mask, y, M, N, slope)
        DLKA_CHECK_LAUNCH();
        return DLKA_OK;
    }
    DLKA_LAUNCH(cl_bn_apply_kernel, dim3(grid_for(M * C, NT)), dim3(NT), 0, st, x, res, w, b, stats, mask, y, M, N, C, slope);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_cl_bn_bwd(const float *g, const float *gmask, const float *x, const float *y, const float *w, const float *stats, float *sums, float *gx, float *gres,
                     const float *gres_add, float *gw, float *gb, long M, long N, int C, float slope, int training, hipStream_t st, bool zeroed)
{
    if (!zeroed) DLKA_TRY_LAUNCH(launch_zero(sums, (size_t)2 * C * 4, st));
    const int rpb = NT / (C < NT ? C : NT);
    if (quad_shape_ok(C, M) && quad_aligned(g) && quad_aligned(x) && quad_aligned(y) && quad_aligned(gx) && quad_aligned(gres) && quad_aligned(gres_add) &&
        quad_aligned(stats) && quad_aligned(sums) && quad_aligned(w) && quad_aligned(gmask)) {
        // At most FOLD_WGS workgroups: every workgroup ends in 2 C fp32 atomics on the same 2 C addresses, which serialise — us at (2, 32^3, 32) for 2048 / 1024 / 512 / 256 / 128
        // workgroups: 58.0 / 33.2 / 22.0 / 15.0 / 14.7 (round 6; 512 until then)
        DLKA_QUAD_DISPATCH(C, cl_bn_bwd_reduce_q_kernel, quad_grid(M, C, 4, FOLD_WGS), g, gmask, x, y, stats, sums, gres, gres_add, M, N, slope)
        DLKA_CHECK_LAUNCH();
        DLKA_QUAD_DISPATCH(C, cl_bn_bwd_apply_q_kernel, quad_grid(M, C, 2, 4096), g, gmask, x, y, w, stats, (const float *)sums, gx, gw, gb, M, N, slope, training)
        DLKA_CHECK_LAUNCH();
        return DLKA_OK;
    }
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
