// Planar (NCDHW) plumbing ops of the full D_LKA_Former around the D-LKA blocks: BatchNorm3d in training mode and 1x1x1
// convolutions on few channels at full resolution (encoder1 / decoder2 / the output heads: 16 channels, 2 x 64 x 128 x 128 voxels,
// 3D/d_lka_former/network_architecture/dynunet_block.py:12-80, synapse/d_lka_former_synapse.py:89-133,148-150).
//
// Why they exist: profiled on the MI355X (round 3, rocprofv3 of one trainer iteration), torch's batch-norm kernels launch ONE workgroup per
// channel — 16 workgroups on 256 CUs: 0.20 + 0.36 + 0.77 ms per layer and iteration for tensors that stream in 0.05 ms — and the 16 -> 14 output
// head as a GEMM lands on a 16 x 16 hipBLASLt macro-tile: 3.8 ms (its gradients 1.2 + 0.5 ms) for 250 MB of traffic.  All of these are
// HBM-bound streams; the kernels below spread every (batch, channel) plane over many workgroups, read 16 bytes per lane, and meet in fp32 atomics
// on per-channel accumulators.
#include "dlka_common.h"
#include "dlka_kernels.h"

namespace dlka {

namespace {
constexpr int PL_THREADS = 256;
constexpr int PL_CHUNK = 4096;    // elements of a plane per workgroup of the streaming kernels (16 per thread)
constexpr int PL_RCHUNK = 32768;  // ... of the reducing kernels (128 per thread: the reduction tail — shuffles, one barrier, two atomics — was most
                                  //     of a 4096-element workgroup's time: 143 us for a 134 MB tensor)

// sum of a value over the workgroup (4 waves); result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float *red)
{
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}
}  // namespace

// ---- BatchNorm3d, training mode ----------------------------------------------------------------------------------------------------
// acc[c] = {sum (x - p_c), sum (x - p_c)^2} with the pivot p_c = x[0][c][0] (|mean| >> std must not cancel the variance away)
__global__ __launch_bounds__(PL_THREADS) void pl_bn_stats_kernel(const float *__restrict__ x, float *__restrict__ acc, int C, long N)
{
    __shared__ float red[PL_THREADS];
    const int plane = blockIdx.y, c = plane % C;
    const float p = x[(long)c * N];
    const float *xp = x + (long)plane * N;
    const long e0 = (long)blockIdx.x * PL_RCHUNK;
    float s1 = 0.f, s2 = 0.f;
    if ((N & 3) == 0) {
        for (long e = e0 + 4 * threadIdx.x; e < e0 + PL_RCHUNK && e < N; e += 4 * PL_THREADS) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xp + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float d = v[k] - p; s1 += d; s2 = fmaf(d, d, s2); }
        }
    } else {
        for (long e = e0 + threadIdx.x; e < e0 + PL_RCHUNK && e < N; e += PL_THREADS) { const float d = xp[e] - p; s1 += d; s2 = fmaf(d, d, s2); }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) { atomicAdd(acc + 2 * c, s1); atomicAdd(acc + 2 * c + 1, s2); }
}

// stats[c] = mean, stats[C + c] = rstd, stats[2C + c] = unbiased variance (for the running estimate), stats[3C + c] = mean - pivot: the kernels
// centre as (x - pivot) - (mean - pivot) — with |mean| >> std the fp32 mean itself is only good to a fraction of the std
__global__ void pl_bn_finish_kernel(const float *__restrict__ x, const float *__restrict__ acc, float *__restrict__ stats, int C, long N, long count, float eps)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float p = x[(long)c * N];
    const float m1 = acc[2 * c] / (float)count, m2 = acc[2 * c + 1] / (float)count;
    float var = m2 - m1 * m1;
    var = var > 0.f ? var : 0.f;
    stats[c] = p + m1;
    stats[C + c] = 1.f / sqrtf(var + eps);
    stats[2 * C + c] = count > 1 ? var * (float)count / (float)(count - 1) : var;
    stats[3 * C + c] = m1;
}

// y = (x - mean) rstd w + b
__global__ __launch_bounds__(PL_THREADS) void pl_bn_apply_kernel(const float *__restrict__ x, const float *__restrict__ stats, const float *__restrict__ w,
                                                                 const float *__restrict__ b, float *__restrict__ y, int C, long N)
{
    const int plane = blockIdx.y, c = plane % C;
    const float piv = x[(long)c * N], dm = stats[3 * C + c], sc = stats[C + c] * (w ? w[c] : 1.f), sh = b ? b[c] : 0.f;
    const float *xp = x + (long)plane * N;
    float *yp = y + (long)plane * N;
    const long e0 = (long)blockIdx.x * PL_CHUNK;
    if ((N & 3) == 0) {
        for (long e = e0 + 4 * threadIdx.x; e < e0 + PL_CHUNK && e < N; e += 4 * PL_THREADS) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(xp + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaf((v[k] - piv) - dm, sc, sh);
            *reinterpret_cast<f32x4 *>(yp + e) = v;
        }
    } else {
        for (long e = e0 + threadIdx.x; e < e0 + PL_CHUNK && e < N; e += PL_THREADS) yp[e] = fmaf((xp[e] - piv) - dm, sc, sh);
    }
}

// acc[c] = {sum g, sum g xhat},  xhat = (x - mean) rstd
__global__ __launch_bounds__(PL_THREADS) void pl_bn_bwd_reduce_kernel(const float *__restrict__ g, const float *__restrict__ x, const float *__restrict__ stats,
                                                                      float *__restrict__ acc, int C, long N)
{
    __shared__ float red[PL_THREADS];
    const int plane = blockIdx.y, c = plane % C;
    const float piv = x[(long)c * N], dm = stats[3 * C + c], rstd = stats[C + c];
    const float *xp = x + (long)plane * N, *gp = g + (long)plane * N;
    const long e0 = (long)blockIdx.x * PL_RCHUNK;
    float s1 = 0.f, s2 = 0.f;
    if ((N & 3) == 0) {
        for (long e = e0 + 4 * threadIdx.x; e < e0 + PL_RCHUNK && e < N; e += 4 * PL_THREADS) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xp + e), q = *reinterpret_cast<const f32x4 *>(gp + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) { s1 += q[k]; s2 = fmaf(q[k], ((v[k] - piv) - dm) * rstd, s2); }
        }
    } else {
        for (long e = e0 + threadIdx.x; e < e0 + PL_RCHUNK && e < N; e += PL_THREADS) { s1 += gp[e]; s2 = fmaf(gp[e], ((xp[e] - piv) - dm) * rstd, s2); }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) { atomicAdd(acc + 2 * c, s1); atomicAdd(acc + 2 * c + 1, s2); }
}

// gx = w rstd (g - mean(g) - xhat mean(g xhat));  the first workgroup of a channel's first plane also writes gw = sum g xhat, gb = sum g
__global__ __launch_bounds__(PL_THREADS) void pl_bn_bwd_apply_kernel(const float *__restrict__ g, const float *__restrict__ x, const float *__restrict__ stats,
                                                                     const float *__restrict__ w, const float *__restrict__ acc, float *__restrict__ gx,
                                                                     float *__restrict__ gw, float *__restrict__ gb, int C, long N, long count)
{
    const int plane = blockIdx.y, c = plane % C;
    const float piv = x[(long)c * N], dm = stats[3 * C + c], rstd = stats[C + c];
    const float sg = acc[2 * c], sgx = acc[2 * c + 1];
    const float k0 = rstd * (w ? w[c] : 1.f), mg = sg / (float)count, mgx = sgx / (float)count;
    if (blockIdx.x == 0 && plane == c && threadIdx.x == 0) {
        if (gw) gw[c] = sgx;
        if (gb) gb[c] = sg;
    }
    const float *xp = x + (long)plane * N, *gp = g + (long)plane * N;
    float *op = gx + (long)plane * N;
    const long e0 = (long)blockIdx.x * PL_CHUNK;
    if ((N & 3) == 0) {
        for (long e = e0 + 4 * threadIdx.x; e < e0 + PL_CHUNK && e < N; e += 4 * PL_THREADS) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xp + e), q = *reinterpret_cast<const f32x4 *>(gp + e);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = k0 * (q[k] - mg - ((v[k] - piv) - dm) * rstd * mgx);
            *reinterpret_cast<f32x4 *>(op + e) = o;
        }
    } else {
        for (long e = e0 + threadIdx.x; e < e0 + PL_CHUNK && e < N; e += PL_THREADS) op[e] = k0 * (gp[e] - mg - ((xp[e] - piv) - dm) * rstd * mgx);
    }
}

int launch_pl_bn_forward(const float *x, const float *w, const float *b, float *stats, float *y, float *scratch, int B, int C, long N, float eps, hipStream_t st)
{
    if (B <= 0 || C <= 0 || N <= 0 || (long)B * C > 65535) return DLKA_ERR_SHAPE;
    if (launch_zero(scratch, (size_t)2 * C * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
    const dim3 grid((unsigned)cdivl(N, PL_CHUNK), B * C), rgrid((unsigned)cdivl(N, PL_RCHUNK), B * C);
    DLKA_LAUNCH(pl_bn_stats_kernel, rgrid, dim3(PL_THREADS), 0, st, x, scratch, C, N);
    DLKA_CHECK_LAUNCH();
    DLKA_LAUNCH(pl_bn_finish_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, x, (const float *)scratch, stats, C, N, (long)B * N, eps);
    DLKA_CHECK_LAUNCH();
    DLKA_LAUNCH(pl_bn_apply_kernel, grid, dim3(PL_THREADS), 0, st, x, (const float *)stats, w, b, y, C, N);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_pl_bn_backward(const float *g, const float *x, const float *w, const float *stats, float *gx, float *gw, float *gb, float *scratch, int B, int C,
                          long N, hipStream_t st)
{
    if (B <= 0 || C <= 0 || N <= 0 || (long)B * C > 65535) return DLKA_ERR_SHAPE;
    if (launch_zero(scratch, (size_t)2 * C * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
    const dim3 grid((unsigned)cdivl(N, PL_CHUNK), B * C), rgrid((unsigned)cdivl(N, PL_RCHUNK), B * C);
    DLKA_LAUNCH(pl_bn_bwd_reduce_kernel, rgrid, dim3(PL_THREADS), 0, st, g, x, stats, scratch, C, N);
    DLKA_CHECK_LAUNCH();
    DLKA_LAUNCH(pl_bn_bwd_apply_kernel, grid, dim3(PL_THREADS), 0, st, g, x, stats, w, (const float *)scratch, gx, gw, gb, C, N, (long)B * N);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// ---- 1x1x1 convolution on planar tensors, few channels --------------------------------------------------------------------------------
// y[b][co][v] = sum_ci W[co][ci] x[b][ci][v] + bias[co].  One thread owns 4 consecutive voxels: CI 16-byte loads (each coalesced across the
// wave), CO 16-byte stores; the weights are wave-uniform (scalar loads).  CI, CO <= 32.
template <int CI>
__global__ __launch_bounds__(PL_THREADS) void pl_pw_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                               float *__restrict__ y, int CO, long N)
{
    const int b = blockIdx.y;
    const long v = ((long)blockIdx.x * PL_THREADS + threadIdx.x) * 4;
    if (v >= N) return;
    const float *xp = x + (long)b * CI * N + v;
    f32x4 xv[CI];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) xv[ci] = *reinterpret_cast<const f32x4 *>(xp + (long)ci * N);
    float *yp = y + (long)b * CO * N + v;
    for (int co = 0; co < CO; ++co) {
        const float bv = bias ? bias[co] : 0.f;
        f32x4 a = {bv, bv, bv, bv};
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
            const float wv = w[co * CI + ci];
            a[0] = fmaf(wv, xv[ci][0], a[0]); a[1] = fmaf(wv, xv[ci][1], a[1]); a[2] = fmaf(wv, xv[ci][2], a[2]); a[3] = fmaf(wv, xv[ci][3], a[3]);
        }
        *reinterpret_cast<f32x4 *>(yp + (long)co * N) = a;
    }
}

// gx[b][ci][v] = sum_co W[co][ci] g[b][co][v]  — the same kernel shape with the roles of the channel axes exchanged (wt = W^T is not formed:
// the weight index is co * CI + ci either way)
template <int CO>
__global__ __launch_bounds__(PL_THREADS) void pl_pw_bwd_data_kernel(const float *__restrict__ g, const float *__restrict__ w, float *__restrict__ gx, int CI, long N)
{
    const int b = blockIdx.y;
    const long v = ((long)blockIdx.x * PL_THREADS + threadIdx.x) * 4;
    if (v >= N) return;
    const float *gp = g + (long)b * CO * N + v;
    f32x4 gv[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) gv[co] = *reinterpret_cast<const f32x4 *>(gp + (long)co * N);
    float *op = gx + (long)b * CI * N + v;
    for (int ci = 0; ci < CI; ++ci) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const float wv = w[co * CI + ci];
            a[0] = fmaf(wv, gv[co][0], a[0]); a[1] = fmaf(wv, gv[co][1], a[1]); a[2] = fmaf(wv, gv[co][2], a[2]); a[3] = fmaf(wv, gv[co][3], a[3]);
        }
        *reinterpret_cast<f32x4 *>(op + (long)ci * N) = a;
    }
}

// gW[co][ci] += sum_v g[b][co][v] x[b][ci][v],  gb[co] += sum_v g[b][co][v]   (outputs zeroed by the launcher; CO <= 16, CI <= 16 * NCT).
// The voxel axis is the contraction: v_mfma_f32_16x16x4_f32 with A[i = co][k = voxel], B[k = voxel][j = ci].  Lane (i, kg = lane >> 4) loads 16
// bytes — voxels 4 kg .. 4 kg + 3 of its channel row, 64 contiguous bytes per row and instruction — and step s of the 4 MFMAs that follow contracts
// element s of every lane (k = kg <-> voxel 4 kg + s: the same assignment on both operands).  A wave streams its run of voxels and ends with one
// atomic per output it holds.
template <int NCT>
__global__ __launch_bounds__(PL_THREADS) void pl_pw_bwd_weight_kernel(const float *__restrict__ g, const float *__restrict__ x, float *__restrict__ gw,
                                                                      float *__restrict__ gb, int CO, int CI, long N, int chunk)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kg = lane >> 4;
    const int per_wave = chunk / (PL_THREADS / 64);
    const long v0 = (long)blockIdx.x * chunk + (long)wave * per_wave;
    const long v1 = v0 + per_wave < N ? v0 + per_wave : N;
    const float *gp = g + ((long)b * CO + (i < CO ? i : 0)) * N;
    f32x4 acc[NCT];
#pragma unroll
    for (int t = 0; t < NCT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float sb = 0.f;
    for (long vb = v0; vb < v1; vb += 16) {   // wave-uniform trip count: the MFMAs below are wave-wide whatever the exec mask
        const long v = vb + 4 * kg;
        const bool in = v < v1;                  // (N % 4 == 0 and chunk % 16 == 0: a lane's four voxels are inside the run or all outside)
        f32x4 q = *reinterpret_cast<const f32x4 *>(gp + (in ? v : 0));
        if (i >= CO || !in) q = f32x4{0.f, 0.f, 0.f, 0.f};
        sb += (q[0] + q[1]) + (q[2] + q[3]);
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
            const int ci = 16 * t + i;
            f32x4 xv = *reinterpret_cast<const f32x4 *>(x + ((long)b * CI + (ci < CI ? ci : 0)) * N + (in ? v : 0));
            if (ci >= CI || !in) xv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[t] = mfma_16x16x4(q[s], xv[s], acc[t]);
        }
    }
    // the workgroup's waves fold their tiles through LDS: one atomic per output and WORKGROUP (round 6: the run per workgroup went from 16 384 to 4096 voxels — 128 workgroups
    // left half the chip idle at 2 x 64 x 128 x 128 — and one atomic per wave would then be 460 k atomics on 224 addresses)
    constexpr int NW = PL_THREADS / 64;
    __shared__ __attribute__((aligned(16))) float red[NW - 1][NCT * 64 * 4 + 64];
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NCT; ++t) *reinterpret_cast<f32x4 *>(&red[wave - 1][(t * 64 + lane) * 4]) = acc[t];
        red[wave - 1][NCT * 256 + lane] = sb;
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) {
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
            const f32x4 o = *reinterpret_cast<const f32x4 *>(&red[w][(t * 64 + lane) * 4]);
            acc[t][0] += o[0]; acc[t][1] += o[1]; acc[t][2] += o[2]; acc[t][3] += o[3];
        }
        sb += red[w][NCT * 256 + lane];
    }
    // D layout: column j = lane & 15 (ci within the tile), rows 4 kg + r (co)
#pragma unroll
    for (int t = 0; t < NCT; ++t) {
        const int ci = 16 * t + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 4 * kg + r;
            if (co < CO && ci < CI) atomicAdd(gw + co * CI + ci, acc[t][r]);
        }
    }
    if (gb) {
        sb += __shfl_xor(sb, 16);
        sb += __shfl_xor(sb, 32);
        if (kg == 0 && i < CO) atomicAdd(gb + i, sb);
    }
}

#define DLKA_PL_CI(M) \
    switch (CIv) { case 1: M(1) break; case 2: M(2) break; case 4: M(4) break; case 8: M(8) break; case 14: M(14) break; case 16: M(16) break; case 32: M(32) break; default: return DLKA_ERR_UNSUPPORTED; }

int launch_pl_pw_forward(const float *x, const float *w, const float *bias, float *y, int B, int CI, int CO, long N, hipStream_t st)
{
    if (B <= 0 || N <= 0 || (N & 3) || CO <= 0 || CO > 64 || B > 65535) return DLKA_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)cdivl(N / 4, PL_THREADS), B);
    const int CIv = CI;
#define M(K) { auto k = pl_pw_fwd_kernel<K>; DLKA_LAUNCH(k, grid, dim3(PL_THREADS), 0, st, x, w, bias, y, CO, N); }
    DLKA_PL_CI(M)
#undef M
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

int launch_pl_pw_backward(const float *x, const float *w, const float *g, float *gx, float *gw, float *gb, int B, int CI, int CO, long N, hipStream_t st)
{
    if (B <= 0 || N <= 0 || (N & 3) || CO <= 0 || CO > 64 || CI <= 0 || CI > 64 || B > 65535) return DLKA_ERR_UNSUPPORTED;
    if (gx) {
        const dim3 grid((unsigned)cdivl(N / 4, PL_THREADS), B);
        const int CIv = CO;   // the kernel is instantiated on the channel count it holds in registers: grad_out's
#define M(K) { auto k = pl_pw_bwd_data_kernel<K>; DLKA_LAUNCH(k, grid, dim3(PL_THREADS), 0, st, g, w, gx, CI, N); }
        DLKA_PL_CI(M)
#undef M
        DLKA_CHECK_LAUNCH();
    }
    if (gw) {
        if (CO > 16 || CI > 64) return DLKA_ERR_UNSUPPORTED;
        if (launch_zero(gw, (size_t)CO * CI * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
        if (gb && launch_zero(gb, (size_t)CO * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
        // voxels per workgroup (a multiple of 16 per wave): the longest run that still gives the chip a workgroup per CU (16 384 left 128 workgroups at
        // 2 x 64 x 128 x 128 and 16 at the 32 -> 14 head's 2 x 32 x 64 x 64: 144 / 150 us; now 78 / see notes)
        int chunk = 16384;
        while (chunk > 1024 && cdivl(N, chunk) * B < 256) chunk /= 4;
        const dim3 grid((unsigned)cdivl(N, chunk), B);
        const int nct = cdiv(CI, 16);
        if (nct == 1) { auto k = pl_pw_bwd_weight_kernel<1>; DLKA_LAUNCH(k, grid, dim3(PL_THREADS), 0, st, g, x, gw, gb, CO, CI, N, chunk); }
        else if (nct == 2) { auto k = pl_pw_bwd_weight_kernel<2>; DLKA_LAUNCH(k, grid, dim3(PL_THREADS), 0, st, g, x, gw, gb, CO, CI, N, chunk); }
        else { auto k = pl_pw_bwd_weight_kernel<4>; DLKA_LAUNCH(k, grid, dim3(PL_THREADS), 0, st, g, x, gw, gb, CO, CI, N, chunk); }
        DLKA_CHECK_LAUNCH();
    }
    return DLKA_OK;
}

}  // namespace dlka
