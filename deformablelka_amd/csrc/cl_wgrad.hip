// Channels-last weight gradients on the matrix cores (fp32-input MFMA, exact fp32).
//
//     gW[co][ci][tap] = sum_m G[m][co] * A(m, tap, ci)            m = (b, voxel)
//   AMODE 0  A = in[b][voxel + tap offset][ci]  (zero padded)     1x1x1 projections, offset-predict conv
//   AMODE 1  A = trilinear sample                                  deformable conv; the reference recomputes the whole
//                                                                  im2col buffer for this (deform_conv_cuda.cu:254-261),
//                                                                  here a 32-row x 7-tap sample tile lives in LDS only
//   GMODE 0  G channels-last [M][Cout];  GMODE 1  G planar [B][Cout][N] (the offset tensor keeps the reference layout)
//
// One wave (64-thread workgroup) owns one 32(co) x 32(ci) output tile for TPW taps and walks a chunk of rows, 32 at a
// time: MFMA A operand = G^T (lane i = co), B operand = A(m,tap,ci) (lane j = ci), k = 32 rows per step pair.
// Partial sums per row-chunk go to a workspace and are folded by cl_wgrad_reduce_kernel (no same-address atomics).
#include "deform_sample.h"
#include "cl_args.h"
#include "dlka_kernels.h"

namespace dlka {

template <int AMODE, int GMODE, int TPW>
__global__ __launch_bounds__(64) void cl_wgrad_kernel(WgradArgs p)
{
    constexpr int SROW = 36;  // padded row (floats): conflict-free 16-byte writes, 16-byte aligned
    __shared__ __attribute__((aligned(16))) float S[(AMODE == 1) ? TPW * 32 * SROW : 4];
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int chunk = blockIdx.x;
    const int ot = blockIdx.y / p.CT, ct = blockIdx.y % p.CT;
    const int tap0 = blockIdx.z * TPW;
    const int co = ot * 32 + i;      // A-operand row (as lane i)
    const int ci = ct * 32 + i;      // B-operand column (as lane j = i)

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int m_lo = chunk * p.rows_per_chunk;
    const int m_hi = min(p.M, m_lo + p.rows_per_chunk);
    for (int mbase = m_lo; mbase < m_hi; mbase += 32) {
        // ---- A operand: G[m = mbase + 16h + s][co], s = 0..15 ----
        float ga[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int m = mbase + 16 * h + s;
            float val = 0.f;
            if (m < m_hi && co < p.Cout) {
                if (GMODE == 0) {
                    val = p.g[(long)m * p.Cout + co];
                } else {
                    const int b = m / p.N, v = m - b * p.N;
                    val = p.g[((long)b * p.Cout + co) * p.N + v];
                }
            }
            ga[s] = val;
        }
        if (AMODE == 1) {
            // ---- phase 1: sample tile S[t][row][32 ch] for this wave's TPW taps (lane = (row i, channel half h)) ----
            __syncthreads();  // previous tile consumed
            const int m = mbase + i;
            const bool row_ok = m < m_hi;
            const int b = row_ok ? m / p.N : 0, v = row_ok ? m - b * p.N : 0;
            const int w0 = v % p.W, h0 = (v / p.W) % p.H, d0 = v / (p.W * p.H);
            const float *base = p.in + (long)b * p.N * p.Cin + ct * 32 + 16 * h;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int tap = tap0 + t;
                float a[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) a[e] = 0.f;
                if (tap < p.K && row_ok) {
                    const int tk = tap % p.kw, tj = (tap / p.kw) % p.kh, ti = tap / (p.kw * p.kh);
                    TapSample<3> s;
                    const float *offp = p.off + ((long)b * 3 * p.K + 3 * tap) * p.N + v;
                    setup_tap<3>(s, offp, p.N, d0 + ti * p.dd - p.pd, h0 + tj * p.dh - p.ph, w0 + tk * p.dw - p.pw, p.D, p.H, p.W);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if ((s.ok >> q) & 1u) {
                            const float4 *r4 = reinterpret_cast<const float4 *>(base + (long)s.idx[q] * p.Cin);
                            const float wq = s.w[q];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float4 x4 = r4[e];
                                a[4 * e] = fmaf(wq, x4.x, a[4 * e]); a[4 * e + 1] = fmaf(wq, x4.y, a[4 * e + 1]);
                                a[4 * e + 2] = fmaf(wq, x4.z, a[4 * e + 2]); a[4 * e + 3] = fmaf(wq, x4.w, a[4 * e + 3]);
                            }
                        }
                    }
                }
                float4 *dst = reinterpret_cast<float4 *>(S + (t * 32 + i) * SROW + 16 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[e] = make_float4(a[4 * e], a[4 * e + 1], a[4 * e + 2], a[4 * e + 3]);
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const float *srow = S + (t * 32 + 16 * h) * SROW + i;
#pragma unroll
                for (int s = 0; s < 16; ++s) acc[t] = mfma_32x32x2(ga[s], srow[s * SROW], acc[t]);
            }
        } else {
            // ---- B operand straight from global: in[neighbour(m = mbase + 16h + s)][ci] ----
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int tap = tap0 + t;
                if (tap >= p.K) continue;  // uniform
                const int tk = tap % p.kw, tj = (tap / p.kw) % p.kh, ti = tap / (p.kw * p.kh);
                float bv[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int m = mbase + 16 * h + s;
                    float val = 0.f;
                    if (m < m_hi) {
                        const int b = m / p.N, v = m - b * p.N;
                        const int zw = v % p.W + tk * p.dw - p.pw, zh = (v / p.W) % p.H + tj * p.dh - p.ph,
                                  zd = v / (p.W * p.H) + ti * p.dd - p.pd;
                        if (zd >= 0 && zd < p.D && zh >= 0 && zh < p.H && zw >= 0 && zw < p.W)
                            val = p.in[((long)b * p.N + (long)(zd * p.H + zh) * p.W + zw) * p.Cin + ci];
                    }
                    bv[s] = val;
                }
#pragma unroll
                for (int s = 0; s < 16; ++s) acc[t] = mfma_32x32x2(ga[s], bv[s], acc[t]);
            }
        }
    }
    // ---- partial tile out: D row = co_local, col = ci_local ----
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tap = tap0 + t;
        if (tap >= p.K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            p.part[(((long)chunk * p.K + tap) * p.CoutP + ot * 32 + row) * p.Cin + ci] = acc[t][r];
        }
    }
}

// gW[co][ci][tap] (reference layout, storage type T) = sum_chunk part[chunk][tap][co][ci]
template <typename T>
__global__ void cl_wgrad_reduce_kernel(const float *__restrict__ part, T *__restrict__ gw, int chunks, int K, int CoutP, int Cout, int Cin)
{
    const long n = (long)K * Cout * Cin;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(e % Cin), co = (int)((e / Cin) % Cout), tap = (int)(e / Cin / Cout);
        float a = 0.f;
        for (int c = 0; c < chunks; ++c) a += part[(((long)c * K + tap) * CoutP + co) * Cin + ci];
        stf(gw + ((long)co * Cin + ci) * K + tap, a);
    }
}

int cl_wgrad_pick_chunks(int M)
{
    const int tiles = cdiv(M, 32);
    int chunks = tiles < 128 ? tiles : 128;
    return chunks < 1 ? 1 : chunks;
}

size_t cl_wgrad_part_floats(int M, int K, int Cout, int Cin)
{
    return (size_t)cl_wgrad_pick_chunks(M) * K * round_up(Cout, 32) * Cin;
}

template <typename T>
int launch_cl_wgrad(int amode, int gmode, WgradArgs a, T *gw, hipStream_t st)
{
    const int chunks = cl_wgrad_pick_chunks(a.M);
    const int tiles = cdiv(a.M, 32);
    a.rows_per_chunk = cdiv(tiles, chunks) * 32;
    const int nchunks = cdiv(a.M, a.rows_per_chunk);
    a.CoutP = round_up(a.Cout, 32);
    a.CT = a.Cin / 32;
    const int OT = a.CoutP / 32;
    if (a.K == 1) {
        dim3 grid(nchunks, OT * a.CT, 1), block(64);
        if (amode == 0 && gmode == 0) { auto k = cl_wgrad_kernel<0, 0, 1>; hipLaunchKernelGGL(k, grid, block, 0, st, a); }
        else return DLKA_ERR_UNSUPPORTED;
    } else {
        constexpr int TPW = 7;
        dim3 grid(nchunks, OT * a.CT, cdiv(a.K, TPW)), block(64);
        if (amode == 0 && gmode == 1) { auto k = cl_wgrad_kernel<0, 1, TPW>; hipLaunchKernelGGL(k, grid, block, 0, st, a); }
        else if (amode == 0 && gmode == 0) { auto k = cl_wgrad_kernel<0, 0, TPW>; hipLaunchKernelGGL(k, grid, block, 0, st, a); }
        else if (amode == 1 && gmode == 0) { auto k = cl_wgrad_kernel<1, 0, TPW>; hipLaunchKernelGGL(k, grid, block, 0, st, a); }
        else return DLKA_ERR_UNSUPPORTED;
    }
    DLKA_CHECK_LAUNCH();
    const long n = (long)a.K * a.Cout * a.Cin;
    long blocks = cdivl(n, 256);
    if (blocks > 2048) blocks = 2048;
    auto rk = cl_wgrad_reduce_kernel<T>;
    hipLaunchKernelGGL(rk, dim3((unsigned)blocks), dim3(256), 0, st, (const float *)a.part, gw, nchunks, a.K, a.CoutP, a.Cout, a.Cin);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

template int launch_cl_wgrad<float>(int, int, WgradArgs, float *, hipStream_t);

// ---------------------------------------------------------------------------------------------
// column sums of a channels-last matrix:  gb[n] = sum_m G[m][n]     (bias gradients)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cl_colsum_kernel(const float *__restrict__ g, float *__restrict__ gb, int M, int Cout, int rows_per_block)
{
    __shared__ float red[256];
    const int cols = Cout < 256 ? Cout : 256;       // columns handled per pass
    const int rpp = 256 / cols;                     // rows in flight per pass
    const int c_in = threadIdx.x % cols, r_in = threadIdx.x / cols;
    const int m_lo = blockIdx.x * rows_per_block, m_hi = min(M, m_lo + rows_per_block);
    for (int cb = 0; cb < Cout; cb += cols) {
        const int c = cb + c_in;
        float a = 0.f;
        if (c < Cout && r_in < rpp)
            for (int m = m_lo + r_in; m < m_hi; m += rpp) a += g[(long)m * Cout + c];
        red[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < cols && c < Cout) {
            float t = 0.f;
            for (int r = 0; r < rpp; ++r) t += red[r * cols + threadIdx.x];
            atomicAdd(gb + c, t);
        }
        __syncthreads();
    }
}

int launch_cl_colsum(const float *g, float *gb32, int M, int Cout, hipStream_t st)
{
    if (hipMemsetAsync(gb32, 0, (size_t)Cout * 4, st) != hipSuccess) return DLKA_ERR_LAUNCH;
    int blocks = cdiv(M, 256);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    const int rpb = cdiv(M, blocks);
    hipLaunchKernelGGL(cl_colsum_kernel, dim3(cdiv(M, rpb)), dim3(256), 0, st, g, gb32, M, Cout, rpb);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

}  // namespace dlka
