// fp32 FMA issue rate of one CU (gfx950): independent v_fma_f32 / v_pk_fma_f32 chains, W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o fma_rate fma_rate.hip && ./fma_rate
// Prints FMA lanes per clock and CU (clock = s_memtime ticks converted with the measured shader clock of the run: wall time of the kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NACC, bool PK>
__global__ __launch_bounds__(1024) void fma_kernel(float *out, int iters, float a, float b)
{
    float acc[NACC];
    f32x2 acc2[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) { acc[k] = (float)(threadIdx.x + k); acc2[k] = f32x2{acc[k], acc[k] + 1.f}; }
    const f32x2 a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int k = 0; k < NACC; ++k) {
                if (PK) acc2[k] = __builtin_elementwise_fma(acc2[k], a2, b2);
                else acc[k] = __builtin_fmaf(acc[k], a, b);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NACC; ++k) s += PK ? acc2[k][0] + acc2[k][1] : acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool PK>
static void run(int waves_per_simd, const char *name)
{
    const int iters = 4096, threads = 64 * 4 * waves_per_simd;   // one workgroup per CU, 4 SIMDs
    int ncu = 256;
    float *out;
    hipMalloc(&out, (size_t)ncu * threads * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((fma_kernel<NACC, PK>), dim3(ncu), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double fma_lanes = (double)ncu * threads * iters * 8 * NACC * (PK ? 2 : 1);
    printf("%-22s NACC=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  = %.1f FMA lanes per ns and CU (x clock GHz^-1 = lanes / clk)\n", name, NACC, waves_per_simd, ms,
           2.0 * fma_lanes / (ms * 1e-3) / 1e12, fma_lanes / ncu / (ms * 1e6));
    hipFree(out);
}

// Operand patterns of the depthwise kernels: every source a VGPR.  MODE 0: v_fma_f32 acc = w * s + acc; 1: v_pk_fma_f32 acc2 = w2 * s2 + acc2 (three 64-bit VGPR sources);
// 2: v_pk_fma_f32 acc2 = w2 * s.xx + acc2 (op_sel broadcast of one dword); 3: v_pk_fma_f32 acc2 = W2(SGPR pair) * s.xx + acc2.
template <int MODE>
__global__ __launch_bounds__(1024) void fma_vgpr_kernel(float *out, const float *in, int iters, float a, float b)
{
    constexpr int NACC = 8;
    float acc[NACC], w[NACC], s[NACC];
    f32x2 acc2[NACC], w2[NACC], s2[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        acc[k] = (float)(threadIdx.x + k); acc2[k] = f32x2{acc[k], acc[k] + 1.f};
        w[k] = in[threadIdx.x + 64 * k]; s[k] = in[threadIdx.x + 64 * (k + 8)];
        w2[k] = f32x2{w[k], in[threadIdx.x + 64 * (k + 16)]}; s2[k] = f32x2{s[k], in[threadIdx.x + 64 * (k + 24)]};
    }
    const f32x2 W2 = {a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int k = 0; k < NACC; ++k) {
                if (MODE == 0) acc[k] = __builtin_fmaf(w[(k + r) & 7], s[(k + 3 * r) & 7], acc[k]);
                if (MODE == 1) acc2[k] = __builtin_elementwise_fma(w2[(k + r) & 7], s2[(k + 3 * r) & 7], acc2[k]);
                if (MODE == 2) acc2[k] = __builtin_elementwise_fma(w2[(k + r) & 7], f32x2{s[(k + 3 * r) & 7], s[(k + 3 * r) & 7]}, acc2[k]);
                if (MODE == 3) acc2[k] = __builtin_elementwise_fma(W2, f32x2{s[(k + 3 * r) & 7], s[(k + 3 * r) & 7]}, acc2[k]);
            }
    }
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < NACC; ++k) t += MODE ? acc2[k][0] + acc2[k][1] : acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int MODE>
static void run_vgpr(int waves_per_simd, const char *name)
{
    const int iters = 4096, threads = 64 * 4 * waves_per_simd, ncu = 256;
    float *out, *in;
    hipMalloc(&out, (size_t)ncu * threads * sizeof(float));
    hipMalloc(&in, (size_t)(1024 + 64 * 32) * sizeof(float));
    hipMemset(in, 0, (size_t)(1024 + 64 * 32) * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((fma_vgpr_kernel<MODE>), dim3(ncu), dim3(threads), 0, 0, out, in, iters, 1.0001f, 0.5f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double fma_lanes = (double)ncu * threads * iters * 8 * 8 * (MODE ? 2 : 1);
    printf("%-44s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  = %.1f FMA lanes per ns and CU\n", name, waves_per_simd, ms, 2.0 * fma_lanes / (ms * 1e-3) / 1e12, fma_lanes / ncu / (ms * 1e6));
    hipFree(out); hipFree(in);
}

int main()
{
    for (int w = 1; w <= 4; w *= 2) {
        run_vgpr<0>(w, "v_fma_f32, 3 VGPR sources");
        run_vgpr<1>(w, "v_pk_fma_f32, 3 VGPR pairs");
        run_vgpr<2>(w, "v_pk_fma_f32, VGPR pair * VGPR.xx + pair");
        run_vgpr<3>(w, "v_pk_fma_f32, SGPR pair * VGPR.xx + pair");
    }
    for (int w = 1; w <= 4; w *= 2) {
        run<1, false>(w, "v_fma_f32 dependent");
        run<8, false>(w, "v_fma_f32 8 chains");
        run<8, true>(w, "v_pk_fma_f32 8 chains");
    }
    return 0;
}
