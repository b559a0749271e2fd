// Micro-benchmark: throughput of LDS and global atomic adds (f32 / u32 / u64) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T, int MODE>   // MODE 0: LDS, 1: global
__global__ __launch_bounds__(256) void k_atomic(T *g, int iters, int gsize)
{
    __shared__ T lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0;
    __syncthreads();
    const int tid = threadIdx.x;
    unsigned a = tid + blockIdx.x * 977;
    for (int it = 0; it < iters; ++it) {
        a = a * 1664525u + 1013904223u;
        // wave-coalesced pattern: 64 lanes hit 64 consecutive elements at a pseudo-random base (like our row pieces)
        const unsigned base = (a >> 8) & ~63u;
        if (MODE == 0) atomicAdd(&lds[(base + (tid & 63) + (tid >> 6) * 64) & 8191], (T)1);
        else atomicAdd(&g[(base * 16 + (tid & 63) + (tid >> 6) * 4096 + blockIdx.x * 64) % gsize], (T)1);
    }
    __syncthreads();
    if (MODE == 0 && tid == 0) g[blockIdx.x] = lds[blockIdx.x & 8191];
}

template <typename T, int MODE>
void run(const char *name)
{
    const int gsize = 1 << 22;
    T *g;
    hipMalloc(&g, sizeof(T) * gsize);
    hipMemset(g, 0, sizeof(T) * gsize);
    const int blocks = 256 * 8, iters = 2000;
    k_atomic<T, MODE><<<blocks, 256>>>(g, 10, gsize);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k_atomic<T, MODE><<<blocks, 256>>>(g, iters, gsize);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * iters;
    printf("%-22s %8.3f ms  %8.1f G atomics/s  (%.2f lane-atomics/clk/CU @2.4GHz)\n", name, ms, n / ms / 1e6, n / ms / 1e6 / 256 / 2.4);
    hipFree(g);
}

int main()
{
    run<float, 0>("LDS f32");
    run<unsigned, 0>("LDS u32");
    run<unsigned long long, 0>("LDS u64");
    run<float, 1>("global f32");
    run<unsigned, 1>("global u32");
    run<unsigned long long, 1>("global u64");
    return 0;
}
