// Micro-benchmark (test infrastructure): ds_add_f64 throughput versus the address pattern of the 64 lanes.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *g, int iters)
{
    __shared__ __attribute__((aligned(16))) double win[8192];   // 64 KB
    for (int i = threadIdx.x; i < 8192; i += 256) win[i] = 0.0;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned aw = (tid >> 6) * 7919 + blockIdx.x * 977 + 1;      // wave-uniform
    unsigned ah = (tid >> 5) * 104729 + blockIdx.x * 977 + 7;    // half-wave-uniform
    unsigned al = tid * 2654435761u + blockIdx.x * 977 + 3;      // per lane
    const int q = (lane >> 2) & 7, c = lane & 3;
    for (int it = 0; it < iters; ++it) {
        aw = aw * 1664525u + 1013904223u;
        ah = ah * 1664525u + 1013904223u;
        al = al * 1664525u + 1013904223u;
        int idx;
        if (MODE == 0) idx = ((aw >> 10) * 64 + lane) & 8191;                                  // 64 consecutive doubles
        if (MODE == 1) idx = (al >> 10) & 8191;                                               // every lane random
        if (MODE == 2) idx = (((aw >> 10) & 4095) + lane + ((al >> 12) & 1)) & 8191;          // consecutive + 0/1 jitter
        if (MODE == 3) idx = (((aw >> 10) & 4095) + lane + ((al >> 12) % 25)) & 8191;         // consecutive + 0..24 jitter
        if (MODE == 4 || MODE == 5 || MODE == 6) {   // half-wave = one sample: 8 corners x 4 channels, cell-major [cell][4]
            const int WW = MODE == 4 ? 10 : (MODE == 5 ? 14 : 16), WHWW = MODE == 4 ? 140 : (MODE == 5 ? 196 : 256);
            const int cell = ((ah >> 10) % 1400) + ((q >> 2) & 1) * WHWW + ((q >> 1) & 1) * WW + (q & 1);
            idx = (cell * 4 + c) & 8191;
        }
        if (MODE == 7) {   // 16 lanes = one sample (8 corners x 2 channel pairs?) -> here: quarter-wave sample, 8 corners x 2 ch
            const int cell = (((al >> 10) - ((al >> 10) % 1)) % 1400);
            idx = cell & 8191;
        }
        atomicAdd(&win[idx], 1.0);
    }
    __syncthreads();
    if (tid == 0) g[blockIdx.x] = (float)win[blockIdx.x & 8191];
}

template <int MODE>
void run(const char *name, float *g)
{
    const int blocks = 2048, iters = 2000;
    k<MODE><<<blocks, 256>>>(g, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(g, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * iters;
    printf("ds_add_f64 %-44s %8.3f ms  %6.2f lanes/clk/CU  (%5.1f clk per wave-instr)\n", name, ms, n / ms / 1e6 / 256 / 2.4, 64.0 / (n / ms / 1e6 / 256 / 2.4));
}

int main()
{
    float *g;
    hipMalloc(&g, 1 << 20);
    run<0>("64 consecutive", g);
    run<1>("every lane random in 64KB", g);
    run<2>("consecutive + 0/1 jitter", g);
    run<3>("consecutive + 0..24 jitter", g);
    run<4>("half-wave sample 8 corners x 4ch, WW=10", g);
    run<5>("half-wave sample 8 corners x 4ch, WW=14", g);
    run<6>("half-wave sample 8 corners x 4ch, WW=16", g);
    hipFree(g);
    return 0;
}
