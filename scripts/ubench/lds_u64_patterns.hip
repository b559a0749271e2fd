// Micro-benchmark (test infrastructure): what the grad_input scatter's LDS traffic costs — 64-bit integer atomics (two fixed-point
// channels per cell), fp64 atomics, plain 64-bit stores and pairs of 32-bit atomics, under the address patterns of the kernel:
// lane = (voxel j of 32 consecutive in w, tap parity h), cell = window base + j + h + jitter, window strides as cl_deform_gx_fx2_kernel<34,10,10>.
// 512 threads per workgroup and ~78 KB of LDS: two workgroups per CU, like the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int SW = 34, SH = 10, SD = 10, PS = SW * SH * SD;   // 3400 cells per channel-pair plane

__device__ __forceinline__ unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 10; }
// small symmetric jitter: sum of two uniform {-1,0,1} draws -> {-2..2}, std ~1.15
__device__ __forceinline__ int jit2(unsigned &s) { const unsigned r = lcg(s); return (int)(r % 3) + (int)((r >> 8) % 3) - 2; }

// OP: 0 = ds_add_u64, 1 = ds_add_f64, 2 = plain 64-bit store, 3 = two ds_add_u32
// PAT: 0 = consecutive cells, no jitter
//      1 = jitter in w only (-2..2)
//      2 = jitter in d, h, w (-2..2 each)  <- the bench's offsets (std ~1 voxel, spatially white)
//      3 = as 2, but only the lanes of one tap parity per instruction (two instructions with half the lanes each)
//      4 = as 2, lanes ordered so that the two half-waves are far apart (h -> +5 planes)
//      5 = every lane random in the plane
template <int OP, int PAT>
__global__ __launch_bounds__(512) void k(float *g, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *win = reinterpret_cast<unsigned long long *>(smem);   // [2][PS]
    for (int i = threadIdx.x; i < 2 * PS; i += 512) win[i] = 0ull;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    unsigned sw = (tid >> 6) * 7919u + blockIdx.x * 977u + 1u;   // wave-uniform stream
    unsigned sl = tid * 2654435761u + blockIdx.x * 977u + 3u;    // per-lane stream
    for (int it = 0; it < iters; ++it) {
        // a row of the brick: (xd, xh) of the tile, wave-uniform; the sample of (voxel j, tap parity h) starts at w = 1 + j + h - 1
        const unsigned r = lcg(sw);
        const int xd0 = 3 + (int)(r % 3), xh0 = 3 + (int)((r >> 4) % 3);
        int xd = xd0, xh = xh0, xw = j + h;
        if (PAT == 1) xw += jit2(sl);
        if (PAT == 2 || PAT == 3 || PAT == 4) { xd += jit2(sl); xh += jit2(sl); xw += jit2(sl); }
        if (PAT == 4) xd = (xd0 - 3) + jit2(sl) + 2 + 5 * h - (h ? 1 : 0);
        xd = xd < 0 ? 0 : (xd > SD - 2 ? SD - 2 : xd);
        xh = xh < 0 ? 0 : (xh > SH - 2 ? SH - 2 : xh);
        xw = xw < 0 ? 0 : (xw > SW - 2 ? SW - 2 : xw);
        int cell = (xd * SH + xh) * SW + xw;
        if (PAT == 5) cell = (int)(lcg(sl) % (PS - SH * SW - SW - 2));
        const bool on = PAT == 3 ? true : true;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int off = ((q >> 2) & 1) * SH * SW + ((q >> 1) & 1) * SW + (q & 1);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                unsigned long long *p = win + pr * PS + cell + off;
                const unsigned long long v = ((unsigned long long)(unsigned)(it + q) << 32) | (unsigned)(lane + pr);
                if (PAT == 3) {
                    if (OP == 0) { if (h == 0) atomicAdd(p, v); if (h == 1) atomicAdd(p, v); }
                    if (OP == 1) { if (h == 0) atomicAdd(reinterpret_cast<double *>(p), 1.0); if (h == 1) atomicAdd(reinterpret_cast<double *>(p), 1.0); }
                    if (OP == 2) { if (h == 0) *reinterpret_cast<volatile unsigned long long *>(p) = v; if (h == 1) *reinterpret_cast<volatile unsigned long long *>(p) = v; }
                    if (OP == 3) { unsigned *p32 = reinterpret_cast<unsigned *>(p); if (h == 0) { atomicAdd(p32, (unsigned)v); atomicAdd(p32 + 1, (unsigned)(v >> 32)); } if (h == 1) { atomicAdd(p32, (unsigned)v); atomicAdd(p32 + 1, (unsigned)(v >> 32)); } }
                } else if (on) {
                    if (OP == 0) atomicAdd(p, v);
                    if (OP == 1) atomicAdd(reinterpret_cast<double *>(p), 1.0);
                    if (OP == 2) *reinterpret_cast<volatile unsigned long long *>(p) = v;
                    if (OP == 3) { unsigned *p32 = reinterpret_cast<unsigned *>(p); atomicAdd(p32, (unsigned)v); atomicAdd(p32 + 1, (unsigned)(v >> 32)); }
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) g[blockIdx.x] = (float)win[blockIdx.x % PS];
}

template <int OP, int PAT>
void run(const char *name, float *g)
{
    const int blocks = 1024, iters = 200;   // 16 instructions (8 corners x 2 pairs) per iteration and wave
    const size_t lds = 2 * PS * 8 + 24 * 1024;   // window + what the kernel keeps next to it (weights, queues): two workgroups per CU
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<OP, PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<OP, PAT><<<blocks, 512, lds>>>(g, 4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<OP, PAT><<<blocks, 512, lds>>>(g, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 8 * iters * 16 * (OP == 3 ? 1 : 1);   // 64-bit wave-instructions (a 32-bit pair counts as one)
    const double clk_per = ms * 1e-3 * 2.4e9 * 256 / winstr;                     // CU-clocks per wave-instruction
    static const char *ops[4] = {"ds_add_u64", "ds_add_f64", "ds_write_b64", "2 x ds_add_u32"};
    printf("%-15s %-52s %8.3f ms  %6.1f clk per 64-bit wave-instr per CU\n", ops[OP], name, ms, clk_per);
}

template <int OP>
void run_all(float *g)
{
    run<OP, 0>("consecutive cells", g);
    run<OP, 1>("jitter in w (-2..2)", g);
    run<OP, 2>("jitter in d, h, w (-2..2): the bench's offsets", g);
    run<OP, 3>("  same, one tap parity (32 lanes) per instruction", g);
    run<OP, 4>("  same, half-waves 5 planes apart", g);
    run<OP, 5>("every lane random in the plane", g);
}

int main()
{
    float *g;
    hipMalloc(&g, 1 << 20);
    run_all<0>(g);
    run_all<1>(g);
    run_all<2>(g);
    run_all<3>(g);
    hipFree(g);
    return 0;
}
