// Micro-benchmark (test infrastructure): trilinear-gather throughput of a [voxel][32 ch] fp32 volume versus the
// lane -> (row, bytes) assignment of the loads.  8 corner rows (128 B each) per (voxel, tap), 27 taps, 65536 voxels.
//   A: lane = (row i of 32, half h): 4 x 16-B loads per lane and corner (64 contiguous bytes per lane)  [igemm AMODE 1]
//   B: lane = (row r of 8, piece p of 8): ONE 16-B load per lane and corner; 8 lanes cover a whole 128-B row; 4 row groups
//   C: lane = (row i of 32, half h), 4 x 16 B at 32-B stride (the grad_offset kernel's D-layout channel sets)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000); }
__device__ __forceinline__ f32x4 ld4(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); }

template <int MODE>
__global__ __launch_bounds__(256) void k(const float *x, const int *base, float *out, int N, int D)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = (blockIdx.x * 4 + wave) * 32;
    auto rx = make_rsrc(x, 2u * N * 128u);
    const int HW = D * D;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < 27; ++tap) {
        if (MODE == 0 || MODE == 2) {
            const int i = lane & 31, h = lane >> 5;
            const int cb = base[(size_t)tap * 2 * N + m0 + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned off = (unsigned)(cb + ((q >> 2) & 1) * HW + ((q >> 1) & 1) * D + (q & 1)) * 128u + (MODE == 0 ? 64u * h : 16u * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc += ld4(rx, off + (MODE == 0 ? 16u : 32u) * e);
            }
        } else {
            const int r = lane >> 3, p = lane & 7;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cb = base[(size_t)tap * 2 * N + m0 + g * 8 + r];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const unsigned off = (unsigned)(cb + ((q >> 2) & 1) * HW + ((q >> 1) & 1) * D + (q & 1)) * 128u + 16u * p;
                    acc += ld4(rx, off);
                }
            }
        }
    }
    out[(size_t)blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main()
{
    const int D = 32, N = D * D * D, M = 2 * N;
    std::vector<int> hb((size_t)27 * M);
    unsigned a = 12345;
    for (int tap = 0; tap < 27; ++tap)
        for (int m = 0; m < M; ++m) {
            const int b = m / N, v = m % N;
            int w = v % D, h = (v / D) % D, d = v / (D * D);
            int z[3] = {d + tap / 9 - 1, h + (tap / 3) % 3 - 1, w + tap % 3 - 1};
            for (int c = 0; c < 3; ++c) { a = a * 1664525u + 1013904223u; z[c] += (int)((a >> 16) % 3) - 1; z[c] = z[c] < 0 ? 0 : (z[c] > D - 2 ? D - 2 : z[c]); }
            hb[(size_t)tap * M + m] = b * N + (z[0] * D + z[1]) * D + z[2];
        }
    float *x, *out; int *base;
    hipMalloc(&x, (size_t)M * 128); hipMemset(x, 0, (size_t)M * 128);
    hipMalloc(&out, (size_t)M / 32 * 64 * 4 * 4);
    hipMalloc(&base, hb.size() * 4); hipMemcpy(base, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    const double bytes = (double)M * 27 * 8 * 128;
    auto run = [&](auto kern, const char *name) {
        kern<<<M / 128, 256>>>(x, base, out, N, D);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) kern<<<M / 128, 256>>>(x, base, out, N, D);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-60s %8.1f us  %7.2f TB/s gathered\n", name, ms * 1e3, bytes / ms / 1e9);
    };
    run(k<0>, "A lane=(row of 32, half): 4 x 16B contiguous per lane");
    run(k<1>, "B lane=(row of 8, piece of 8): 1 x 16B, full 128B rows");
    run(k<2>, "C lane=(row of 32, half): 4 x 16B at 32B stride");
    return 0;
}
