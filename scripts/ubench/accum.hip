// Micro-benchmark (test infrastructure, not product): how fast can a gfx950 CU accumulate scattered fp32
// contributions?  Decides the design of the deformable-conv grad_input scatter.
//   LDS:    ds_add_f32 / ds_add_rtn_f32 / ds_add_f64 / ds_add_u32 / ds_add_u64 / non-atomic read-add-write
//   global: global_atomic_add_f32 (128-B row per half-wave, as the scatter issues it) / add_f64 / pk_add_bf16 / add_u64
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/ubench/accum.hip -o scripts/ubench/accum_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

enum { L_F32, L_F32_RTN, L_F64, L_U32, L_U64, L_RMW32, L_RMW64, L_RMW128 };

template <int MODE>
__global__ __launch_bounds__(256) void k_lds(float *g, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char raw[64 * 1024];
    for (int i = threadIdx.x; i < 16384; i += 256) ((unsigned *)raw)[i] = 0;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned a = tid / 64 * 7919 + blockIdx.x * 977;   // wave-uniform LCG: all lanes of a wave hit one contiguous run
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
        a = a * 1664525u + 1013904223u;
        const unsigned base = (a >> 10);
        if (MODE == L_F32) atomicAdd(&((float *)raw)[(base * 64 + lane) & 16383], 1.0f);
        if (MODE == L_F32_RTN) keep += atomicAdd(&((float *)raw)[(base * 64 + lane) & 16383], 1.0f);
        if (MODE == L_F64) atomicAdd(&((double *)raw)[(base * 64 + lane) & 8191], 1.0);
        if (MODE == L_U32) atomicAdd(&((unsigned *)raw)[(base * 64 + lane) & 16383], 1u);
        if (MODE == L_U64) atomicAdd(&((unsigned long long *)raw)[(base * 64 + lane) & 8191], 1ull);
        if (MODE == L_RMW32) {   // wave-private 16 KB quarter: read, add, write (no atomics; dependent chain)
            float *p = (float *)raw + wave * 4096 + ((base * 64 + lane) & 4095);
            *p = *p + 1.0f;
        }
        if (MODE == L_RMW64) {
            float2 *p = (float2 *)raw + wave * 2048 + ((base * 64 + lane) & 2047);
            float2 v = *p; v.x += 1.f; v.y += 2.f; *p = v;
        }
        if (MODE == L_RMW128) {
            float4 *p = (float4 *)raw + wave * 1024 + ((base * 64 + lane) & 1023);
            float4 v = *p; v.x += 1.f; v.y += 2.f; v.z += 3.f; v.w += 4.f; *p = v;
        }
    }
    __syncthreads();
    if (tid == 0) g[blockIdx.x] = ((float *)raw)[blockIdx.x & 16383] + keep;
}

enum { G_F32_ROW128, G_F32_ROW256, G_F64, G_PKBF16, G_U64, G_F32_LOCAL };

template <int MODE>
__global__ __launch_bounds__(256) void k_glb(float *g, int iters, unsigned nrows128)
{
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned a = (tid >> 5) * 7919 + blockIdx.x * 977 + 12345;   // half-wave-uniform
    unsigned aw = (tid >> 6) * 104729 + blockIdx.x * 977 + 99;   // wave-uniform
    const unsigned brick = (blockIdx.x * 2654435761u) % (nrows128 - 4096);
    for (int it = 0; it < iters; ++it) {
        a = a * 1664525u + 1013904223u;
        aw = aw * 1664525u + 1013904223u;
        if (MODE == G_F32_ROW128) {   // each half-wave adds into one random 128-B row (32 floats)
            const unsigned row = (a >> 8) % nrows128;
            atomicAdd(g + (size_t)row * 32 + (lane & 31), 1.0f);
        }
        if (MODE == G_F32_LOCAL) {    // rows within a 4096-row (512 KB) neighbourhood of the block's brick: L2-local
            const unsigned row = brick + ((a >> 8) & 4095);
            atomicAdd(g + (size_t)row * 32 + (lane & 31), 1.0f);
        }
        if (MODE == G_F32_ROW256) {   // the wave adds into one random 256-B row (64 floats)
            const unsigned row = (aw >> 8) % (nrows128 / 2);
            atomicAdd(g + (size_t)row * 64 + lane, 1.0f);
        }
        if (MODE == G_F64) {          // half-wave: 32 doubles = 256 B
            const unsigned row = (a >> 8) % (nrows128 / 2);
            atomicAdd((double *)g + (size_t)row * 32 + (lane & 31), 1.0);
        }
        if (MODE == G_U64) {
            const unsigned row = (a >> 8) % (nrows128 / 2);
            atomicAdd((unsigned long long *)g + (size_t)row * 32 + (lane & 31), 1ull);
        }
        if (MODE == G_PKBF16) {       // 16 lanes x 2 bf16... here: each lane adds a packed pair (64 B per half-wave)
            const unsigned row = (a >> 8) % nrows128;
            typedef short bf2 __attribute__((ext_vector_type(2)));
            bf2 v = {0x3f80, 0x3f80};
            __builtin_amdgcn_global_atomic_fadd_v2bf16((bf2 __attribute__((address_space(1))) *)(g + (size_t)row * 32 + (lane & 31)), v);
        }
    }
}

template <typename F>
static float timeit(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float *g;
    const size_t gbytes = 64ull << 20;   // 64 MB
    hipMalloc(&g, gbytes);
    hipMemset(g, 0, gbytes);
    const int blocks = 2048, iters = 2000;
    const double n = (double)blocks * 256 * iters;
#define RUN_L(M, name, elems)                                                                            \
    {                                                                                                    \
        float ms = timeit([&] { k_lds<M><<<blocks, 256>>>(g, iters); });                                  \
        printf("LDS %-28s %8.3f ms  %8.1f G lane-ops/s  %6.2f lane-ops/clk/CU  %6.2f elems/clk/CU\n", name, ms, n / ms / 1e6, \
               n / ms / 1e6 / 256 / 2.4, n * elems / ms / 1e6 / 256 / 2.4);                              \
    }
    RUN_L(L_F32, "ds_add_f32", 1)
    RUN_L(L_F32_RTN, "ds_add_rtn_f32", 1)
    RUN_L(L_F64, "ds_add_f64", 1)
    RUN_L(L_U32, "ds_add_u32", 1)
    RUN_L(L_U64, "ds_add_u64", 1)
    RUN_L(L_RMW32, "read-add-write b32 (4 waves)", 1)
    RUN_L(L_RMW64, "read-add-write b64 (4 waves)", 2)
    RUN_L(L_RMW128, "read-add-write b128 (4 waves)", 4)
    const unsigned rows_all = (unsigned)(gbytes / 128), rows_8mb = (8u << 20) / 128;
#define RUN_G(M, name, rows)                                                                             \
    {                                                                                                    \
        float ms = timeit([&] { k_glb<M><<<blocks, 256>>>(g, iters / 4, rows); });                        \
        printf("GLB %-28s %8.3f ms  %8.1f G lane-atomics/s\n", name, ms, n / 4 / ms / 1e6);              \
    }
    RUN_G(G_F32_ROW128, "f32 128B rows over 8MB", rows_8mb)
    RUN_G(G_F32_ROW128, "f32 128B rows over 64MB", rows_all)
    RUN_G(G_F32_LOCAL, "f32 128B rows, 512KB/block", rows_8mb)
    RUN_G(G_F32_ROW256, "f32 256B rows over 8MB", rows_8mb)
    RUN_G(G_F64, "f64 256B rows over 8MB", rows_8mb)
    RUN_G(G_U64, "u64 256B rows over 8MB", rows_8mb)
    RUN_G(G_PKBF16, "pk_bf16 128B rows over 8MB", rows_8mb)
    hipFree(g);
    return 0;
}
