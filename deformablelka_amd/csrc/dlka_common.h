// Shared device/host helpers for the D-LKA HIP kernels (gfx950).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "dlka.h"
#include "dlka_intrin.h"

namespace dlka {

// ---------------------------------------------------------------------------------------------
// storage types: fp32, or bf16 storage with fp32 arithmetic
// ---------------------------------------------------------------------------------------------
struct bf16_t { uint16_t v; };

__host__ __device__ __forceinline__ float ldf(const float *p) { return *p; }
__host__ __device__ __forceinline__ float ldf(const bf16_t *p) {
    uint32_t u = (uint32_t)p->v << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__host__ __device__ __forceinline__ void stf(float *p, float x) { *p = x; }
__host__ __device__ __forceinline__ void stf(bf16_t *p, float x) {  // round-to-nearest-even, NaN kept quiet
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) { p->v = (uint16_t)((u >> 16) | 0x40); return; }
    u += 0x7fffu + ((u >> 16) & 1u);
    p->v = (uint16_t)(u >> 16);
}

// ---------------------------------------------------------------------------------------------
// geometry with the derived sizes every kernel needs
// ---------------------------------------------------------------------------------------------
struct Geom {
    int B, C, D, H, W, Cout;
    int kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw;
    int group, dg;
    int Do, Ho, Wo;
    int K, Cg, Og, cpdg;  // taps, in-channels per group, out-channels per group, channels per deformable group
    int No, Ni;           // output / input voxels per (b, channel) plane
};

__host__ __device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ __forceinline__ long cdivl(long a, long b) { return (a + b - 1) / b; }
__host__ __device__ __forceinline__ int round_up(int a, int b) { return cdiv(a, b) * b; }

// exact (erf) GELU and its derivative, as nn.GELU() in the reference block (transformerblock.py:660)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_f(float x)
{
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Workgroup b of a launch runs on XCD b % 8 (observed placement, MI355X_MICROARCH.md: used for speed only, never for correctness), and
// each XCD has its own 4 MB L2.  A launch whose work items are ordered so that neighbours share data (row chunks of a volume) therefore
// wants item ranges, not item stripes, per XCD: block b -> item (b % 8) * ceil(total / 8) + b / 8.  The grid is 8 * ceil(total / 8)
// blocks; the few blocks beyond `total` return -1.
__host__ __device__ __forceinline__ int xcd_item(int block, int total)
{
    const int per = (total + 7) / 8;
    const int r = (block % 8) * per + block / 8;
    return r < total ? r : -1;
}
__host__ __device__ __forceinline__ int xcd_grid(int total) { return 8 * ((total + 7) / 8); }
// blockIdx.x of a launch whose x dimension enumerates spatial tiles: nx > 0 = swizzled (gridDim.x == xcd_grid(nx); since that is a multiple
// of 8, the XCD of a block is blockIdx.x % 8 whatever blockIdx.y / z are), -1 for the padding blocks; nx == 0 = plain.
#define DLKA_XCD_BX(nx) ((nx) > 0 ? xcd_item((int)blockIdx.x, (nx)) : (int)blockIdx.x)
inline bool xcd_swizzle_enabled()
{
    static const bool on = getenv("DLKA_NO_XCD_SWIZZLE") == nullptr;
    return on;
}
// launches with fewer tiles than this keep the plain order (nothing to gain); DLKA_XCD_MIN lowers it so that small test shapes take the
// swizzled path too (not cached: tests toggle it)
inline int xcd_min_blocks()
{
    const char *e = getenv("DLKA_XCD_MIN");
    return e ? atoi(e) : 16;
}

#define DLKA_THREADS 256

#define DLKA_CHECK_LAUNCH()                                  \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return DLKA_ERR_LAUNCH; \
    } while (0)

}  // namespace dlka
