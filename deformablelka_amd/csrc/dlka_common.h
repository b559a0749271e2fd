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


// ---- activation storage type of the channels-last kernels: float, or bf16_t (DLKA_BF16: half the bytes, fp32 arithmetic) ----
// Kernels keep their `const float *` argument blocks and reinterpret them; byte offsets scale with sizeof(T).
template <typename T> __device__ __forceinline__ f32x4 act_buf_load4(BufRsrc r, unsigned byteoff);   // 4 consecutive elements
template <> __device__ __forceinline__ f32x4 act_buf_load4<float>(BufRsrc r, unsigned byteoff) { return buf_load_f32x4(r, byteoff); }
template <> __device__ __forceinline__ f32x4 act_buf_load4<bf16_t>(BufRsrc r, unsigned byteoff) { return buf_load_bf16x4(r, byteoff); }
template <typename T> __device__ __forceinline__ float act_buf_load1(BufRsrc r, unsigned byteoff);
template <> __device__ __forceinline__ float act_buf_load1<float>(BufRsrc r, unsigned byteoff) { return buf_load_f32(r, byteoff); }
template <> __device__ __forceinline__ float act_buf_load1<bf16_t>(BufRsrc r, unsigned byteoff) { return buf_load_bf16(r, byteoff); }
// the same with a wave-uniform offset on top of the (range-checked) per-lane one — see buf_load_f32_s
template <typename T> __device__ __forceinline__ float act_buf_load1_s(BufRsrc r, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ float act_buf_load1_s<float>(BufRsrc r, unsigned voff, unsigned soff) { return buf_load_f32_s(r, voff, soff); }
template <> __device__ __forceinline__ float act_buf_load1_s<bf16_t>(BufRsrc r, unsigned voff, unsigned soff) { return buf_load_bf16_s(r, voff, soff); }
// pointer flavours: element index, 4 consecutive elements (aligned to 4 elements)
struct alignas(8) ActU2 { unsigned x, y; };
__device__ __forceinline__ f32x4 act_load4(const float *p, long i) { return *reinterpret_cast<const f32x4 *>(p + i); }
__device__ __forceinline__ f32x4 act_load4(const bf16_t *p, long i)
{
    const ActU2 w = *reinterpret_cast<const ActU2 *>(p + i);
    f32x4 v;
    v[0] = __uint_as_float(w.x << 16); v[1] = __uint_as_float(w.x & 0xffff0000u);
    v[2] = __uint_as_float(w.y << 16); v[3] = __uint_as_float(w.y & 0xffff0000u);
    return v;
}
__device__ __forceinline__ void act_store4(float *p, long i, f32x4 v) { *reinterpret_cast<f32x4 *>(p + i) = v; }
__device__ __forceinline__ void act_store4(bf16_t *p, long i, f32x4 v)
{
    ActU2 w;
    w.x = (unsigned)bf16_bits(v[0]) | ((unsigned)bf16_bits(v[1]) << 16);
    w.y = (unsigned)bf16_bits(v[2]) | ((unsigned)bf16_bits(v[3]) << 16);
    *reinterpret_cast<ActU2 *>(p + i) = w;
}
__device__ __forceinline__ float act_load1(const float *p, long i) { return p[i]; }
__device__ __forceinline__ float act_load1(const bf16_t *p, long i) { return __uint_as_float((unsigned)p[i].v << 16); }
__device__ __forceinline__ void act_store1(float *p, long i, float v) { p[i] = v; }
__device__ __forceinline__ void act_store1(bf16_t *p, long i, float v) { p[i].v = bf16_bits(v); }
// two consecutive elements (i even)
__device__ __forceinline__ void act_store2(float *p, long i, float v0, float v1) { *reinterpret_cast<f32x2 *>(p + i) = f32x2{v0, v1}; }
__device__ __forceinline__ void act_store2(bf16_t *p, long i, float v0, float v1)
{
    *reinterpret_cast<unsigned *>(p + i) = (unsigned)bf16_bits(v0) | ((unsigned)bf16_bits(v1) << 16);
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

// Hot-loop index decoding WITHOUT runtime integer divisions.  A division by a run-time value compiles to ~35 dependent instructions (v_rcp_iflag,
// v_mul_hi, corrections); s_memtime stamps inside the deformable forward kernel (scripts/fwd_stamps.py, round 3) showed "unit -> tap", "tap -> (i, j, k)"
// and the description that follows taking a THIRD of every (tile, tap) step — five such divisions on the wave's critical path.  The D-LKA block's
// kernels are 3^3 (2-D offset nets: 5x5, 7x7) and its channel-chunk counts are powers of two: constant divisors become multiply-shift, powers of
// two a shift; anything else still takes the general form.  All branches are wave-uniform.
__device__ __forceinline__ void tap_decode(int tap, int kw, int kh, int &ti, int &tj, int &tk)
{
    if (kw == 3 && kh == 3) { ti = tap / 9; const int r = tap - 9 * ti; tj = r / 3; tk = r - 3 * tj; }
    else if (kw == 5 && kh == 5) { ti = tap / 25; const int r = tap - 25 * ti; tj = r / 5; tk = r - 5 * tj; }
    else if (kw == 7 && kh == 7) { ti = tap / 49; const int r = tap - 49 * ti; tj = r / 7; tk = r - 7 * tj; }
    else { tk = tap % kw; tj = (tap / kw) % kh; ti = tap / (kw * kh); }
}
// q = x / n, r = x % n for x >= 0, n > 0
__device__ __forceinline__ int divmod_fast(int x, int n, int &r)
{
    if ((n & (n - 1)) == 0) { const int sh = __builtin_ctz((unsigned)n); r = x & (n - 1); return x >> sh; }
    const int q = x / n;
    r = x - q * n;
    return q;
}

#define DLKA_THREADS 256

// Every kernel launch of the library goes through DLKA_LAUNCH.  With the launch trace switched on (dlka_trace_start, include/dlka.h: a
// measurement aid for bench.py's `roofline` block) a timing event is recorded on the launch's own stream right behind each launch; the
// interval between consecutive events is that kernel's duration as the timed step really runs it (same arguments, same predecessor state).
// Off (the normal state): one predictable branch on a global.  Never switch it on during hipGraph capture.
#if defined(HIPEMU)
#define DLKA_LAUNCH(kernel, grid, block, shmem, stream, ...) hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)
#else
extern int g_trace_on;
void trace_after_launch(const void *kernel_fn, hipStream_t st);
#define DLKA_LAUNCH(kernel, grid, block, shmem, stream, ...)                                             \
    do {                                                                                                 \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                             \
        if (dlka::g_trace_on) dlka::trace_after_launch(reinterpret_cast<const void *>(kernel), stream);  \
    } while (0)
#endif

#define DLKA_CHECK_LAUNCH()                                  \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return DLKA_ERR_LAUNCH; \
    } while (0)

}  // namespace dlka
