// Elementwise pieces of the D-LKA block: exact-erf GELU (nn.GELU() default,
// 3D/d_lka_former/network_architecture/synapse/transformerblock.py:660), the u*attn gate (:652) and the
// residual add (:671).  Pure HBM-bound streaming: 16-byte vector accesses, grid-stride.
#include "cl_args.h"
#include "dlka_kernels.h"

namespace dlka {


struct OpGelu { __device__ __forceinline__ float operator()(float x) const { return gelu_f(x); } };

template <typename T, int MODE>
__global__ __launch_bounds__(256) void eltwise_kernel(const T *a, const T *b, const T *c,
                                                      T *o1, T *o2, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (MODE == 0) {  // gelu fwd: o1 = gelu(a)
            stf(o1 + i, gelu_f(ldf(a + i)));
        } else if (MODE == 1) {  // gelu bwd: o1 = b * dgelu(a)
            stf(o1 + i, ldf(b + i) * dgelu_f(ldf(a + i)));
        } else if (MODE == 2) {  // mul fwd
            stf(o1 + i, ldf(a + i) * ldf(b + i));
        } else if (MODE == 3) {  // mul bwd: o1 = c*b ; o2 = c*a
            const float gy = ldf(c + i);
            const float av = ldf(a + i), bv = ldf(b + i);
            if (o1) stf(o1 + i, gy * bv);
            if (o2) stf(o2 + i, gy * av);
        } else if (MODE == 4) {  // add
            stf(o1 + i, ldf(a + i) + ldf(b + i));
        } else {  // gelu bwd of a sum: o1 = (b + c) * dgelu(a)
            stf(o1 + i, (ldf(b + i) + ldf(c + i)) * dgelu_f(ldf(a + i)));
        }
    }
}

template <typename T, int MODE>
static int launch_elt(const T *a, const T *b, const T *c, T *o1, T *o2, long n, hipStream_t st)
{
    if (n <= 0) return DLKA_OK;
    long blocks = cdivl(n, 256);
    if (blocks > 2048) blocks = 2048;
    auto k = eltwise_kernel<T, MODE>;
    DLKA_LAUNCH(k, dim3((unsigned)blocks), dim3(256), 0, st, a, b, c, o1, o2, n);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

template <typename T> int launch_gelu_fwd(const T *x, T *y, long n, hipStream_t st) { return launch_elt<T, 0>(x, nullptr, nullptr, y, nullptr, n, st); }
template <typename T> int launch_gelu_bwd(const T *x, const T *gy, T *gx, long n, hipStream_t st) { return launch_elt<T, 1>(x, gy, nullptr, gx, nullptr, n, st); }
template <typename T> int launch_mul_fwd(const T *a, const T *b, T *y, long n, hipStream_t st) { return launch_elt<T, 2>(a, b, nullptr, y, nullptr, n, st); }
template <typename T> int launch_mul_bwd(const T *a, const T *b, const T *gy, T *ga, T *gb, long n, hipStream_t st) { return launch_elt<T, 3>(a, b, gy, ga, gb, n, st); }
template <typename T> int launch_add_fwd(const T *a, const T *b, T *y, long n, hipStream_t st) { return launch_elt<T, 4>(a, b, nullptr, y, nullptr, n, st); }
template <typename T> int launch_gelu_bwd_sum(const T *x, const T *g1, const T *g2, T *gx, long n, hipStream_t st) { return launch_elt<T, 5>(x, g1, g2, gx, nullptr, n, st); }

// Zero fill as an ordinary kernel.  (hipMemsetAsync nodes captured into a hipGraph did not reproduce the eager result on
// ROCm 7.2 / MI355X: from the second replay on, buffers zeroed this way held garbage — scripts/debug_graph.py.)
__global__ __launch_bounds__(256) void zero_fill_kernel(float *__restrict__ p, long n4, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) reinterpret_cast<f32x4 *>(p)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0.f;
}

int launch_zero(void *ptr, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return DLKA_OK;
    if (((uintptr_t)ptr & 15) || (bytes & 3)) return hipMemsetAsync(ptr, 0, bytes, st) == hipSuccess ? DLKA_OK : DLKA_ERR_LAUNCH;
    const long n = (long)(bytes / 4), n4 = n / 4;
    long blocks = cdivl(n4 > 0 ? n4 : n, 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    DLKA_LAUNCH(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (float *)ptr, n4, n);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

// one workgroup clears 4096 floats of one region
__global__ __launch_bounds__(256) void zero_batch_kernel(ZeroBatch b)
{
    int r = 0;
    while (r + 1 < b.n && blockIdx.x >= b.block0[r + 1]) ++r;
    float *p = b.p[r];
    const long n = b.cnt[r], base = (long)(blockIdx.x - b.block0[r]) * 4096;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long i = base + (long)(k * 256 + threadIdx.x) * 4;
        if (i + 3 < n) *reinterpret_cast<f32x4 *>(p + i) = f32x4{0.f, 0.f, 0.f, 0.f};
        else for (long j = i; j < n; ++j) p[j] = 0.f;
    }
}

int launch_zero_batch(ZeroBatch &b, hipStream_t st)
{
    if (b.overflow) return DLKA_ERR_WORKSPACE;
    if (b.n <= 0) return DLKA_OK;
    unsigned blk = 0;
    for (int r = 0; r < b.n; ++r) {
        if ((uintptr_t)b.p[r] & 15) return DLKA_ERR_UNSUPPORTED;
        b.block0[r] = blk;
        blk += (unsigned)cdivl(b.cnt[r], 4096);
    }
    b.block0[b.n] = blk;
    DLKA_LAUNCH(zero_batch_kernel, dim3(blk), dim3(256), 0, st, b);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

#define DLKA_INST(T)                                                                    \
    template int launch_gelu_fwd<T>(const T *, T *, long, hipStream_t);                  \
    template int launch_gelu_bwd<T>(const T *, const T *, T *, long, hipStream_t);       \
    template int launch_mul_fwd<T>(const T *, const T *, T *, long, hipStream_t);        \
    template int launch_mul_bwd<T>(const T *, const T *, const T *, T *, T *, long, hipStream_t); \
    template int launch_add_fwd<T>(const T *, const T *, T *, long, hipStream_t);         \
    template int launch_gelu_bwd_sum<T>(const T *, const T *, const T *, T *, long, hipStream_t);
DLKA_INST(float)
DLKA_INST(bf16_t)
#undef DLKA_INST

}  // namespace dlka
