// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// A tiny CPU "wavefront emulator" that stands in for <hip/hip_runtime.h> when the kernel sources under
// deformablelka_amd/csrc are compiled with a HOST compiler (tests/emu/build.py: clang++ -I tests/emu/include).
// Purpose: check kernel index arithmetic, LDS tiling, barrier placement and MFMA fragment bookkeeping in the
// CPU-only container before a GPU minute is spent.  The product library (libdlka_hip.so) is built by hipcc
// against the real HIP runtime and never sees this file.
//
// Model: one workgroup at a time per OS thread; every work-item is a ucontext fiber; __syncthreads() and the
// wave-collective operations (shuffle, ballot, readfirstlane, MFMA) are rendezvous points served by a
// round-robin scheduler.  Wave = 64 consecutive work-items (x fastest), as on gfx950.
#pragma once
#define HIPEMU 1

#include <ucontext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { unsigned x, y, z; };

struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
using std::max;
using std::min;

typedef void *hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum hipMemcpyKind { hipMemcpyDeviceToDevice = 3 };
// streams / events: everything is synchronous in the emulator
typedef void *hipEvent_t;
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }

namespace hipemu {

constexpr int WAVE = 64;
constexpr size_t STACK = 96 * 1024;

enum State { RUN = 0, BARRIER = 1, WAVEOP = 2, DONE = 3 };

struct Fiber {
    ucontext_t ctx;
    char *stack = nullptr;
    uint3 tid;
    int lin = 0;   // linear id in block
    int state = DONE;
};

struct alignas(16) Slot { unsigned char b[160]; };  // per-lane deposit for collectives (a,b,c of an MFMA fit)

struct Block {
    dim3 grid, block;
    uint3 bid;
    std::vector<Fiber> fib;
    int nthreads = 0;
    int cur = -1;
    ucontext_t sched;
    std::vector<unsigned char> dyn;       // dynamic LDS
    std::vector<Slot> xchg[2];            // [parity][lin]
    std::vector<int> waveop_count;        // per wave: collective sequence number
    std::function<void()> body;
};

inline Block *&tls_block() { static thread_local Block *b = nullptr; return b; }
inline Block &B() { return *tls_block(); }
inline Fiber &F() { Block &b = B(); return b.fib[b.cur]; }

inline void yield_to_sched() { Block &b = B(); Fiber &f = b.fib[b.cur]; swapcontext(&f.ctx, &b.sched); }

inline void trampoline() {
    Block &b = B();
    b.body();
    b.fib[b.cur].state = DONE;
    // returning resumes uc_link (= sched)
}

inline void run_block(Block &b) {
    tls_block() = &b;
    const int n = b.nthreads;
    for (int i = 0; i < n; ++i) {
        Fiber &f = b.fib[i];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = &b.sched;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
        f.state = RUN;
        f.lin = i;
        f.tid.x = i % b.block.x;
        f.tid.y = (i / b.block.x) % b.block.y;
        f.tid.z = i / (b.block.x * b.block.y);
    }
    std::fill(b.waveop_count.begin(), b.waveop_count.end(), 0);
    const int nw = (n + WAVE - 1) / WAVE;
    int live = n;
    while (live > 0) {
        bool progressed = false;
        for (int i = 0; i < n; ++i) {
            if (b.fib[i].state == RUN) {
                b.cur = i;
                swapcontext(&b.sched, &b.fib[i].ctx);
                progressed = true;
            }
        }
        live = 0;
        int at_bar = 0;
        for (int i = 0; i < n; ++i) {
            if (b.fib[i].state != DONE) ++live;
            if (b.fib[i].state == BARRIER) ++at_bar;
        }
        bool released = false;
        if (live > 0 && at_bar == live) {  // finished work-items do not take part (matches s_barrier semantics)
            for (int i = 0; i < n; ++i) if (b.fib[i].state == BARRIER) b.fib[i].state = RUN;
            released = true;
        }
        for (int w = 0; w < nw; ++w) {
            int lo = w * WAVE, hi = std::min(n, lo + WAVE), lv = 0, at = 0;
            for (int i = lo; i < hi; ++i) {
                if (b.fib[i].state != DONE) ++lv;
                if (b.fib[i].state == WAVEOP) ++at;
            }
            if (lv > 0 && at == lv) {
                for (int i = lo; i < hi; ++i) if (b.fib[i].state == WAVEOP) b.fib[i].state = RUN;
                released = true;
            }
        }
        if (live > 0 && !progressed && !released) {
            fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %d live, %d at barrier (divergent barrier / wave op?)\n",
                    b.bid.x, b.bid.y, b.bid.z, live, at_bar);
            abort();
        }
    }
}

// ---- collectives -----------------------------------------------------------------------------
inline void barrier() { F().state = BARRIER; yield_to_sched(); }

template <typename Kernel, typename... Args>
void launch(Kernel k, dim3 grid, dim3 block, size_t shmem, Args... args) {
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    const int nthreads = (int)(block.x * block.y * block.z);
    unsigned hw = std::thread::hardware_concurrency();
    const char *env = getenv("HIPEMU_THREADS");
    if (env) hw = (unsigned)atoi(env);
    const unsigned nworkers = (unsigned)std::max<size_t>(1, std::min<size_t>(hw ? hw : 1, nblocks));
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        Block b;
        b.grid = grid; b.block = block; b.nthreads = nthreads;
        b.fib.resize(nthreads);
        for (auto &f : b.fib) f.stack = (char *)malloc(STACK);
        b.dyn.assign(shmem + 64, 0);
        b.xchg[0].resize(nthreads); b.xchg[1].resize(nthreads);
        b.waveop_count.assign((nthreads + WAVE - 1) / WAVE, 0);
        b.body = [&]() { k(args...); };
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= nblocks) break;
            b.bid.x = (unsigned)(i % grid.x);
            b.bid.y = (unsigned)((i / grid.x) % grid.y);
            b.bid.z = (unsigned)(i / ((size_t)grid.x * grid.y));
            run_block(b);
        }
        for (auto &f : b.fib) free(f.stack);
        tls_block() = nullptr;
    };
    if (nworkers == 1) { worker(); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nworkers; ++t) th.emplace_back(worker);
    for (auto &t : th) t.join();
}

inline unsigned char *dyn_smem() {
    Block &b = B();
    uintptr_t p = (uintptr_t)b.dyn.data();
    return (unsigned char *)((p + 15) & ~(uintptr_t)15);
}

}  // namespace hipemu

#define threadIdx (hipemu::F().tid)
#define blockIdx (hipemu::B().bid)
#define blockDim (hipemu::B().block)
#define gridDim (hipemu::B().grid)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__)

inline void __syncthreads() { hipemu::barrier(); }

// ---- atomics (blocks run on several OS threads) ------------------------------------------------
inline float atomicAdd(float *addr, float v) {
    auto *a = reinterpret_cast<std::atomic<uint32_t> *>(addr);
    uint32_t old = a->load(std::memory_order_relaxed), nw;
    float f;
    do {
        memcpy(&f, &old, 4);
        f += v;
        memcpy(&nw, &f, 4);
    } while (!a->compare_exchange_weak(old, nw, std::memory_order_relaxed));
    memcpy(&f, &old, 4);
    return f;
}
inline double atomicAdd(double *addr, double v) {
    auto *a = reinterpret_cast<std::atomic<uint64_t> *>(addr);
    uint64_t old = a->load(std::memory_order_relaxed), nw;
    double f;
    do {
        memcpy(&f, &old, 8);
        f += v;
        memcpy(&nw, &f, 8);
    } while (!a->compare_exchange_weak(old, nw, std::memory_order_relaxed));
    memcpy(&f, &old, 8);
    return f;
}
inline int atomicAdd(int *addr, int v) { return reinterpret_cast<std::atomic<int> *>(addr)->fetch_add(v); }
inline unsigned long long atomicAdd(unsigned long long *addr, unsigned long long v) {
    return reinterpret_cast<std::atomic<unsigned long long> *>(addr)->fetch_add(v);
}
inline unsigned atomicMax(unsigned *addr, unsigned v) {
    auto *a = reinterpret_cast<std::atomic<unsigned> *>(addr);
    unsigned old = a->load(std::memory_order_relaxed);
    while (old < v && !a->compare_exchange_weak(old, v, std::memory_order_relaxed)) {}
    return old;
}

// ---- wave collectives ---------------------------------------------------------------------------
namespace hipemu {
template <typename T>
inline T shfl_any(T v, int src_lane) {
    Fiber &f = F();
    const int w0 = (f.lin / WAVE) * WAVE;
    Block &b = B();
    // parity by private counter stored in the slot area is not needed: use two rendezvous (simple, safe)
    memcpy(b.xchg[0][f.lin].b, &v, sizeof(T));
    f.state = WAVEOP; yield_to_sched();
    T r;
    int s = w0 + (src_lane & (WAVE - 1));
    if (s >= b.nthreads || b.fib[s].state == DONE) s = f.lin;
    memcpy(&r, b.xchg[0][s].b, sizeof(T));
    f.state = WAVEOP; yield_to_sched();  // nobody overwrites before everyone has read
    return r;
}
}  // namespace hipemu

template <typename T> inline T __shfl(T v, int src, int = 64) { return hipemu::shfl_any(v, src); }
template <typename T> inline T __shfl_xor(T v, int mask, int = 64) { return hipemu::shfl_any(v, (hipemu::F().lin & 63) ^ mask); }
template <typename T> inline T __shfl_down(T v, unsigned d, int = 64) {
    int l = hipemu::F().lin & 63;
    return hipemu::shfl_any(v, (l + (int)d < 64) ? l + (int)d : l);
}
template <typename T> inline T __shfl_up(T v, unsigned d, int = 64) {
    int l = hipemu::F().lin & 63;
    return hipemu::shfl_any(v, (l - (int)d >= 0) ? l - (int)d : l);
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline unsigned long long __ballot(int pred) {
    using namespace hipemu;
    Fiber &f = F();
    Block &b = B();
    const int w0 = (f.lin / WAVE) * WAVE;
    int p = pred ? 1 : 0;
    memcpy(b.xchg[0][f.lin].b, &p, sizeof(int));
    f.state = WAVEOP; yield_to_sched();
    unsigned long long m = 0;
    for (int i = 0; i < WAVE && w0 + i < b.nthreads; ++i) {
        if (b.fib[w0 + i].state == DONE) continue;
        int q; memcpy(&q, b.xchg[0][w0 + i].b, sizeof(int));
        if (q) m |= 1ull << i;
    }
    f.state = WAVEOP; yield_to_sched();
    return m;
}

// ---- intrinsic wrappers used by csrc/dlka_intrin.h ------------------------------------------------
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));

namespace hipemu {
template <typename T> inline T readfirstlane(T v) {
    Fiber &f = F(); Block &b = B();
    const int w0 = (f.lin / WAVE) * WAVE;
    int first = w0;
    memcpy(b.xchg[0][f.lin].b, &v, sizeof(T));
    f.state = WAVEOP; yield_to_sched();
    while (first < b.nthreads && b.fib[first].state == DONE) ++first;
    T r; memcpy(&r, b.xchg[0][first].b, sizeof(T));
    f.state = WAVEOP; yield_to_sched();
    return r;
}

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
// (guide: cdna_hip_programming.md §3).  Result == k-ordered fmaf chain.
inline hipemu_f32x16 mfma_f32_32x32x2f32(float a, float bv, hipemu_f32x16 c) {
    Fiber &f = F(); Block &b = B();
    const int w0 = (f.lin / WAVE) * WAVE, l = f.lin - w0;
    float ab[2] = {a, bv};
    memcpy(b.xchg[0][f.lin].b, ab, sizeof(ab));
    f.state = WAVEOP; yield_to_sched();
    hipemu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float A[2], Bm[2];
            memcpy(A, b.xchg[0][w0 + row + 32 * k].b, sizeof(A));
            memcpy(Bm, b.xchg[0][w0 + col + 32 * k].b, sizeof(Bm));
            acc = fmaf(A[0], Bm[1], acc);
        }
        d[r] = acc;
    }
    f.state = WAVEOP; yield_to_sched();
    return d;
}

// v_mfma_f32_32x32x16_bf16: lane (i = l&31, g = l>>5) holds A[i][8g..8g+7] and B[8g..8g+7][j = l&31] as bf16 (raw 16-bit patterns
// here); D as 32x32x2.  Products of bf16 values are exact in fp32; accumulation in fp32.
struct hipemu_bf16x8 { unsigned short v[8]; };
inline float hipemu_bf16_to_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
inline unsigned short hipemu_f32_to_bf16(float x) {   // round to nearest even
    unsigned u; memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
inline hipemu_f32x16 mfma_f32_32x32x16bf16(hipemu_bf16x8 a, hipemu_bf16x8 bv, hipemu_f32x16 c) {
    Fiber &f = F(); Block &b = B();
    const int w0 = (f.lin / WAVE) * WAVE, l = f.lin - w0;
    unsigned short ab[16];
    memcpy(ab, a.v, 16); memcpy(ab + 8, bv.v, 16);
    memcpy(b.xchg[0][f.lin].b, ab, sizeof(ab));
    f.state = WAVEOP; yield_to_sched();
    hipemu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int g = 0; g < 2; ++g) {
            unsigned short A[16], Bm[16];
            memcpy(A, b.xchg[0][w0 + row + 32 * g].b, sizeof(A));
            memcpy(Bm, b.xchg[0][w0 + col + 32 * g].b, sizeof(Bm));
            for (int e = 0; e < 8; ++e) acc += hipemu_bf16_to_f32(A[e]) * hipemu_bf16_to_f32(Bm[8 + e]);
        }
        d[r] = acc;
    }
    f.state = WAVEOP; yield_to_sched();
    return d;
}

// v_mfma_f32_16x16x32_bf16 (gfx950): lane (i = l&15, g = l>>4) holds A[i][8g .. 8g+7] and B[8g .. 8g+7][j = l&15]; D: col=l&15, row=(l>>4)*4+r
inline hipemu_f32x4 mfma_f32_16x16x32bf16(hipemu_bf16x8 a, hipemu_bf16x8 bv, hipemu_f32x4 c) {
    Fiber &f = F(); Block &b = B();
    const int w0 = (f.lin / WAVE) * WAVE, l = f.lin - w0;
    unsigned short ab[16];
    memcpy(ab, a.v, 16); memcpy(ab + 8, bv.v, 16);
    memcpy(b.xchg[0][f.lin].b, ab, sizeof(ab));
    f.state = WAVEOP; yield_to_sched();
    hipemu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int g = 0; g < 4; ++g) {
            unsigned short A[16], Bm[16];
            memcpy(A, b.xchg[0][w0 + row + 16 * g].b, sizeof(A));
            memcpy(Bm, b.xchg[0][w0 + col + 16 * g].b, sizeof(Bm));
            for (int e = 0; e < 8; ++e) acc += hipemu_bf16_to_f32(A[e]) * hipemu_bf16_to_f32(Bm[8 + e]);
        }
        d[r] = acc;
    }
    f.state = WAVEOP; yield_to_sched();
    return d;
}

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=(l>>4)*4+r
inline hipemu_f32x4 mfma_f32_16x16x4f32(float a, float bv, hipemu_f32x4 c) {
    Fiber &f = F(); Block &b = B();
    const int w0 = (f.lin / WAVE) * WAVE, l = f.lin - w0;
    float ab[2] = {a, bv};
    memcpy(b.xchg[0][f.lin].b, ab, sizeof(ab));
    f.state = WAVEOP; yield_to_sched();
    hipemu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float A[2], Bm[2];
            memcpy(A, b.xchg[0][w0 + row + 16 * k].b, sizeof(A));
            memcpy(Bm, b.xchg[0][w0 + col + 16 * k].b, sizeof(Bm));
            acc = fmaf(A[0], Bm[1], acc);
        }
        d[r] = acc;
    }
    f.state = WAVEOP; yield_to_sched();
    return d;
}
}  // namespace hipemu
