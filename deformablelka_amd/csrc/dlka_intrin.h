// Thin names for the gfx950 intrinsics the kernels use.  The HIPEMU branch exists only so that
// tests/emu can compile the kernel sources with a host compiler (test infrastructure, see
// tests/emu/include/hip/hip_runtime.h); hipcc never defines HIPEMU.
#pragma once
#include <hip/hip_runtime.h>

namespace dlka {
#define DLKA_OOB 0x80000000u   // byte offset beyond any buffer the gather kernels accept (< 2 GB)
#if defined(HIPEMU)
typedef hipemu_f32x4 f32x4;
typedef hipemu_f32x16 f32x16;
__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) { return hipemu::mfma_f32_32x32x2f32(a, b, c); }
__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) { return hipemu::mfma_f32_16x16x4f32(a, b, c); }
__device__ __forceinline__ int readfirstlane(int v) { return hipemu::readfirstlane(v); }
typedef hipemu::hipemu_bf16x8 bf16x8;
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return hipemu::mfma_f32_32x32x16bf16(a, b, c); }
__device__ __forceinline__ f32x4 mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) { return hipemu::mfma_f32_16x16x32bf16(a, b, c); }
// a[e] = hi[e] + lo[e] + O(2^-18 |a[e]|): two-term bf16 split of 8 fp32 values
__device__ __forceinline__ void split_bf16x8(const float *a, bf16x8 &hi, bf16x8 &lo)
{
    for (int e = 0; e < 8; ++e) {
        hi.v[e] = hipemu::hipemu_f32_to_bf16(a[e]);
        lo.v[e] = hipemu::hipemu_f32_to_bf16(a[e] - hipemu::hipemu_bf16_to_f32(hi.v[e]));
    }
}
// a[e] = hi[e] + mid[e] + lo[e] up to fp32 rounding: three bf16 terms carry the 24-bit significand
__device__ __forceinline__ void split3_bf16x8(const float *a, bf16x8 &hi, bf16x8 &mid, bf16x8 &lo)
{
    for (int e = 0; e < 8; ++e) {
        hi.v[e] = hipemu::hipemu_f32_to_bf16(a[e]);
        const float r1 = a[e] - hipemu::hipemu_bf16_to_f32(hi.v[e]);
        mid.v[e] = hipemu::hipemu_f32_to_bf16(r1);
        lo.v[e] = hipemu::hipemu_f32_to_bf16(r1 - hipemu::hipemu_bf16_to_f32(mid.v[e]));
    }
}
__device__ __forceinline__ unsigned short bf16_bits(float x) { return hipemu::hipemu_f32_to_bf16(x); }
// 8 bf16 values that are already packed in four 32-bit words (element 0 = low half of word 0): no conversion
__device__ __forceinline__ bf16x8 bf16x8_from_words(const float *w4) { bf16x8 r; memcpy(r.v, w4, 16); return r; }
__device__ __forceinline__ float bf16_value(unsigned short h) { return hipemu::hipemu_bf16_to_f32(h); }
#define DLKA_DYN_SMEM(type, name) type *name = reinterpret_cast<type *>(hipemu::dyn_smem())
// lanes of one wave exchange data through LDS without a workgroup barrier: on the GPU the wave executes in lockstep and
// LDS operations retire in program order; the emulator runs lanes as fibers and needs a rendezvous
__device__ __forceinline__ void wave_sync() { (void)hipemu::shfl_any(0, 0); }
// a value the caller guarantees to be equal in all active lanes of the wave (moves it to a scalar register on the GPU)
__device__ __forceinline__ int wave_uniform(int x) { return x; }
// the value lane `l` holds (l wave-uniform; every lane of the wave must call it): v_readlane_b32 on the GPU
__device__ __forceinline__ int lane_bcast(int x, int l) { return hipemu::shfl_any(x, l); }
__device__ __forceinline__ float lane_bcast(float x, int l) { return hipemu::shfl_any(x, l); }
// the value of lane ^ 1 (every lane of the wave must call it)
__device__ __forceinline__ unsigned lane_xor1(unsigned x) { return hipemu::shfl_any(x, (hipemu::F().lin & 63) ^ 1); }
__device__ __forceinline__ int rint_i32(float x) { return (int)lrintf(x); }
__device__ __forceinline__ unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
__device__ __forceinline__ float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
// sum over the 8 lanes that share lane >> 3 (result in all 8)
__device__ __forceinline__ float sum8(float x)
{
    x += hipemu::shfl_any(x, (hipemu::F().lin & 63) ^ 1);
    x += hipemu::shfl_any(x, (hipemu::F().lin & 63) ^ 2);
    x += hipemu::shfl_any(x, (hipemu::F().lin & 63) ^ 4);
    return x;
}
__device__ __forceinline__ void sched_fence() {}
// sum over the 4 lanes that share lane >> 2 (result in all 4)
__device__ __forceinline__ float sum4(float x)
{
    x += hipemu::shfl_any(x, (hipemu::F().lin & 63) ^ 1);
    x += hipemu::shfl_any(x, (hipemu::F().lin & 63) ^ 2);
    return x;
}
// the value of lane - 1 / lane + 1 inside each row of 16 lanes; 0 in the first / last lane of a row (every lane of the wave must call them)
__device__ __forceinline__ float row_shr1(float x)
{
    const int l = hipemu::F().lin & 63;
    const float v = hipemu::shfl_any(x, (l & 15) ? l - 1 : l);
    return (l & 15) ? v : 0.f;
}
__device__ __forceinline__ float row_shl1(float x)
{
    const int l = hipemu::F().lin & 63;
    const float v = hipemu::shfl_any(x, (l & 15) != 15 ? l + 1 : l);
    return (l & 15) != 15 ? v : 0.f;
}
// buffer resource: loads at a byte offset >= the buffer size return 0 (the hardware range check of buffer_load)
struct BufRsrc { const unsigned char *base; unsigned bytes; };
__device__ __forceinline__ BufRsrc make_rsrc(const void *p, size_t bytes) { return BufRsrc{(const unsigned char *)p, (unsigned)bytes}; }
__device__ __forceinline__ float buf_load_f32(BufRsrc r, unsigned off)
{
    float v = 0.f;
    if (off < r.bytes && off + 4 <= r.bytes) memcpy(&v, r.base + off, 4);
    return v;
}
__device__ __forceinline__ f32x4 buf_load_f32x4(BufRsrc r, unsigned off)
{
    f32x4 v;
    for (int e = 0; e < 4; ++e) v[e] = buf_load_f32(r, off + 4u * e);
    return v;
}
// voff: per-lane byte offset (range-checked: DLKA_OOB -> 0); soff: wave-uniform byte offset added on top (NOT range-checked by the
// hardware, so voff + soff must stay inside the buffer by construction)
__device__ __forceinline__ float buf_load_f32_s(BufRsrc r, unsigned voff, unsigned soff)
{
    if (!(voff < r.bytes)) return 0.f;
    return buf_load_f32(r, voff + soff);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 buf_load_f32x2_s(BufRsrc r, unsigned voff, unsigned soff)
{
    f32x2 v = {0.f, 0.f};
    if (!(voff < r.bytes)) return v;
    v[0] = buf_load_f32(r, voff + soff); v[1] = buf_load_f32(r, voff + soff + 4u);
    return v;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { f32x2 d; d[0] = fmaf(a[0], b[0], c[0]); d[1] = fmaf(a[1], b[1], c[1]); return d; }
// 4-byte store, dropped when the offset is out of range (DLKA_OOB)
__device__ __forceinline__ void buf_store_f32(BufRsrc r, unsigned off, float v)
{
    if (!(off < r.bytes) || (size_t)off + 4 > r.bytes) return;
    memcpy(const_cast<unsigned char *>(r.base) + off, &v, 4);
}
// 16-byte store, dropped when the offset is out of range (DLKA_OOB)
__device__ __forceinline__ void buf_store_f32x4(BufRsrc r, unsigned off, f32x4 v)
{
    if (!(off < r.bytes) || (size_t)off + 16 > r.bytes) return;
    memcpy(const_cast<unsigned char *>(r.base) + off, &v, 16);
}
// 8 floats -> 8 bf16 (round to nearest even), one 16-byte store
__device__ __forceinline__ void buf_store_bf16x8(BufRsrc r, unsigned off, f32x4 lo, f32x4 hi)
{
    if (!(off < r.bytes) || (size_t)off + 16 > r.bytes) return;
    unsigned short h[8];
    for (int e = 0; e < 4; ++e) { h[e] = hipemu::hipemu_f32_to_bf16(lo[e]); h[4 + e] = hipemu::hipemu_f32_to_bf16(hi[e]); }
    memcpy(const_cast<unsigned char *>(r.base) + off, h, 16);
}
// fp16 (IEEE half) <-> float: the stored-sample hand-over of the fp32 path keeps its samples as halves (cl_deform_bwd2.hip / cl_wgrad.hip, round 6); values beyond the
// half range saturate at +-65504 instead of becoming infinite
__device__ __forceinline__ unsigned short f16_bits(float x)
{
    const float c = fminf(fmaxf(x, -65504.f), 65504.f);
    return __builtin_bit_cast(unsigned short, (_Float16)c);
}
__device__ __forceinline__ float f16_value(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
// 4 floats -> 4 halves (round to nearest even, saturating), one 8-byte store, dropped when the offset is out of range (DLKA_OOB)
__device__ __forceinline__ void buf_store_f16x4(BufRsrc r, unsigned off, f32x4 v)
{
    if (!(off < r.bytes) || (size_t)off + 8 > r.bytes) return;
    unsigned short h[4];
    for (int e = 0; e < 4; ++e) h[e] = f16_bits(v[e]);
    memcpy(const_cast<unsigned char *>(r.base) + off, h, 8);
}
// ---- bf16 activation storage (DLKA_BF16 token path): 4 consecutive bf16 -> 4 floats, one 8-byte buffer load ----
__device__ __forceinline__ f32x4 buf_load_bf16x4(BufRsrc r, unsigned off)
{
    f32x4 v;
    for (int e = 0; e < 4; ++e) {
        unsigned short h = 0;
        if (off < r.bytes && off + 2u * e + 2 <= r.bytes) memcpy(&h, r.base + off + 2u * e, 2);
        v[e] = hipemu::hipemu_bf16_to_f32(h);
    }
    return v;
}
__device__ __forceinline__ float buf_load_bf16(BufRsrc r, unsigned off)
{
    unsigned short h = 0;
    if (off < r.bytes && off + 2 <= r.bytes) memcpy(&h, r.base + off, 2);
    return hipemu::hipemu_bf16_to_f32(h);
}
__device__ __forceinline__ float buf_load_bf16_s(BufRsrc r, unsigned voff, unsigned soff)
{
    if (!(voff < r.bytes)) return 0.f;
    return buf_load_bf16(r, voff + soff);
}
// 8 consecutive bf16 (one 16-byte load) -> lo = elements 0..3, hi = 4..7
__device__ __forceinline__ void buf_load_bf16x8(BufRsrc r, unsigned off, f32x4 &lo, f32x4 &hi)
{
    lo = buf_load_bf16x4(r, off);
    hi = buf_load_bf16x4(r, off == DLKA_OOB ? DLKA_OOB : off + 8u);
}
// two-term bf16 split packed in one word: hi << 16 | lo
__device__ __forceinline__ float pack_split2(float x)
{
    const unsigned short hi = hipemu::hipemu_f32_to_bf16(x), lo = hipemu::hipemu_f32_to_bf16(x - hipemu::hipemu_bf16_to_f32(hi));
    const unsigned u = ((unsigned)hi << 16) | lo;
    float f; memcpy(&f, &u, 4); return f;
}
__device__ __forceinline__ void unpack_split2x8(const float *w, bf16x8 &hi, bf16x8 &lo)
{
    for (int e = 0; e < 8; ++e) { unsigned u; memcpy(&u, w + e, 4); hi.v[e] = (unsigned short)(u >> 16); lo.v[e] = (unsigned short)(u & 0xffffu); }
}
#else
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// v_mfma_f32_32x32x2_f32 : exact fp32 (k-ordered fmaf chain), 64 cycles/SIMD.  A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
// D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// v_mfma_f32_16x16x4_f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D: col = l&15, row = (l>>4)*4 + r.
__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int readfirstlane(int v) { return __builtin_amdgcn_readfirstlane(v); }
// v_mfma_f32_32x32x16_bf16 (32 cycles/SIMD, 16x the fp32-input rate): lane (i = l&31, g = l>>5) holds A[i][8g..8g+7] and
// B[8g..8g+7][j = l&31]; D as above.  bf16 x bf16 products are exact in fp32 and the accumulation is fp32, so a two-term
// split  a = a_hi + a_lo  (a_lo = bf16(a - a_hi)) with the three products hi*hi + hi*lo + lo*hi reproduces an fp32 product
// to ~3 * 2^-18 relative (1.1e-5 worst case) at 3/16 of the fp32-input MFMA cost.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
// v_mfma_f32_16x16x32_bf16 (gfx950): lane (i = l&15, g = l>>4) holds A[i][8g..8g+7] and B[8g..8g+7][j = l&15]; D: col = l&15, row = 4 (l>>4) + r — one
// instruction contracts a whole 32-channel chunk of a 16 x 16 tile.
__device__ __forceinline__ f32x4 mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void split_bf16x8(const float *a, bf16x8 &hi, bf16x8 &lo)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hi[e] = (__bf16)a[e];
        lo[e] = (__bf16)(a[e] - (float)hi[e]);
    }
}
__device__ __forceinline__ void split3_bf16x8(const float *a, bf16x8 &hi, bf16x8 &mid, bf16x8 &lo)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hi[e] = (__bf16)a[e];
        const float r1 = a[e] - (float)hi[e];   // exact in fp32
        mid[e] = (__bf16)r1;
        lo[e] = (__bf16)(r1 - (float)mid[e]);
    }
}
__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
// 8 bf16 values that are already packed in four 32-bit words (element 0 = low half of word 0): no conversion
__device__ __forceinline__ bf16x8 bf16x8_from_words(const float *w4)
{
    const f32x4 t = {w4[0], w4[1], w4[2], w4[3]};
    return __builtin_bit_cast(bf16x8, t);
}
__device__ __forceinline__ float bf16_value(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
// Buffer loads (buffer_load_dword / _dwordx4 ... offen): 32-bit byte offsets against a wave-uniform descriptor, and the
// hardware range check returns 0 for offsets >= the buffer size.  The gather kernels use that for zero padding and for
// corners outside the volume: the offset of a dropped element is DLKA_OOB, so every load is unconditional — no branch
// around it, no select after it, and the loads of the next tile stay in flight under the MFMAs of the current one
// (a conditional load compiles to s_cbranch + s_waitcnt vmcnt(0) per element; measured in profiles/archive/r01g).
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }
// a value the caller guarantees to be equal in all active lanes of the wave: into a scalar register, so that pointers / buffer
// descriptors derived from it are built by the scalar unit
__device__ __forceinline__ int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
// the value lane `l` holds, as a scalar (l must be wave-uniform: v_readlane_b32)
__device__ __forceinline__ int lane_bcast(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ float lane_bcast(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
// the value of lane ^ 1: one DPP move (quad_perm [1, 0, 3, 2])
__device__ __forceinline__ unsigned lane_xor1(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false); }
__device__ __forceinline__ int rint_i32(float x) { return __float2int_rn(x); }
// sum over the 8 lanes that share lane >> 3 (result in all 8): three DPP adds, no LDS traffic
// quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141 (lane k <-> 7-k inside each 8)
__device__ __forceinline__ float sum8(float x)
{
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false));
    return x;
}
// nothing moves across it in the instruction scheduler (keeps register live ranges short where the scheduler would otherwise interleave two
// independent register-hungry sections)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// sum over the 4 lanes that share lane >> 2 (result in all 4): the two quad_perm steps of sum8
__device__ __forceinline__ float sum4(float x)
{
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));
    return x;
}
// the value of lane - 1 / lane + 1 inside each row of 16 lanes, 0 in the first / last lane of a row: one DPP move (row_shr:1 = 0x111, row_shl:1 = 0x101,
// bound_ctrl: lanes without a source read 0)
__device__ __forceinline__ float row_shr1(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xF, 0xF, true)); }
__device__ __forceinline__ float row_shl1(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x101, 0xF, 0xF, true)); }
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc make_rsrc(const void *p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(BufRsrc r, unsigned off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); }
__device__ __forceinline__ f32x4 buf_load_f32x4(BufRsrc r, unsigned off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); }
// voff: per-lane byte offset (range-checked: DLKA_OOB -> 0); soff: wave-uniform byte offset (SGPR) added on top — not part of the hardware
// range check, so voff + soff must stay inside the buffer by construction.  Saves the per-load VALU address add.
__device__ __forceinline__ float buf_load_f32_s(BufRsrc r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 buf_load_f32x2_s(BufRsrc r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
// v_pk_fma_f32: two fp32 FMAs per lane in one issue slot (each half rounds like fmaf)
#if defined(DLKA_NO_PK)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { f32x2 d; d[0] = fmaf(a[0], b[0], c[0]); d[1] = fmaf(a[1], b[1], c[1]); return d; }
#else
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#endif
// No wave-uniform (SGPR) offset variant for stores, on purpose.  Measured on the MI355X (scripts/debug_samp.py, round 2): with an SGPR soffset the
// compiler (ROCm 7.2 clang) assumes the "VALU overwrites the data registers of a > 64-bit VMEM store" hazard away and schedules such a VALU
// write directly behind buffer_store_dwordx4 — lanes 8-15 / 24-31 / 40-47 / 56-63 then stored the NEW register contents.  With the whole offset
// in the VGPR (soffset = literal 0) the hazard recogniser inserts the wait state.
__device__ __forceinline__ void buf_store_f32(BufRsrc r, unsigned off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0); }
__device__ __forceinline__ void buf_store_f32x4(BufRsrc r, unsigned off, f32x4 v)
{
    typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0)) raw128_t;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(raw128_t, v), r, off, 0, 0);
}
// 8 floats -> 8 bf16 (round to nearest even), one 16-byte store
__device__ __forceinline__ void buf_store_bf16x8(BufRsrc r, unsigned off, f32x4 lo, f32x4 hi)
{
    typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0)) raw128_t;
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    u4_t w;
    w[0] = (unsigned)bf16_bits(lo[0]) | ((unsigned)bf16_bits(lo[1]) << 16); w[1] = (unsigned)bf16_bits(lo[2]) | ((unsigned)bf16_bits(lo[3]) << 16);
    w[2] = (unsigned)bf16_bits(hi[0]) | ((unsigned)bf16_bits(hi[1]) << 16); w[3] = (unsigned)bf16_bits(hi[2]) | ((unsigned)bf16_bits(hi[3]) << 16);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(raw128_t, w), r, off, 0, 0);
}
// fp16 (IEEE half) <-> float: the stored-sample hand-over of the fp32 path keeps its samples as halves (cl_deform_bwd2.hip / cl_wgrad.hip, round 6); values beyond the
// half range saturate at +-65504 instead of becoming infinite
__device__ __forceinline__ unsigned short f16_bits(float x)
{
    const float c = fminf(fmaxf(x, -65504.f), 65504.f);
    return __builtin_bit_cast(unsigned short, (_Float16)c);
}
__device__ __forceinline__ float f16_value(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
// 4 floats -> 4 halves (round to nearest even, saturating), one 8-byte store (whole offset in the VGPR: see the note above)
__device__ __forceinline__ void buf_store_f16x4(BufRsrc r, unsigned off, f32x4 v)
{
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    typedef decltype(__builtin_amdgcn_raw_buffer_load_b64(r, 0, 0, 0)) raw64_t;
    u2_t w;
    w[0] = (unsigned)f16_bits(v[0]) | ((unsigned)f16_bits(v[1]) << 16);
    w[1] = (unsigned)f16_bits(v[2]) | ((unsigned)f16_bits(v[3]) << 16);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(raw64_t, w), r, off, 0, 0);
}
// ---- bf16 activation storage (DLKA_BF16 token path): 4 consecutive bf16 -> 4 floats, one 8-byte buffer load ----
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 buf_load_bf16x4(BufRsrc r, unsigned off)
{
    const u32x2_t w = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    f32x4 v;
    v[0] = __builtin_bit_cast(float, w[0] << 16); v[1] = __builtin_bit_cast(float, w[0] & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, w[1] << 16); v[3] = __builtin_bit_cast(float, w[1] & 0xffff0000u);
    return v;
}
__device__ __forceinline__ float buf_load_bf16(BufRsrc r, unsigned off)
{
    return __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0) << 16);
}
__device__ __forceinline__ float buf_load_bf16_s(BufRsrc r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0) << 16);
}
// 8 consecutive bf16 (one 16-byte load) -> lo = elements 0..3, hi = 4..7
__device__ __forceinline__ void buf_load_bf16x8(BufRsrc r, unsigned off, f32x4 &lo, f32x4 &hi)
{
    const u32x4_t w = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    lo[0] = __builtin_bit_cast(float, w[0] << 16); lo[1] = __builtin_bit_cast(float, w[0] & 0xffff0000u);
    lo[2] = __builtin_bit_cast(float, w[1] << 16); lo[3] = __builtin_bit_cast(float, w[1] & 0xffff0000u);
    hi[0] = __builtin_bit_cast(float, w[2] << 16); hi[1] = __builtin_bit_cast(float, w[2] & 0xffff0000u);
    hi[2] = __builtin_bit_cast(float, w[3] << 16); hi[3] = __builtin_bit_cast(float, w[3] & 0xffff0000u);
}
// two-term bf16 split packed in one word (hi << 16 | lo): the producer splits ONCE, consumers that contract the value many times (27 taps)
// rebuild their MFMA operands with one v_perm per pair instead of ~5 VALU instructions per value per use
__device__ __forceinline__ float pack_split2(float x)
{
    const __bf16 hi = (__bf16)x;
    const __bf16 lo = (__bf16)(x - (float)hi);
    return __builtin_bit_cast(float, ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16) | (unsigned)__builtin_bit_cast(unsigned short, lo));
}
__device__ __forceinline__ void unpack_split2x8(const float *w, bf16x8 &hi, bf16x8 &lo)
{
    u32x4_t h4, l4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned w0 = __builtin_bit_cast(unsigned, w[2 * q]), w1 = __builtin_bit_cast(unsigned, w[2 * q + 1]);
        h4[q] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);   // {w1.hi16, w0.hi16}
        l4[q] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);   // {w1.lo16, w0.lo16}
    }
    hi = __builtin_bit_cast(bf16x8, h4);
    lo = __builtin_bit_cast(bf16x8, l4);
}
#define DLKA_DYN_SMEM(type, name)                                             \
    extern __shared__ __attribute__((aligned(16))) unsigned char dlka_dyn_lds[]; \
    type *name = reinterpret_cast<type *>(dlka_dyn_lds)
#endif
}  // namespace dlka
