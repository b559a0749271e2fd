// Thin names for the gfx950 intrinsics the kernels use.  The HIPEMU branch exists only so that
// tests/emu can compile the kernel sources with a host compiler (test infrastructure, see
// tests/emu/include/hip/hip_runtime.h); hipcc never defines HIPEMU.
#pragma once
#include <hip/hip_runtime.h>

namespace dlka {
#if defined(HIPEMU)
typedef hipemu_f32x4 f32x4;
typedef hipemu_f32x16 f32x16;
__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) { return hipemu::mfma_f32_32x32x2f32(a, b, c); }
__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) { return hipemu::mfma_f32_16x16x4f32(a, b, c); }
__device__ __forceinline__ int readfirstlane(int v) { return hipemu::readfirstlane(v); }
#define DLKA_DYN_SMEM(type, name) type *name = reinterpret_cast<type *>(hipemu::dyn_smem())
#else
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// v_mfma_f32_32x32x2_f32 : exact fp32 (k-ordered fmaf chain), 64 cycles/SIMD.  A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
// D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// v_mfma_f32_16x16x4_f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D: col = l&15, row = (l>>4)*4 + r.
__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int readfirstlane(int v) { return __builtin_amdgcn_readfirstlane(v); }
#define DLKA_DYN_SMEM(type, name)                                             \
    extern __shared__ __attribute__((aligned(16))) unsigned char dlka_dyn_lds[]; \
    type *name = reinterpret_cast<type *>(dlka_dyn_lds)
#endif
}  // namespace dlka
