/* TEST INFRASTRUCTURE ONLY (oracle/_ref).  Lets the reference's CUDA sources
 * (3D/dcn/src/cuda/deform_conv_cuda.cu, deform_im2col_cuda.cuh) be compiled UNMODIFIED by hipcc for gfx950, from
 * where they lie under /root/reference, so that the reference's own arithmetic can pin the oracle and the HIP kernels.
 * Never included or linked by the product (deformablelka_amd/). */
#pragma once
#include <hip/hip_runtime.h>
typedef hipStream_t cudaStream_t;
typedef hipError_t cudaError_t;
#define cudaSuccess hipSuccess
#define cudaGetLastError hipGetLastError
#define cudaGetErrorString hipGetErrorString
