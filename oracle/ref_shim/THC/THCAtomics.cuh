/* TEST INFRASTRUCTURE ONLY -- see ../cuda_runtime.h.  The reference includes THC/THCAtomics.cuh
 * (deform_im2col_cuda.cuh:9) for atomicAdd on float / double, which HIP provides natively; nvcc also pre-includes
 * cuda_runtime.h (the .cuh uses cudaStream_t without including it), so this shim pulls the runtime shim in. */
#pragma once
#include "../cuda_runtime.h"
