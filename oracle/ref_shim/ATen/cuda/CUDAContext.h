/* TEST INFRASTRUCTURE ONLY — see ../../cuda_runtime.h.  The reference asks ATen for the current CUDA stream
 * (deform_conv_cuda.cu:97,234,244,254); on a ROCm build of torch that is the current HIP stream. */
#pragma once
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
namespace at { namespace cuda {
inline c10::hip::HIPStreamMasqueradingAsCUDA getCurrentCUDAStream() {
  return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA();
}
}}

/* torch <= 1.x let AT_DISPATCH_FLOATING_TYPES take `tensor.type()` (a DeprecatedTypeProperties), which is how the
 * reference calls it (deform_conv_cuda.cu:96,233); torch 2.10 only takes a ScalarType.  Same dispatch, old spelling. */
namespace dlka_ref_shim {
inline at::ScalarType scalar_type_of(at::ScalarType t) { return t; }
inline at::ScalarType scalar_type_of(const at::DeprecatedTypeProperties& t) { return t.scalarType(); }
}
#undef AT_DISPATCH_FLOATING_TYPES
#define AT_DISPATCH_FLOATING_TYPES(TYPE, NAME, ...) \
  AT_DISPATCH_SWITCH(::dlka_ref_shim::scalar_type_of(TYPE), NAME, AT_DISPATCH_CASE_FLOATING_TYPES(__VA_ARGS__))
