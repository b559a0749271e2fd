// Internal launcher interface between dlka_capi.hip (validation + sequencing) and the kernel files.
#pragma once
#include "dlka_common.h"

namespace dlka {

// ---- deform_conv.hip (general gather+contract path) -------------------------------------------------------
int deform_pick_cob(int Og);
int deform_fwd_wt_floats(const Geom &g);  // floats needed for the re-laid-out weight
template <typename T>
int launch_relayout_weight(const T *w, float *wt, int group, int Og, int Cg, int K, int OgP, hipStream_t st);
template <typename T, int NOFF>
int launch_deform_fwd(const T *x, const T *off, const T *w, const T *bias, T *out, float *wt, const Geom &g, hipStream_t st);
template <typename T, int NOFF>
int launch_deform_bwd_input_offset(const T *x, const T *off, const float *wt, int OgP, const T *gout,
                                   float *gx32, T *goff, const Geom &g, hipStream_t st);
template <typename T, int NOFF>
int launch_deform_bwd_weight(const T *x, const T *off, const T *gout, float *gw32, const Geom &g, hipStream_t st);
template <typename T>
int launch_bias_grad(const T *gout, T *gb, int B, int Cout, int No, hipStream_t st);
template <typename T>
int launch_cast_from_f32(const float *src, T *dst, long n, hipStream_t st);
template <typename T>
int launch_sample_index(const T *off, int32_t *idx, uint8_t *mask, const Geom &g, hipStream_t st);

// ---- conv.hip (general grouped convolution: depthwise, dense, pointwise) -----------------------------------
int conv_fwd_wt_floats(const Geom &g);
int conv_bwd_wb_floats(const Geom &g);
template <typename T>
int launch_conv_fwd(const T *x, const T *w, const T *bias, T *out, float *wt, const Geom &g, hipStream_t st);
template <typename T>
int launch_conv_bwd_data(const T *gout, const T *w, T *gx, float *wb, const Geom &g, hipStream_t st);
template <typename T>
int launch_conv_bwd_weight(const T *x, const T *gout, float *gw32, const Geom &g, hipStream_t st);

// ---- eltwise.hip -------------------------------------------------------------------------------------------
template <typename T> int launch_gelu_fwd(const T *x, T *y, long n, hipStream_t st);
template <typename T> int launch_gelu_bwd(const T *x, const T *gy, T *gx, long n, hipStream_t st);
template <typename T> int launch_mul_fwd(const T *a, const T *b, T *y, long n, hipStream_t st);
template <typename T> int launch_mul_bwd(const T *a, const T *b, const T *gy, T *ga, T *gb, long n, hipStream_t st);
template <typename T> int launch_add_fwd(const T *a, const T *b, T *y, long n, hipStream_t st);

}  // namespace dlka
