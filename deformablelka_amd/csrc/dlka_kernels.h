// Internal launcher interface between dlka_capi.hip (validation + sequencing) and the kernel files.
#pragma once
#include "cl_args.h"
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
int launch_sample_index(const T *off, int32_t *idx, uint8_t *mask, const Geom &g, int path, hipStream_t st);
template <typename T>
int launch_sample_index2(const T *off, int32_t *idx, uint8_t *mask, const Geom &g, int path, hipStream_t st);

// ---- deform_conv_f64.hip: DLKA_F64, the general NCDHW path in double (AT_DISPATCH_FLOATING_TYPES' second type, deform_conv_cuda.cu:96,233) -----------------------
template <int NOFF> int launch_deform_fwd_f64(const double *x, const double *off, const double *w, const double *bias, double *out, const Geom &g, hipStream_t st);
template <int NOFF> int launch_deform_bwd_f64(const double *x, const double *off, const double *w, const double *gout, double *gx, double *goff, double *gw, double *gb,
                                              const Geom &g, hipStream_t st);
int launch_conv_fwd_f64(const double *x, const double *w, const double *bias, double *out, const Geom &g, hipStream_t st);
int launch_conv_bwd_f64(const double *x, const double *w, const double *gout, double *gx, double *gw, double *gb, const Geom &g, hipStream_t st);

// ---- conv.hip (general grouped convolution: depthwise, dense, pointwise) -----------------------------------
int conv_fwd_wt_floats(const Geom &g);
int conv_bwd_wb_floats(const Geom &g);
template <typename T>
int launch_conv_fwd(const T *x, const T *w, const T *bias, T *out, float *wt, const Geom &g, hipStream_t st);
template <typename T>
int launch_conv_bwd_data(const T *gout, const T *w, T *gx, float *wb, const Geom &g, hipStream_t st);
// `part` (conv_bwd_weight_part_floats(g, sizeof(T)) floats; may be null): per-workgroup tiles of the MFMA weight gradient, folded by a second launch
// instead of one atomic per element and workgroup
size_t conv_bwd_weight_part_floats(const Geom &g, size_t elem_bytes);
template <typename T>
int launch_conv_bwd_weight(const T *x, const T *gout, float *gw32, const Geom &g, hipStream_t st, float *part = nullptr);

// ---- eltwise.hip -------------------------------------------------------------------------------------------
int launch_zero(void *ptr, size_t bytes, hipStream_t st);
struct ZeroBatch;
int launch_zero_batch(ZeroBatch &b, hipStream_t st);   // zero fill by kernel (graph-replay safe)
template <typename T> int launch_gelu_fwd(const T *x, T *y, long n, hipStream_t st);
template <typename T> int launch_gelu_bwd(const T *x, const T *gy, T *gx, long n, hipStream_t st);
template <typename T> int launch_mul_fwd(const T *a, const T *b, T *y, long n, hipStream_t st);
template <typename T> int launch_mul_bwd(const T *a, const T *b, const T *gy, T *ga, T *gb, long n, hipStream_t st);
template <typename T> int launch_add_fwd(const T *a, const T *b, T *y, long n, hipStream_t st);
template <typename T> int launch_gelu_bwd_sum(const T *x, const T *g1, const T *g2, T *gx, long n, hipStream_t st);

// ---- channels-last fast path (cl_*.hip) --------------------------------------------------------------------------
int launch_cl_prep_weight(const float *w, float *wp, int Cout, int Cin, int K, int KP, int NP, int mode, hipStream_t st);
int launch_cl_prep_batch(const PrepBatch &b, hipStream_t st);
int cl_prep_table_blocks(const PrepJob &j);   // workgroups job j takes in a prep table / batch launch
int launch_cl_prep_table(const PrepJob *jobs_dev, const int *first_dev, int job_lo, int job_hi, int nblocks, hipStream_t st);
int cl_igemm_pick_splits(int M, int units, int epi, int K);
int launch_cl_pointwise(const IgemmArgs &a, hipStream_t st);
int launch_cl_pointwise_pair(const PwPairArgs &a, hipStream_t st);   // two dependent pointwise convs in one launch (C = 32 / 64)
int launch_cl_conv_wave(int amode, int omode, const IgemmArgs &a, int splits, hipStream_t st);
bool cl_conv_kw_applies(int amode, int omode, int split_bf16, int K, int epi, int NP, bool act_bf16, bool volume);   // cl_conv_kw.hip: K split over the waves of a workgroup
int launch_cl_conv_kw(int amode, int omode, const IgemmArgs &a, hipStream_t st);   // deterministic small-volume contraction (no tap split, no atomics, no zero fill)
bool cl_conv_brick_supported(const IgemmArgs &a);
int cl_conv_brick_split(const IgemmArgs &a);
bool cl_conv_brick3_supported(const IgemmArgs &a);
int launch_cl_conv_brick3(const IgemmArgs &a, hipStream_t st);   // the forward offset conv (three-term) from an LDS brick
int launch_cl_conv_brick(const IgemmArgs &a, hipStream_t st);   // cl_conv_brick.hip: planar-input 3^3 data gradient from an LDS brick
int launch_cl_igemm(int amode, int omode, IgemmArgs a, int splits, hipStream_t st);
int launch_cl_deform_fwd(IgemmArgs a, int splits, hipStream_t st);
int cl_deform_fwd_actual_splits(int K, int CinP, int splits);   // the slab count launch_cl_deform_fwd really uses for a requested tap split
int launch_cl_slab_reduce(const float *slab, int S, long n, void *out, int out_bf16, hipStream_t st);   // out = sum of the S slabs in slab order (deterministic)
int cl_wgrad_pick_chunks(int M, int K, int Cout, int Cin, int amode);
size_t cl_wgrad_part_floats(int M, int K, int Cout, int Cin);
size_t cl_wgrad_pad_bytes(int B, int D, int H, int W, int Cin, int kd, int kh, int kw, int dd, int dh, int dw, int act_bf16);   // the zero-padded input copy of the padded dense kernels (WgradArgs::pad)
template <typename T> int launch_cl_wgrad(int amode, int gmode, WgradArgs a, T *gw, T *gb, hipStream_t st, FinalizeJob *defer = nullptr);
size_t cl_wgrad_part_floats_mode(int M, int K, int Cout, int Cin, int amode);
int launch_cl_wgrad_finalize(FinalizeBatch &b, hipStream_t st);
// planar_ops.hip: NCDHW plumbing of the full net (BatchNorm3d in training mode, 1x1x1 convs on few channels)
int launch_pl_bn_forward(const float *x, const float *w, const float *b, float *stats, float *y, float *scratch, int B, int C, long N, float eps, hipStream_t st);
int launch_pl_bn_backward(const float *g, const float *x, const float *w, const float *stats, float *gx, float *gw, float *gb, float *scratch, int B, int C,
                          long N, hipStream_t st);
int launch_pl_pw_forward(const float *x, const float *w, const float *bias, float *y, int B, int CI, int CO, long N, hipStream_t st);
int launch_pl_pw_backward(const float *x, const float *w, const float *g, float *gx, float *gw, float *gb, int B, int CI, int CO, long N, hipStream_t st);
long cl_wgrad_finalize_plan_job(FinalizeJob &j);   // workgroups the job needs (sets its fold variant)
int launch_cl_wgrad_finalize_table(const FinalizeJob *jobs_device, int job_lo, int job_hi, long nblocks, hipStream_t st);
int launch_cl_wgrad_pw3(const WgradArgs *jobs, float *const *gw, float *const *gb, hipStream_t st, FinalizeJob *defer);
int launch_cl_colsum(const float *g, float *gb32, int M, int Cout, hipStream_t st);
int launch_cl_dw_prep_weight(const float *w, float *wp, int C, int K, int flip, hipStream_t st);
int launch_cl_dwconv(const DwArgs &a, int kw, int dil_w, hipStream_t st);
bool cl_dwpair_small_supported(const DwPairArgs &a);   // cl_dwpair.hip: two chained depthwise convs of a small volume (N <= 512) in one launch
int launch_cl_dwpair_small(const DwPairArgs &a, hipStream_t st);   // DLKA_ERR_UNSUPPORTED = run the convs one by one
int launch_cl_dwconv_lds(const DwArgs &a, int kw, int dil_w, hipStream_t st);
bool cl_dwconv_lds_selected(const DwArgs &a, int kw, int dil_w);   // would launch_cl_dwconv take the LDS-brick kernel?
int cl_dwconv_lds_mode();   // DLKA_DW_LDS (0 = never: no blocked copies are sized or carved)
size_t cl_dwconv_blk_floats(int B, int C, int D, int H, int W, int dil);   // floats of the class-blocked copy (DwArgs::blk)   // cl_dwconv_lds.hip: DLKA_ERR_UNSUPPORTED = keep the register-row kernels
int launch_cl_dwconv_wgrad(DwWgradArgs a, int kw, int dil_w, hipStream_t st, bool zero_init = true);
template <typename T> int launch_cl_dw_unprep(const float *gwp, T *gw, int C, int K, hipStream_t st);
int launch_cl_transpose(const float *src, float *dst, int B, int C, int N, int to_cl, hipStream_t st, int bf16 = 0);
size_t cl_deform_bwd2_scratch_floats(const DeformBwdArgs &a);
int launch_cl_deform_bwd2(const DeformBwdArgs &a, float *scratch, hipStream_t st);
int cl_deform_goff_ccsplit(const DeformBwdArgs &a);

// ---- cl_ddw2d.hip: channels-last 2-D depthwise deformable conv (the 2-D D-LKA block's large-kernel convs) ------------------------------
int cl_ddw2d_supported(int C);
size_t cl_ddw2d_part_floats(int M, int K, int C);
int launch_cl_ddw2d_fwd(const DwArgs2d &d, hipStream_t st);
int launch_cl_ddw2d_bwd(const DwArgs2d &d, float *gw, hipStream_t st, hipStream_t gx_st = nullptr);
// the 2-D D-LKA block on the channels-last kernels (dlka_capi_cl.hip); NCHW in / out, transposed inside
int lka2d_cl_supported(int B, int C, int H, int W, int dtype);
size_t lka2d_cl_saved_bytes(int B, int C, int H, int W, int dtype);
int lka2d_cl_saved_offsets(int B, int C, int H, int W, int dtype, size_t byte_offsets[2], int *elem_bytes);
size_t lka2d_cl_workspace_bytes(int B, int C, int H, int W, int dtype);
int lka2d_cl_forward(const void *x, const dlka_lka2d_params *p, void *y, void *saved, size_t saved_bytes, void *workspace, size_t workspace_bytes, int B,
                     int C, int H, int W, int dtype, hipStream_t st);
int lka2d_cl_backward(const void *x, const dlka_lka2d_params *p, const void *gy, const void *saved, size_t saved_bytes, void *gx, const dlka_lka2d_grads *gr,
                      void *workspace, size_t workspace_bytes, int B, int C, int H, int W, int dtype, hipStream_t st);

// ---- cl_norm.hip: the non-convolutional pieces of TransformerBlock_3D_single_deform_LKA ---------------------------------
int launch_cl_layernorm_fwd(const float *x, int x_planar, const float *pos, const float *w, const float *b, float *xt, float *xn, float *stats, int B, int N,
                            int C, float eps, hipStream_t st, int lo = 0, float *xn32 = nullptr);   // lo: the tensor that crosses into the D-LKA block (xn / g / e / ge) is bf16 storage; xn32: + its unrounded fp32 twin
int launch_cl_layernorm_bwd(const float *g, const float *g_res, const float *xt, const float *stats, const float *w, float *gxt, float *gw, float *gb,
                            float *gpos, int B, int N, int C, hipStream_t st, bool zeroed = false, int lo = 0);
int launch_cl_scale_residual_fwd(const float *xt, const float *e, const float *gamma, float *out, long M, int C, hipStream_t st, int lo = 0);
int launch_cl_scale_residual_bwd(const float *g, const float *e, const float *gamma, float *ge, float *ggamma, long M, int C, hipStream_t st, bool zeroed = false, int lo = 0);
int launch_cl_bn_stats(const float *x, float *sums, float *stats, long M, int C, float eps, hipStream_t st, bool zeroed = false, float *det_part = nullptr, size_t det_floats = 0);
int launch_cl_bn_apply(const float *x, const float *res, const float *w, const float *b, const float *stats, const float *mask, float *y, long M, long N, int C,
                       float slope, hipStream_t st);
int launch_cl_bn_bwd(const float *g, const float *gmask, const float *x, const float *y, const float *w, const float *stats, float *sums, float *gx, float *gres,
                     const float *gres_add, float *gw, float *gb, long M, long N, int C, float slope, int training, hipStream_t st, bool zeroed = false);
int launch_cl_channel_scale(const float *x, const float *mask, float *y, int B, long N, int C, hipStream_t st);

}  // namespace dlka
