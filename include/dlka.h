/*
 * dlka.h — C-ABI of libdlka_hip.so: the MI355X (gfx950) implementation of the Deformable
 * Large-Kernel-Attention hot path of xmindflow/deformableLKA.
 *
 * This is the drop-in boundary.  Every entry point takes plain device pointers, a geometry
 * struct of ints and an opaque hipStream_t; no torch / ATen type crosses it.  Each declaration
 * cites the reference interface it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - All tensor pointers are DEVICE pointers to dense, contiguous buffers in the layouts stated
 *     per function (the reference asserts contiguity: 3D/dcn/src/cuda/deform_conv_cuda.cu:41-42).
 *   - Inputs are borrowed and never written.  Outputs and workspaces are caller-allocated
 *     (the reference allocates outputs itself with at::empty/zeros_like, deform_conv_cuda.cu:82,
 *     202-205; here the host-language wrapper owns allocation so that the library stays ATen-free).
 *   - Every call is asynchronous on `stream` (the reference uses the current CUDA stream,
 *     deform_conv_cuda.cu:97) and is legal inside hipGraph capture: no allocation, no sync.
 *   - Return value: DLKA_OK (0) or a negative dlka_status.  Nothing throws across this boundary.
 *     dlka_status_string() gives the message the Python wrapper turns into RuntimeError, matching
 *     the reference's AT_ASSERTM -> c10::Error -> RuntimeError behaviour.
 *   - dtype: DLKA_F32 everywhere (the reference dispatches float/double only,
 *     deform_conv_cuda.cu:96,233).  DLKA_BF16 = bf16 storage with fp32 accumulation (new capability).
 */
#ifndef DLKA_H_
#define DLKA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLKA_ABI_VERSION 1

typedef enum dlka_status {
    DLKA_OK = 0,
    DLKA_ERR_NULL = -1,         /* a required pointer is NULL                                          */
    DLKA_ERR_GROUP = -2,        /* channels / out_channels not divisible by group (cu:65-66)           */
    DLKA_ERR_DEFORM_GROUP = -3, /* channels not divisible by deformable_group                          */
    DLKA_ERR_SHAPE = -4,        /* non-positive size or output size (cu:78-80)                         */
    DLKA_ERR_IM2COL_STEP = -5,  /* batch % min(batch, im2col_step) != 0 (cu:61-63)                      */
    DLKA_ERR_DTYPE = -6,        /* unsupported dtype                                                    */
    DLKA_ERR_WORKSPACE = -7,    /* workspace too small / NULL                                           */
    DLKA_ERR_UNSUPPORTED = -8,  /* parameter combination not implemented                                */
    DLKA_ERR_LAUNCH = -9        /* hipGetLastError() != hipSuccess after a launch (reference only printf's, cuh:425-429) */
} dlka_status;

/* DLKA_F64: the reference's op dispatches float AND double (AT_DISPATCH_FLOATING_TYPES, 3D/dcn/src/cuda/deform_conv_cuda.cu:96,233; its scripts import gradcheck,
 * 3D/dcn/test.py:9).  Accepted by the GENERAL NCDHW entry points — dlka_deform_conv3d_*, dlka_deform_conv2d_*, dlka_conv3d_* (forward, backward; every tensor double) —
 * and nowhere else: the fused blocks and the channels-last fast paths return DLKA_ERR_DTYPE / DLKA_ERR_UNSUPPORTED for it. */
typedef enum dlka_dtype { DLKA_F32 = 0, DLKA_BF16 = 1, DLKA_F64 = 2 } dlka_dtype;

/* Geometry of one (deformable or plain) N-d convolution.  2-D ops use D = kd = sd = dd = 1, pd = 0. */
typedef struct dlka_conv_geom {
    int32_t B, C, D, H, W;     /* input  [B][C][D][H][W]                                  */
    int32_t Cout;              /* weight [Cout][C/group][kd][kh][kw]                      */
    int32_t kd, kh, kw;
    int32_t sd, sh, sw;        /* stride                                                  */
    int32_t pd, ph, pw;        /* zero padding                                            */
    int32_t dd, dh, dw;        /* dilation                                                */
    int32_t group;             /* weight groups                                           */
    int32_t deformable_group;  /* offset groups (deformable ops only; else ignored)       */
    int32_t im2col_step;       /* accepted and validated for API parity only (cu:59-63);
                                  no column buffer exists here, so it does not change the result */
} dlka_conv_geom;

int         dlka_abi_version(void);
const char *dlka_status_string(int status);
/* Output spatial size  (in + 2p - (d(k-1)+1))/s + 1   (deform_conv_cuda.cu:78-80). */
int         dlka_conv_out_size(int in, int pad, int dil, int k, int stride);

/* =======================================================================================
 * 3-D deformable convolution — the D3D operator
 * ======================================================================================= */

/* Replaces  D3D.deform_conv_forward  (3D/dcn/src/vision.cpp:5, 3D/dcn/src/deform_conv.h:10-47)
 *           = deform_conv_cuda_forward (3D/dcn/src/cuda/deform_conv_cuda.cu:18-126)
 *           + deformable_im2col_gpu_kernel (3D/dcn/src/cuda/deform_im2col_cuda.cuh:192-265).
 *   x      [B][C][D][H][W]
 *   offset [B][dg*3*K][Do][Ho][Wo]   channel = dg_idx*3K + 3*tap + {0:d,1:h,2:w}, tap=(i*kh+j)*kw+k (cuh:237-239)
 *   weight [Cout][C/group][kd][kh][kw]
 *   bias   [Cout]                    always added, also when the module was built with bias=False
 *   out    [B][Cout][Do][Ho][Wo]
 *   workspace: dlka_deform_conv3d_forward_workspace(g, dtype) bytes (re-laid-out weights; no im2col columns). */
size_t dlka_deform_conv3d_forward_workspace(const dlka_conv_geom *g, int dtype);
int dlka_deform_conv3d_forward(const void *x, const void *offset, const void *weight, const void *bias,
                               void *out, void *workspace, size_t workspace_bytes,
                               const dlka_conv_geom *g, int dtype, void *stream);

/* Replaces  D3D.deform_conv_backward (3D/dcn/src/vision.cpp:6, deform_conv.h:49-91)
 *           = deform_conv_cuda_backward (deform_conv_cuda.cu:128-285)
 *           + deformable_col2im_coord_gpu_kernel (cuh:336-405)  -> grad_offset
 *           + deformable_col2im_gpu_kernel       (cuh:267-334)  -> grad_x   (fp32 atomics, order non-deterministic as in the reference)
 *           + deformable_im2col_gpu_kernel again + addmm/addmv (cu:254-278) -> grad_weight, grad_bias.
 *   All four gradients are fully overwritten (the reference returns fresh zeros_like tensors, cu:202-205).
 *   Any of the grad_* pointers may be NULL to skip that gradient. */
size_t dlka_deform_conv3d_backward_workspace(const dlka_conv_geom *g, int dtype);
int dlka_deform_conv3d_backward(const void *x, const void *offset, const void *weight, const void *grad_out,
                                void *grad_x, void *grad_offset, void *grad_weight, void *grad_bias,
                                void *workspace, size_t workspace_bytes,
                                const dlka_conv_geom *g, int dtype, void *stream);

/* Debug/parity entry point: floor() sampling indices and the in-range guard of cuh:247 for every
 * (b, dg, tap, out voxel).  idx int32 [B][dg][K][No][3], mask uint8 [B][dg][K][No].
 * Backs the "integer sampling indices bit-exact" requirement of BASELINE.json. */
int dlka_deform_conv3d_sample_index(const void *offset, int32_t *idx, uint8_t *mask,
                                    const dlka_conv_geom *g, int dtype, void *stream);
/* Same outputs, computed by the very device functions the hot kernels call, so that the bit-exact index claim covers
 * the code that runs.  ONE device function forms q, tests the guard and takes the floor (sample_cell3, deform_sample.h);
 * path 0 = that function on its own (cl_deform_gx_fx2_kernel calls it directly); path 1 = setup_tap<3> (every general NCDHW
 * kernel); path 2 = gather_describe3 (cl_gather.h — the channels-last forward / grad_offset / weight-gradient kernels);
 * path 3 = lane_tap (the grad_input window kernels).  idx = 0 where mask == 0 (outside the guard of cuh:247 no cell is formed). */
int dlka_deform_conv3d_sample_index_path(const void *offset, int32_t *idx, uint8_t *mask,
                                         const dlka_conv_geom *g, int dtype, int path, void *stream);

/* =======================================================================================
 * 2-D deformable convolution — torchvision.ops.deform_conv2d(mask=None) semantics
 * ======================================================================================= */

/* Replaces  torchvision.ops.DeformConv2d.forward / deform_conv2d as called at
 *           2D/deformable_LKA/deformable_LKA.py:18-30 (torchvision==0.12.0, 2D/requirements.txt:69).
 *   x [B][C][H][W]; offset [B][og*2*K][Ho][Wo] with (dy,dx) per tap; weight [Cout][C/group][kh][kw];
 *   bias [Cout] or NULL (the reference passes bias=False, deformable_LKA.py:25); out [B][Cout][Ho][Wo].
 *   Geometry: D=kd=sd=dd=1, pd=0; deformable_group = offset groups. */
size_t dlka_deform_conv2d_forward_workspace(const dlka_conv_geom *g, int dtype);
int dlka_deform_conv2d_forward(const void *x, const void *offset, const void *weight, const void *bias,
                               void *out, void *workspace, size_t workspace_bytes,
                               const dlka_conv_geom *g, int dtype, void *stream);
size_t dlka_deform_conv2d_backward_workspace(const dlka_conv_geom *g, int dtype);
int dlka_deform_conv2d_backward(const void *x, const void *offset, const void *weight, const void *grad_out,
                                void *grad_x, void *grad_offset, void *grad_weight, void *grad_bias,
                                void *workspace, size_t workspace_bytes,
                                const dlka_conv_geom *g, int dtype, void *stream);

/* =======================================================================================
 * Plain grouped N-d convolution — the nn.Conv3d / nn.Conv2d calls that sit on the path
 * ======================================================================================= */

/* Replaces the cuDNN-backed nn.Conv3d / nn.Conv2d modules inside the D-LKA block:
 *   depthwise 5^3 pad 2 and 7^3 dil 3 pad 9   (3D/d_lka_former/network_architecture/synapse/transformerblock.py:637-638)
 *   offset-predict conv C->81, 3^3 pad 1      (3D/d_lka_former/network_architecture/synapse/deform_conv.py:80-85)
 *   1x1x1 convs proj_1 / conv1 / proj_2       (transformerblock.py:641,659,662)
 *   2-D offset nets C->50 (5x5) / C->98 (7x7 dil 3) (2D/deformable_LKA/deformable_LKA.py:10-16)
 *   x [B][C][D][H][W]; weight [Cout][C/group][kd][kh][kw]; bias [Cout] or NULL; out [B][Cout][Do][Ho][Wo]. */
size_t dlka_conv3d_forward_workspace(const dlka_conv_geom *g, int dtype);
int dlka_conv3d_forward(const void *x, const void *weight, const void *bias, void *out,
                        void *workspace, size_t workspace_bytes,
                        const dlka_conv_geom *g, int dtype, void *stream);
/* grad_x / grad_weight / grad_bias may each be NULL to skip. All are fully overwritten.
 * Arithmetic (DLKA_F32): fp32 products and sums — except the shape the assembled net's full-resolution plumbing uses (3^3, stride 1, padding 1, group 1, <= 16 -> <= 16
 * channels, W % 8 == 0): its grad_x (W <= 128, >= 8 output channels) and grad_weight contract on the bf16 matrix cores with BOTH operands as two bf16 terms and fp32
 * accumulation (~1e-5 of max|gradient|, inside the 1e-3 gradient contract; the rule of the token-layout block's gradients, see dlka_lka3d_attention_tokens_backward).
 * DLKA_EXACT_FP32=1 keeps fp32-input MFMAs there too. */
size_t dlka_conv3d_backward_workspace(const dlka_conv_geom *g, int dtype);
int dlka_conv3d_backward(const void *x, const void *weight, const void *grad_out,
                         void *grad_x, void *grad_weight, void *grad_bias,
                         void *workspace, size_t workspace_bytes,
                         const dlka_conv_geom *g, int dtype, void *stream);

/* =======================================================================================
 * Elementwise pieces of the block (GELU, gate)
 * ======================================================================================= */

/* y = GELU_erf(x)  (nn.GELU() default, transformerblock.py:660);  n elements. */
int dlka_gelu_forward(const void *x, void *y, int64_t n, int dtype, void *stream);
/* gx = gy * dGELU(x) */
int dlka_gelu_backward(const void *x, const void *gy, void *gx, int64_t n, int dtype, void *stream);
/* y = a * b   (the "u * attn" gate, transformerblock.py:652; deformable_LKA.py:104) */
int dlka_mul_forward(const void *a, const void *b, void *y, int64_t n, int dtype, void *stream);
/* ga = gy * b ; gb = gy * a */
int dlka_mul_backward(const void *a, const void *b, const void *gy, void *ga, void *gb, int64_t n, int dtype, void *stream);
/* y = a + b */
int dlka_add_forward(const void *a, const void *b, void *y, int64_t n, int dtype, void *stream);

/* =======================================================================================
 * Whole D-LKA attention blocks (fused launch sequences; one C call per block)
 * ======================================================================================= */

/* Parameters of LKA_Attention3d_deform (transformerblock.py:655-673) in state_dict order. All [..] dense. */
typedef struct dlka_lka3d_params {
    const void *proj_1_w, *proj_1_b;             /* [C][C][1][1][1], [C]              proj_1                         */
    const void *conv0_w, *conv0_b;               /* [C][1][5][5][5], [C]              spatial_gating_unit.conv0       */
    const void *conv_spatial_w, *conv_spatial_b; /* [C][1][7][7][7], [C]              spatial_gating_unit.conv_spatial */
    const void *offset_w, *offset_b;             /* [81][C][3][3][3], [81]            ...deform_conv.conv_offset      */
    const void *deform_w, *deform_b;             /* [C][C][3][3][3], [C]              ...deform_conv.{weight,bias}    */
    const void *conv1_w, *conv1_b;               /* [C][C][1][1][1], [C]              spatial_gating_unit.conv1       */
    const void *proj_2_w, *proj_2_b;             /* [C][C][1][1][1], [C]              proj_2                         */
} dlka_lka3d_params;

typedef struct dlka_lka3d_grads { /* same shapes as dlka_lka3d_params; all fully overwritten */
    void *proj_1_w, *proj_1_b, *conv0_w, *conv0_b, *conv_spatial_w, *conv_spatial_b, *offset_w, *offset_b,
         *deform_w, *deform_b, *conv1_w, *conv1_b, *proj_2_w, *proj_2_b;
} dlka_lka3d_grads;

/* Replaces  LKA_Attention3d_deform.forward(x, B, C, H, W, D)  (transformerblock.py:664-673) including
 *           LKA3d_deform.forward (:644-652) and DeformConvPack.forward (synapse/deform_conv.py:93-105).
 *   x, y: NCDHW volumes [B][C][D][H][W] (the (B,N,C)<->NCDHW permutes of :665,672 stay in the caller:
 *   they are views/copies on the Python side exactly as in the reference).
 *   saved: activations the backward needs; dlka_lka3d_saved_bytes(B,C,D,H,W,dtype) bytes, caller-owned,
 *          must stay untouched until the matching backward call.
 *   workspace: scratch, dlka_lka3d_workspace_bytes(...) bytes, may be reused right after the call. */
size_t dlka_lka3d_saved_bytes(int B, int C, int D, int H, int W, int dtype);
size_t dlka_lka3d_workspace_bytes(int B, int C, int D, int H, int W, int dtype);
int dlka_lka3d_attention_forward(const void *x, const dlka_lka3d_params *p, void *y,
                                 void *saved, size_t saved_bytes, void *workspace, size_t workspace_bytes,
                                 int B, int C, int D, int H, int W, int dtype, void *stream);
/* Backward of the above: grad_y -> grad_x and all 14 parameter gradients. */
int dlka_lka3d_attention_backward(const void *x, const dlka_lka3d_params *p, const void *grad_y,
                                  const void *saved, size_t saved_bytes,
                                  void *grad_x, const dlka_lka3d_grads *grads,
                                  void *workspace, size_t workspace_bytes,
                                  int B, int C, int D, int H, int W, int dtype, void *stream);

/* Parameters of the 2-D deformable_LKA_Attention (2D/deformable_LKA/deformable_LKA.py:124-140). */
typedef struct dlka_lka2d_params {
    const void *proj_1_w, *proj_1_b;             /* [C][C][1][1], [C]                                             */
    const void *conv0_offset_w, *conv0_offset_b; /* [50][C][5][5], [50]   conv0.offset_net                         */
    const void *conv0_w;                         /* [C][1][5][5]          conv0.deform_conv.weight (bias=False)    */
    const void *conv_spatial_offset_w, *conv_spatial_offset_b; /* [98][C][7][7], [98]  conv_spatial.offset_net      */
    const void *conv_spatial_w;                  /* [C][1][7][7]          conv_spatial.deform_conv.weight          */
    const void *conv1_w, *conv1_b;               /* [C][C][1][1], [C]                                             */
    const void *proj_2_w, *proj_2_b;             /* [C][C][1][1], [C]                                             */
} dlka_lka2d_params;

typedef struct dlka_lka2d_grads {
    void *proj_1_w, *proj_1_b, *conv0_offset_w, *conv0_offset_b, *conv0_w,
         *conv_spatial_offset_w, *conv_spatial_offset_b, *conv_spatial_w, *conv1_w, *conv1_b, *proj_2_w, *proj_2_b;
} dlka_lka2d_grads;

/* Replaces  deformable_LKA_Attention.forward (deformable_LKA.py:133-140) incl. deformable_LKA.forward (:98-104)
 *           and DeformConv.forward (:27-30).  x, y: [B][C][H][W]. */
/* dtype DLKA_BF16 (BASELINE.json config 2, "bf16 training"): x / y / grad_y / grad_x and the saved activations are bf16 storage, the
 * PARAMETERS and their gradients stay fp32 masters, offsets and accumulation are fp32, and the chain that decides the sampling cells
 * (a -> offset net 5 -> DDW5 -> offset net 7) is kept in fp32.  Channels-last fast path only: C / 32 in {1, 2, 3, 4, 6, 8, 12};
 * other widths return DLKA_ERR_UNSUPPORTED for DLKA_BF16. */
size_t dlka_lka2d_saved_bytes(int B, int C, int H, int W, int dtype);
size_t dlka_lka2d_workspace_bytes(int B, int C, int H, int W, int dtype);
int dlka_lka2d_attention_forward(const void *x, const dlka_lka2d_params *p, void *y,
                                 void *saved, size_t saved_bytes, void *workspace, size_t workspace_bytes,
                                 int B, int C, int H, int W, int dtype, void *stream);
/* Forces (1) / releases (0) the general NCHW kernels for blocks the channels-last fast path would take; returns the previous setting.  Initial
 * value: 1 iff the environment variable DLKA_LKA2D_GENERAL is set when the first 2-D block call is made.  For A/B runs and the parity test of
 * one path against the other; the size queries return the maximum over both paths, so a switch never under-sizes a buffer. */
int dlka_lka2d_force_general(int on);
/* Diagnostics (bench health check, the parity tests' cell-flip analysis): where inside the opaque `saved` buffer of a forward call the two
 * predicted offset tensors live — conv0's [B][50][H][W] and conv_spatial's [B][98][H][W], planar as torchvision lays them out — for the path a
 * call with these arguments takes NOW (fast / general, see dlka_lka2d_force_general).  byte_offsets[0..1]: byte offsets from `saved`;
 * *elem_bytes: 4 (fp32; always on the channels-last path) or 2 (general path with bf16 tensors). */
int dlka_lka2d_saved_offsets(int B, int C, int H, int W, int dtype, size_t byte_offsets[2], int *elem_bytes);
int dlka_lka2d_attention_backward(const void *x, const dlka_lka2d_params *p, const void *grad_y,
                                  const void *saved, size_t saved_bytes,
                                  void *grad_x, const dlka_lka2d_grads *grads,
                                  void *workspace, size_t workspace_bytes,
                                  int B, int C, int H, int W, int dtype, void *stream);


/* =======================================================================================
 * Channels-last (NDHWC / token layout) fast path — fp32, stride 1, same-size output
 * ======================================================================================= *
 * The D-LKA block's real interface is the token tensor (B, N, C) (transformerblock.py:664-673): the reference permutes
 * it to NCDHW (a copy), runs the block, and permutes back (another copy).  On MI355X the whole block runs directly in
 * token = channels-last layout: dense contractions (1x1x1 projections, offset-predict conv, deformable conv and their
 * gradients) are implicit GEMMs on the fp32-input matrix cores, depthwise convs are register-tiled vector kernels.
 * Offsets keep the reference's planar layout [B][3K][N].  Functions return DLKA_ERR_UNSUPPORTED for shapes outside
 * the fast path (caller then uses the general NCDHW entry points above).                                              */

/* x [B][D][H][W][C]; weight/bias in the reference layout; out [B][D][H][W][Cout], or planar [B][Cout][N] if out_planar.
 * Depthwise (group == C == Cout, kw/dilation in {5/1, 7/3, 3/1, 5/3, 7/1}) or dense (group == 1, C % 32 == 0). */
size_t dlka_conv3d_cl_workspace(const dlka_conv_geom *g, int dtype, int backward);
int dlka_conv3d_forward_cl(const void *x, const void *weight, const void *bias, void *out, int out_planar,
                           void *workspace, size_t workspace_bytes, const dlka_conv_geom *g, int dtype, void *stream);
int dlka_conv3d_backward_cl(const void *x, const void *weight, const void *grad_out, int grad_out_planar,
                            void *grad_x, void *grad_weight, void *grad_bias,
                            void *workspace, size_t workspace_bytes, const dlka_conv_geom *g, int dtype, void *stream);

/* Deformable conv with x / out / grad_x channels-last and offset / grad_offset planar (reference layout).
 * group == deformable_group == 1, C and Cout in {32, 64, 96, 128, 256}.  Same semantics as dlka_deform_conv3d_*. */
size_t dlka_deform_conv3d_cl_workspace(const dlka_conv_geom *g, int dtype, int backward);
int dlka_deform_conv3d_forward_cl(const void *x, const void *offset, const void *weight, const void *bias, void *out,
                                  void *workspace, size_t workspace_bytes, const dlka_conv_geom *g, int dtype, void *stream);
int dlka_deform_conv3d_backward_cl(const void *x, const void *offset, const void *weight, const void *grad_out,
                                   void *grad_x, void *grad_offset, void *grad_weight, void *grad_bias,
                                   void *workspace, size_t workspace_bytes, const dlka_conv_geom *g, int dtype, void *stream);

/* [B][C][N] <-> [B][N][C] */
int dlka_ncdhw_to_ndhwc(const void *src, void *dst, int B, int C, int N, int dtype, void *stream);
int dlka_ndhwc_to_ncdhw(const void *src, void *dst, int B, int C, int N, int dtype, void *stream);

/* Replaces  LKA_Attention3d_deform.forward(x, B, C, H, W, D)  (transformerblock.py:664-673) on the TOKEN tensor itself:
 *   x, y, grad_x, grad_y: [B][N][C] with N = D*H*W voxels in (d,h,w) order of the reference's reshape(B,C,H,W,D)
 *   (its "H,W,D" are just the three spatial extents).  No permute/copy on either side.
 *   Supported: C in {32, 64, 128, 256} (the four D_LKA_Former stage widths); dtype
 *     DLKA_F32   everything fp32 storage and fp32 accumulation (the parity path: 1e-4 forward, 1e-3 gradients against the reference).  ARITHMETIC of the contractions:
 *                the FORWARD deformable conv and the pointwise convs run on the fp32-input MFMA (exact fp32 products); the forward offset-predict conv as a THREE-term
 *                bf16 split (six products per fp32 product: fp32-equivalent — its output decides floor()); the BACKWARD contractions of the deformable conv (Col of
 *                grad_offset / grad_input), its WEIGHT gradient (fp32 grad_out x the samples grad_offset handed over as IEEE halves — a half is exactly two bf16 terms)
 *                and the offset conv's data / weight gradients as TWO-term bf16 splits (three products, fp32 accumulation, ~1e-5 relative: inside the 1e-3 gradient
 *                contract).  DLKA_EXACT_FP32=1 (environment, read once) puts every contraction on the fp32-input MFMA (and keeps the samples fp32);
 *     DLKA_BF16  x, y, grad_y, grad_x and every saved / intermediate activation are bf16 STORAGE; parameters (dlka_lka3d_params), their
 *                gradients, the predicted offsets, grad_offset and all accumulation are fp32.  The offset-predict conv runs single bf16
 *                MFMA products against two-term (fp32-exact) weights.  The reference registers no autocast policy and would raise on half
 *                inputs (deform_conv_cuda.cu:96); this is the policy the Python modules apply inside torch.autocast(dtype=bfloat16). */
int    dlka_lka3d_tokens_supported(int B, int C, int D, int H, int W, int dtype);
size_t dlka_lka3d_tokens_saved_bytes(int B, int C, int D, int H, int W, int dtype);
size_t dlka_lka3d_tokens_workspace_bytes(int B, int C, int D, int H, int W, int dtype);
int dlka_lka3d_attention_tokens_forward(const void *x, const dlka_lka3d_params *p, void *y,
                                        void *saved, size_t saved_bytes, void *workspace, size_t workspace_bytes,
                                        int B, int C, int D, int H, int W, int dtype, void *stream);
/* Weight preparation of MANY blocks in ONE launch.  The forward call above re-lays the block's weights into MFMA operand order (12 small
 * jobs, one launch per block; inside a hipGraph every dependent node costs ~4.5 us, and at C = 256 the launch itself is 41 us).  The prepared
 * forms depend on the parameters only, so a model that steps all its blocks (the 21 blocks of a D_LKA_Former patch) prepares them together,
 * once per optimizer step:
 *   plan_bytes  size of the job table;  _prepare_plan fills a HOST buffer (pointers of every block's parameters and `saved` area; they must stay
 *   put); the caller copies it to the device once;  _prepare_run(plan_device, plan_host, n) = one launch;  _forward_prepared = the forward call
 *   without its preparation launch.  params: array of n structs; dims5: n x (B, C, D, H, W). */
size_t dlka_lka3d_tokens_prepare_plan_bytes(int nblocks);
int dlka_lka3d_tokens_prepare_plan(int nblocks, const dlka_lka3d_params *params, void *const *saved, const size_t *saved_bytes,
                                   const int *dims5, int dtype, void *plan_host, size_t plan_bytes);
int dlka_lka3d_tokens_prepare_run(const void *plan_device, const void *plan_host, int nblocks, void *stream);
/* the same for blocks [block_lo, block_hi) only: a caller can prepare the first blocks, start their forward passes, and prepare the rest on another stream */
int dlka_lka3d_tokens_prepare_run_range(const void *plan_device, const void *plan_host, int nblocks, int block_lo, int block_hi, void *stream);
int dlka_lka3d_attention_tokens_forward_prepared(const void *x, const dlka_lka3d_params *p, void *y,
                                                 void *saved, size_t saved_bytes, void *workspace, size_t workspace_bytes,
                                                 int B, int C, int D, int H, int W, int dtype, void *stream);
int dlka_lka3d_attention_tokens_backward(const void *x, const dlka_lka3d_params *p, const void *grad_y,
                                         const void *saved, size_t saved_bytes,
                                         void *grad_x, const dlka_lka3d_grads *grads,
                                         void *workspace, size_t workspace_bytes,
                                         int B, int C, int D, int H, int W, int dtype, void *stream);
/* Weight-gradient finalisation of MANY blocks in ONE launch.  The backward call above ends the block's seven weight gradients with one
 * "finalize" launch (fold of the row-chunk partial sums, re-layout of the depthwise staging): 21 dependent launches of 15 - 30 us per step for the
 * blocks of a D_LKA_Former patch, most of each latency.  Nothing later in the backward pass reads their results, so a model that steps all its
 * blocks lets the partial sums of every block land in a block-PRIVATE area and folds them all at the end of the pass (or of a slice of it):
 *   _partials_bytes_v        size of a block's private partial-sum area;
 *   _plan_bytes / _plan_init a HOST job table for n blocks;
 *   _backward_deferred_v     the backward call without its finalize launch; partial sums go to `partials`; with plan_host != NULL it records the
 *                            block's jobs in slot `plan_slot` (pointers of `partials` and of the gradient buffers: they must stay put);
 *   _run_slot                (before sealing) finalises ONE recorded block with the ordinary per-block launch — what a caller does while the table is
 *                            still being recorded, e.g. when its first backward pass covers only a slice of the blocks;
 *   _plan_seal               after every slot has been recorded once: computes the launch geometry; the caller then copies the table to the device;
 *   _run(plan_device, plan_host, block_lo, block_hi)   ONE launch that finalises the weight gradients of blocks [block_lo, block_hi).
 * Results are identical to the per-block launch (same folds in the same order). */
size_t dlka_lka3d_tokens_partials_bytes_v(int B, int C, int D, int H, int W, int dtype, int variant);
size_t dlka_wgrad_finalize_plan_bytes(int nblocks);
int dlka_wgrad_finalize_plan_init(void *plan_host, size_t plan_bytes, int nblocks);
int dlka_lka3d_attention_tokens_backward_deferred_v(const void *x, const dlka_lka3d_params *p, const void *grad_y,
                                                    const void *saved, size_t saved_bytes,
                                                    void *grad_x, const dlka_lka3d_grads *grads,
                                                    void *workspace, size_t workspace_bytes,
                                                    void *partials, size_t partials_bytes, void *plan_host, int plan_slot,
                                                    int B, int C, int D, int H, int W, int dtype, int variant, void *stream);
/* The deferred backward pass in two calls: phase 1 = the data-gradient chain (produces grad_x and, in `workspace`, everything the weight gradients
 * read), phase 2 = the five weight-gradient launches (reads them; records the block's finalize jobs).  Phase 2 may run on ANOTHER stream, after an
 * event recorded behind phase 1, so that it overlaps the next block's data chain — the caller then alternates two workspaces and makes phase 1 of a
 * block wait for the phase 2 that last used its workspace.  Same results as the one-call form. */
int dlka_lka3d_attention_tokens_backward_phase_v(const void *x, const dlka_lka3d_params *p, const void *grad_y,
                                                 const void *saved, size_t saved_bytes,
                                                 void *grad_x, const dlka_lka3d_grads *grads,
                                                 void *workspace, size_t workspace_bytes,
                                                 void *partials, size_t partials_bytes, void *plan_host, int plan_slot, int phase,
                                                 int B, int C, int D, int H, int W, int dtype, int variant, void *stream);
int dlka_wgrad_finalize_run_slot(const void *plan_host, int plan_slot, void *stream);
int dlka_wgrad_finalize_plan_seal(void *plan_host);
int dlka_wgrad_finalize_run(const void *plan_device, const void *plan_host, int block_lo, int block_hi, void *stream);

/* =======================================================================================
 * TransformerBlock_3D_single_deform_LKA: what surrounds the D-LKA block
 * ======================================================================================= *
 * The reference wrapper (3D/d_lka_former/network_architecture/synapse/transformerblock.py:617-630) reshapes the NCDHW
 * volume to tokens (a permute copy), adds pos_embed, applies nn.LayerNorm, computes x + gamma * epa_block(...), permutes
 * back (another copy) and runs UnetResBlock (two 3^3 convs, BatchNorm3d, LeakyReLU 0.01; dynunet_block.py:12-80) and
 * Dropout3d + 1x1x1 conv.  Here everything stays in token / channels-last layout [M = B*N][C], fp32, C <= 256; the
 * convolutions go through dlka_conv3d_*_cl, the rest through the entry points below.                                  */

/* xt = tokens(x) (+ pos[N][C]);  xn = LayerNorm(xt) * w + b (biased variance, eps inside the sqrt: nn.LayerNorm, :609,:624);
 * stats[m] = {mean, rstd}.  x is the block input either as the NCDHW tensor itself (x_planar = 1: [B][C][N], read strided —
 * replaces the reshape/permute copy of :620) or already as tokens (x_planar = 0). */
int dlka_layernorm_tokens_forward(const void *x, int x_planar, const void *pos, const void *w, const void *b, void *xt, void *xn,
                                  void *stats, int B, int N, int C, float eps, int dtype, void *stream);
/* gxt = (g_res ? g_res : 0) + LayerNorm backward of g_xn;  gw, gb, gpos ([N][C], optional) fully overwritten. */
int dlka_layernorm_tokens_backward(const void *g_xn, const void *g_res, const void *xt, const void *stats, const void *w, void *gxt,
                                   void *gw, void *gb, void *gpos, int B, int N, int C, int dtype, void *stream);
/* out = xt + gamma[c] * e   (:624, gamma = 1e-6 * ones at construction, :610) */
int dlka_scale_residual_forward(const void *xt, const void *e, const void *gamma, void *out, int64_t M, int C, int dtype, void *stream);
/* ge = gamma[c] * g;  ggamma[c] = sum_m g * e */
int dlka_scale_residual_backward(const void *g, const void *e, const void *gamma, void *ge, void *ggamma, int64_t M, int C, int dtype,
                                 void *stream);
/* y = LeakyReLU(BatchNorm(x) (+ res))   (dynunet_block.py:68-79; nn.BatchNorm3d over (B, spatial) per channel).
 * training = 1: batch statistics are computed and written to stats = {mean[C], rstd[C], unbiased var[C]};
 * training = 0: stats[0..2C) = {running_mean, 1/sqrt(running_var + eps)} supplied by the caller.  scratch: 2*C floats. */
int dlka_batchnorm_cl_forward(const void *x, const void *res, const void *w, const void *b, void *stats, int training, void *y,
                              void *scratch, int64_t M, int C, float eps, float slope, int dtype, void *stream);
/* gx, gw, gb fully overwritten; gres (optional) = gradient of the residual input.  scratch: 2*C floats. */
int dlka_batchnorm_cl_backward(const void *g, const void *x, const void *y, const void *w, const void *stats, int training, void *gx,
                               void *gres, void *gw, void *gb, void *scratch, int64_t M, int C, float slope, int dtype, void *stream);
/* y[b][n][c] = x[b][n][c] * mask[b][c]   (nn.Dropout3d drops whole channels per sample, :611; the mask comes from the caller's RNG) */
int dlka_channel_scale(const void *x, const void *mask, void *y, int B, int64_t N, int C, int dtype, void *stream);

/* ---- planar (NCDHW) plumbing of the full D_LKA_Former --------------------------------------------------------------
 * nn.BatchNorm3d in TRAINING mode over fp32 [B][C][N] tensors (the UnetResBlock norms of encoder1 / decoder2 at full resolution,
 * 3D/d_lka_former/network_architecture/dynunet_block.py:66-80): y = (x - mean) rstd w + b with batch statistics;
 * stats = {mean[C], rstd[C], unbiased var[C], mean - x[0][c][0]} — 4*C floats; the caller updates its running estimates from rows 0 and 2;
 * w / b may be NULL; scratch: 2*C floats.
 * backward: gx fully overwritten, gw / gb (optional) = the affine gradients. */
int dlka_batchnorm_planar_forward(const void *x, const void *w, const void *b, void *stats, void *y, void *scratch,
                                  int B, int C, int64_t N, float eps, void *stream);
int dlka_batchnorm_planar_backward(const void *g, const void *x, const void *w, const void *stats, void *gx, void *gw, void *gb,
                                   void *scratch, int B, int C, int64_t N, void *stream);
/* 1x1x1 nn.Conv3d on fp32 planar tensors with few channels (the output heads, d_lka_former_synapse.py:148-150; conv3 of a UnetResBlock whose channel
 * count changes): y[b][co][v] = sum_ci W[co][ci] x[b][ci][v] + bias[co].  N % 4 == 0, Cin in {1, 2, 4, 8, 14, 16, 32}, Cout <= 64 (weight gradient:
 * Cout <= 16); anything else returns DLKA_ERR_UNSUPPORTED.  backward: gx / gw (+ gb) optional, fully overwritten. */
int dlka_pointwise_planar_forward(const void *x, const void *w, const void *bias, void *y, int B, int Cin, int Cout, int64_t N, void *stream);
int dlka_pointwise_planar_backward(const void *x, const void *w, const void *g, void *gx, void *gw, void *gb,
                                   int B, int Cin, int Cout, int64_t N, void *stream);

/* ---- the whole wrapper block, one call per direction --------------------------------------------------------------- */
/* Parameters of TransformerBlock_3D_single_deform_LKA other than epa_block's (those travel as dlka_lka3d_params):
 * state_dict keys norm.*, gamma, pos_embed, conv51.conv{1,2}.conv.weight, conv51.norm{1,2}.*, conv8.1.* (:609-616). */
typedef struct dlka_tblock3d_params {
    const void *norm_w, *norm_b;                         /* [C]             nn.LayerNorm                                   */
    const void *gamma;                                   /* [C]                                                           */
    const void *pos_embed;                               /* [N][C] or NULL  (pos_embed=False)                             */
    const void *conv51_conv1_w, *conv51_conv2_w;         /* [C][C][3][3][3] UnetResBlock convs, no bias                   */
    const void *conv51_norm1_w, *conv51_norm1_b;         /* [C]             BatchNorm3d affine                            */
    const void *conv51_norm2_w, *conv51_norm2_b;         /* [C]                                                           */
    const void *conv8_w, *conv8_b;                       /* [C][C][1][1][1], [C]   conv8[1]                               */
} dlka_tblock3d_params;

typedef struct dlka_tblock3d_grads { /* same shapes; all fully overwritten; pos_embed NULL iff the parameter is */
    void *norm_w, *norm_b, *gamma, *pos_embed, *conv51_conv1_w, *conv51_conv2_w, *conv51_norm1_w, *conv51_norm1_b,
         *conv51_norm2_w, *conv51_norm2_b, *conv8_w, *conv8_b;
} dlka_tblock3d_grads;

int dlka_tblock3d_supported(int B, int C, int D, int H, int W, int dtype);   /* fp32, C in {32,64,128,256} */
size_t dlka_tblock3d_saved_bytes(int B, int C, int D, int H, int W, int dtype);
size_t dlka_tblock3d_workspace_bytes(int B, int C, int D, int H, int W, int dtype);
/* Replaces TransformerBlock_3D_single_deform_LKA.forward (transformerblock.py:617-630).
 *   x: the block input, NCDHW [B][C][N] (x_planar = 1) or already tokens [B][N][C] (x_planar = 0, e.g. the previous block's y);
 *   y: tokens [B][N][C] — the reference's (B, C, H, W, D) result is the permuted VIEW of this memory;
 *   drop_mask: [B][C] multipliers of conv8[0] = Dropout3d(0.1) drawn by the caller's RNG ({0, 1/(1-p)}), NULL = no dropout (eval);
 *   bn_stats: 6*C floats {mean1, rstd1, var1, mean2, rstd2, var2}: written in training mode (var = unbiased batch variance, for the
 *             caller's running-statistics update), read in eval mode (mean = running_mean, rstd = 1/sqrt(running_var + eps));
 *             must be handed unchanged to the matching backward call, like `saved`. */
int dlka_tblock3d_forward(const void *x, int x_planar, const dlka_tblock3d_params *p, const dlka_lka3d_params *lka,
                          const void *drop_mask, int training, void *bn_stats, void *y,
                          void *saved, size_t saved_bytes, void *workspace, size_t workspace_bytes,
                          int B, int C, int D, int H, int W, float ln_eps, float bn_eps, int dtype, void *stream);
/* grad_y, grad_x: tokens [B][N][C] (grad wrt the NCDHW input is the permuted view of grad_x). */
int dlka_tblock3d_backward(const dlka_tblock3d_params *p, const dlka_lka3d_params *lka, const void *drop_mask, int training,
                           const void *bn_stats, const void *grad_y, const void *saved, size_t saved_bytes, void *grad_x,
                           const dlka_tblock3d_grads *grads, const dlka_lka3d_grads *lka_grads,
                           void *workspace, size_t workspace_bytes, int B, int C, int D, int H, int W, int dtype, void *stream);

/* 2-D counterpart (torchvision 0.12 deform_conv2d, un-vendored; 2D/deformable_LKA/deformable_LKA.py:18-30): idx int32 [B][og][K][No][2]
 * = the floor cell (y, x) where `reach`, else 0; mask uint8: bit 0 = the sample is inside the guard (-1 < q < size), bit 1 = reach
 * (q >= -1 && q < size: the domain of torchvision's unguarded coordinate weight).  path 0 = sample_cell2 (deform_sample.h; the
 * window scatter of cl_ddw2d.hip calls it directly), path 1 = setup_tap<2> (general kernels), path 2 = describe2 (cl_ddw2d.hip). */
int dlka_deform_conv2d_sample_index_path(const void *offset, int32_t *idx, uint8_t *mask,
                                         const dlka_conv_geom *g, int dtype, int path, void *stream);

/* Channels-last 2-D DEPTHWISE deformable conv on its own — the two large-kernel convs of the 2-D D-LKA block
 * (2D/deformable_LKA/deformable_LKA.py:93-94 -> :18-30: torchvision.ops.DeformConv2d(C, C, k, padding, groups=C, dilation,
 * bias=False)(x, offsets)).  x / out / grad_out / grad_x [B][H][W][C]; offset / grad_offset [B][2 kh kw][H][W] ((dy, dx) per tap, as
 * torchvision); weight / grad_weight [C][1][kh][kw].  Stride 1, "same" padding, C % 32 == 0, fp32.  All three gradients at once. */
size_t dlka_deform_dwconv2d_cl_workspace(const dlka_conv_geom *g, int dtype, int backward);
int dlka_deform_dwconv2d_forward_cl(const void *x, const void *offset, const void *weight, void *out,
                                    void *workspace, size_t workspace_bytes, const dlka_conv_geom *g, int dtype, void *stream);
int dlka_deform_dwconv2d_backward_cl(const void *x, const void *offset, const void *weight, const void *grad_out,
                                     void *grad_x, void *grad_offset, void *grad_weight,
                                     void *workspace, size_t workspace_bytes, const dlka_conv_geom *g, int dtype, void *stream);

/* =======================================================================================
 * Net variants of the 3-D block (the depthwise pair conv0 / conv_spatial of LKA3d_deform)
 * =======================================================================================
 * DLKA_LKA3D_SYNAPSE  3D/d_lka_former/network_architecture/synapse/transformerblock.py:637-638 (and the pancreas copy): 5^3 pad 2, then
 *                     7^3 dilation 3 pad 9, at every width.  All entry points without the _v suffix.
 * DLKA_LKA3D_ACDC     3D/d_lka_former/network_architecture/acdc/transformerblock.py:213-237: C <= 64: 5^3 pad 2, (5,7,7) dilation 3 pad (6,9,9);
 *                     C = 128: 5^3 pad 2, (3,5,5) dilation (1,3,3) pad (1,6,6); C = 256: 3^3 pad 1, 3^3 pad 1 — the stem (1,4,4) net whose stage
 *                     shapes BASELINE.json config 5's 40x224x224 tiles divide through (acdc/model_components.py:21).  conv0_w / conv_spatial_w
 *                     (and their gradients) have the variant's kernel shapes.
 * The _v entry points are their un-suffixed namesakes with the variant as an extra argument (before `stream`). */
typedef enum dlka_lka3d_variant { DLKA_LKA3D_SYNAPSE = 0, DLKA_LKA3D_ACDC = 1 } dlka_lka3d_variant;
int    dlka_lka3d_tokens_supported_v(int B, int C, int D, int H, int W, int dtype, int variant);
size_t dlka_lka3d_tokens_saved_bytes_v(int B, int C, int D, int H, int W, int dtype, int variant);
/* Forces (1) / releases (0) the deformable weight gradient that GATHERS for itself (no sample tensor handed over by the grad_offset kernel);
 * returns the previous setting.  Initial value: 1 iff the environment variable DLKA_WGRAD_GATHER is set when the first token-path call is made.
 * For A/B runs and the hand-over parity test; the workspace size query always covers both routes. */
int dlka_lka3d_force_wgrad_gather(int on);
/* Fork contexts.  Some backward entry points run independent kernels of ONE call on library-internal streams, forked from and
 * joined back into the caller's stream inside the call (dlka_lka2d_attention_backward; the data-chain pass of the block-stack engine).  The streams and
 * events of a call are a context leased from a per-DEVICE pool for the duration of that call: a call issued with device k current uses handles of device k
 * only (nn.DataParallel replicas, 2D/trainer_MaxViT_deform_LKA.py:107-108), two host threads inside the library at once hold different contexts, contexts
 * are never created inside a stream capture, and an error return joins what was forked.  dlka_fork_stats: contexts created on / leases handed out for
 * `device` so far in this process (tests: no handle of device 0 is touched by a call on device 1; two concurrent callers got two contexts).
 * dlka_env_refresh: the switches DLKA_GX_FORK_MIN_ROWS and DLKA_LKA2D_FORK are read once per process; this re-reads them (tests and A/B scripts that
 * change them in-process call it afterwards). */
int  dlka_fork_stats(int device, int64_t *contexts, int64_t *leases);
void dlka_env_refresh(void);
/* Diagnostics: launches so far (this process) of the opt-in LDS-brick depthwise kernel (csrc/cl_dwconv_lds.hip; DLKA_DW_LDS=1: the dw 5^3 / 7^3
 * dilation-3 convs of the block and their data gradients take it where the volume is large enough, =2: wherever its geometry fits; default: never —
 * it measured no faster than the register-row kernel); the tests use it to assert WHICH kernel produced the result they compare. */
long dlka_dwconv_lds_launch_count(void);
/* launches so far of the LDS-brick data gradient of the offset-predict conv (csrc/cl_conv_brick.hip): parity tests assert which kernel ran */
long dlka_conv_brick_launch_count(void);
/* launches so far of the fused small-volume depthwise pair (csrc/cl_dwpair.hip: dw 5^3 -> dw 7^3 dil 3, or their data gradients + GELU', of a volume of at most
 * 512 voxels with W in {4, 8} in ONE launch; DLKA_DWPAIR=0, read per call, keeps one launch per conv): parity tests assert which kernel ran */
long dlka_dwpair_launch_count(void);
/* Diagnostics: launches of cl_conv_kw_kernel (round 6: the small-volume split-operand convs — offset-predict conv, its data gradient, UnetResBlock's 3^3 convs at the
 * 16^3 / 8^3 / 4^3 stages — with the contraction split over the waves of ONE workgroup and summed in wave order in LDS: bitwise reproducible, no global atomics;
 * DLKA_CONV_KW=0 restores the tap split over the grid) and of the deformable forward's workgroup-split variants.  Tests assert which kernel ran. */
long dlka_conv_kw_launch_count(void);
/* launches so far of the opt-in software-pipelined depthwise kernel (csrc/cl_dwconv.hip: cl_dwconv_rows2p_kernel, round 6; DLKA_DW_2P=1 | 2 sends the dw 5^3 /
 * 7^3 dilation-3 convs and their data gradients through it wherever its geometry fits; default: never — it measured slower than the row kernel) */
long dlka_dwconv_2p_launch_count(void);
size_t dlka_lka3d_tokens_workspace_bytes_v(int B, int C, int D, int H, int W, int dtype, int variant);
int dlka_lka3d_attention_tokens_forward_v(const void *x, const dlka_lka3d_params *p, void *y, void *saved, size_t saved_bytes,
                                          void *workspace, size_t workspace_bytes, int B, int C, int D, int H, int W, int dtype, int variant, void *stream);
int dlka_lka3d_attention_tokens_backward_v(const void *x, const dlka_lka3d_params *p, const void *grad_y, const void *saved, size_t saved_bytes,
                                           void *grad_x, const dlka_lka3d_grads *grads, void *workspace, size_t workspace_bytes,
                                           int B, int C, int D, int H, int W, int dtype, int variant, void *stream);
int    dlka_tblock3d_supported_v(int B, int C, int D, int H, int W, int dtype, int variant);
/* dtype on the wrapper-block entry points: DLKA_F32, or DLKA_BF16 = MIXED precision — x, y, grad_y, grad_x, the residual stream, LayerNorm / BatchNorm
 * statistics, the 3^3 convs of UnetResBlock and all parameter gradients stay fp32 (the pointers are fp32 tensors on both dtypes); the D-LKA attention
 * inside (transformerblock.py:624) runs DLKA_BF16: its input, output and their gradients are bf16 storage, with the token path's rule (the chain that
 * decides the sampling cells fp32, fp32 parameters / offsets / accumulation).  What torch.autocast(bfloat16) selects for the module. */
size_t dlka_tblock3d_saved_bytes_v(int B, int C, int D, int H, int W, int dtype, int variant);
/* Diagnostics (bench health check, the parity tests' cell-flip analysis): byte offset, inside the opaque `saved` buffer of a token-layout
 * forward call / of a wrapper-block forward call, of the predicted sampling offsets [B][81][D][H][W] (fp32 on both dtypes, the reference's planar
 * layout, deform_im2col_cuda.cuh:237-243). */
int dlka_lka3d_tokens_saved_offsets_v(int B, int C, int D, int H, int W, int dtype, int variant, size_t *byte_offset);
int dlka_tblock3d_saved_offsets_v(int B, int C, int D, int H, int W, int dtype, int variant, size_t *byte_offset);
/* Diagnostics (the parity tests' LeakyReLU-kink analysis): byte offsets, inside the `saved` buffer of a wrapper-block forward call, of the two tensors
 * whose SIGN is the activation pattern of UnetResBlock's two LeakyReLUs (dynunet_block.py:69-70,77-79): byte_offsets[0] = a1 = lrelu(norm1(conv1(x))),
 * byte_offsets[1] = rd = dropout3d(lrelu(norm2(conv2(a1)) + x)); both fp32 tokens [B][N][C] on both dtypes. */
int dlka_tblock3d_saved_activations_v(int B, int C, int D, int H, int W, int dtype, int variant, size_t byte_offsets[2]);
size_t dlka_tblock3d_workspace_bytes_v(int B, int C, int D, int H, int W, int dtype, int variant);
int dlka_tblock3d_forward_v(const void *x, int x_planar, const dlka_tblock3d_params *p, const dlka_lka3d_params *lka,
                            const void *drop_mask, int training, void *bn_stats, void *y, void *saved, size_t saved_bytes,
                            void *workspace, size_t workspace_bytes, int B, int C, int D, int H, int W, float ln_eps, float bn_eps,
                            int dtype, int variant, void *stream);
int dlka_tblock3d_backward_v(const dlka_tblock3d_params *p, const dlka_lka3d_params *lka, const void *drop_mask, int training,
                             const void *bn_stats, const void *grad_y, const void *saved, size_t saved_bytes, void *grad_x,
                             const dlka_tblock3d_grads *grads, const dlka_lka3d_grads *lka_grads,
                             void *workspace, size_t workspace_bytes, int B, int C, int D, int H, int W, int dtype, int variant, void *stream);
/* The same pass in two parts (phase 0 = dlka_tblock3d_backward_v).  phase 1: the DATA-gradient chain — grad_x and the gradients its own kernels produce (norm / norm1 /
 * norm2 affine parameters, gamma, pos_embed) —, leaving in `workspace` what phase 2 reads; phase 2: the weight gradients of conv51.conv1 / conv2, conv8 and of the D-LKA
 * attention, and the fold of their partial sums.  Issue phase 2 on another stream behind an event recorded after phase 1 (same arguments; `workspace` untouched in
 * between) and a block's weight gradients overlap the next block's data chain (deformablelka_amd/transformerblock.py: wgrad_overlap). */
int dlka_tblock3d_backward_phase_v(const dlka_tblock3d_params *p, const dlka_lka3d_params *lka, const void *drop_mask, int training, const void *bn_stats,
                                   const void *grad_y, const void *saved, size_t saved_bytes, void *grad_x, const dlka_tblock3d_grads *grads,
                                   const dlka_lka3d_grads *lka_grads, void *workspace, size_t workspace_bytes, int B, int C, int D, int H, int W, int dtype,
                                   int variant, int phase, void *stream);

/* =======================================================================================
 * Launch trace — measurement aid (no reference counterpart; the reference has no profiling hooks)
 * =======================================================================================
 * Between dlka_trace_start and dlka_trace_stop every kernel launch of the library is followed by a HIP timing event on the
 * launch's own stream; record i = (kernel name, milliseconds since the previous record).  bench.py runs the timed step once
 * more under the trace to report the roofline of the kernels the step REALLY launches.  dlka_trace_mark inserts a record
 * without a kernel (name "(mark)") to separate phases.  Not thread-safe; illegal during hipGraph capture.  */
int dlka_trace_start(int max_events, void *stream);
int dlka_trace_mark(void *stream);
int dlka_trace_stop(void);                 /* synchronises with the last record; DLKA_ERR_WORKSPACE if records were dropped */
int dlka_trace_count(void);
int dlka_trace_get(int i, char *name, size_t name_cap, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* DLKA_H_ */
