/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's deformable-convolution algorithm for the D-LKA
 * hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product path (deformablelka_amd/, libdlka_hip.so) never links or calls it.
 *
 * This header is included twice by dlka_oracle.c, once with REAL=float / SFX=_f32 and once with
 * REAL=double / SFX=_f64 (the reference dispatches AT_DISPATCH_FLOATING_TYPES = float,double:
 * 3D/dcn/src/cuda/deform_conv_cuda.cu:96,233).
 *
 * All file:line citations are relative to /root/reference.
 *   [cuh] = 3D/dcn/src/cuda/deform_im2col_cuda.cuh
 *   [cu]  = 3D/dcn/src/cuda/deform_conv_cuda.cu
 *   [tv]  = torchvision==0.12.0 torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp
 *           (un-vendored third-party dependency, pinned at 2D/requirements.txt:69; restated from
 *            its published algorithm, anchored on the call sites 2D/deformable_LKA/deformable_LKA.py:18-30)
 *
 * PARITY PINNING: the reference ships no golden vectors or asserting tests for this path
 * (SURVEY.md §4, §8c), and its own build refuses without CUDA (3D/dcn/setup.py:30-41; the CPU
 * branch throws, src/cpu/deform_cpu.cpp:28,53).  Its SOURCES, however, compile unmodified for
 * gfx950: oracle/ref.mk builds 3D/dcn/src/{vision.cpp, cuda/deform_conv_cuda.cu, cuda/
 * deform_im2col_cuda.cuh, cpu/deform_cpu.cpp} where they lie under /root/reference with hipcc and
 * four shim headers (oracle/ref_shim/) into oracle/_ref/D3D.so — the reference's own pybind11
 * module.  THIS ORACLE IS PINNED TO IT:
 *   - tests/test_ref_d3d_gpu.py, tests/test_ref_d3d_2d_gpu.py (-m gpu) run D3D.so next to this
 *     oracle on the reference's smoke-script shapes, the four stage shapes of the 64x128x128 patch
 *     and the edge cases (3-D; and the 2-D operator through the depth-1 embedding);
 *   - tests/golden/d3d_reference_vectors{,_2d}.pt — inputs and outputs of D3D.so recorded on an
 *     MI355X by tests/golden/make_ref_golden.py — hold this file to the reference's arithmetic in
 *     the CPU suite (tests/test_oracle_vs_reference_vectors.py, tests/test_oracle_2d_pinned.py).
 * The one restated line that cannot be pinned that way — torchvision's UNGUARDED coordinate weight
 * at q == -1 exactly ([tv]; D3D's is guarded, [cuh]:391-394) — is isolated in its own test.
 * Module-level golden vectors come from the reference's own Python classes (tests/golden/
 * make_golden.py, make_golden_nets.py); known-answer properties (zero-offset == conv3d, integer
 * shift, grid_sample cross-check, fp64 gradcheck) are in tests/test_oracle_*.py.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

/* ------------------------------------------------------------------------------------------
 * 3-D sampling helpers
 * ---------------------------------------------------------------------------------------- */

/* Trilinear sample with per-corner zeroing.  Follows dmcn_im2col_bilinear, [cuh]:26-72.
 * Corner order and weight association (hd*hh*hw, left to right) are kept so that the fp32
 * result is the same expression tree as the reference's. */
static REAL FN(trilinear)(const REAL *vol, int D, int H, int W, REAL d, REAL h, REAL w)
{
    const int d0 = (int)floor((double)d), h0 = (int)floor((double)h), w0 = (int)floor((double)w);
    const REAL ld = d - (REAL)d0, lh = h - (REAL)h0, lw = w - (REAL)w0;
    const REAL fd[2] = {(REAL)1 - ld, ld}, fh[2] = {(REAL)1 - lh, lh}, fw[2] = {(REAL)1 - lw, lw};
    REAL acc = 0;
    for (int cd = 0; cd < 2; ++cd)
        for (int ch = 0; ch < 2; ++ch)
            for (int cw = 0; cw < 2; ++cw) {
                const int zd = d0 + cd, zh = h0 + ch, zw = w0 + cw;
                /* low corner needs >= 0, high corner needs <= size-1  ([cuh]:43-66) */
                const int ok = (cd ? zd <= D - 1 : zd >= 0) && (ch ? zh <= H - 1 : zh >= 0) &&
                               (cw ? zw <= W - 1 : zw >= 0);
                const REAL v = ok ? vol[((long)zd * H + zh) * W + zw] : (REAL)0;
                const REAL wt = fd[cd] * fh[ch] * fw[cw]; /* [cuh]:67-68 */
                acc = (cd | ch | cw) ? acc + wt * v : wt * v; /* [cuh]:70 sum in corner order */
            }
    return acc;
}

/* Trilinear weight of integer voxel (pd,ph,pw) for fractional sample (a_d,a_h,a_w).
 * Follows dmcn_get_gradient_weight, [cuh]:74-109. */
static REAL FN(grad_weight_of_voxel)(REAL ad, REAL ah, REAL aw, int pd, int ph, int pw, int D, int H, int W)
{
    if (ad <= -1 || ad >= D || ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0; /* :78-82 */
    const int d0 = (int)floor((double)ad), h0 = (int)floor((double)ah), w0 = (int)floor((double)aw);
    const int md = (pd == d0) ? 0 : (pd == d0 + 1) ? 1 : -1;
    const int mh = (ph == h0) ? 0 : (ph == h0 + 1) ? 1 : -1;
    const int mw = (pw == w0) ? 0 : (pw == w0 + 1) ? 1 : -1;
    if (md < 0 || mh < 0 || mw < 0) return 0; /* none of the eight ifs at :92-107 fires */
    const REAL fd = md ? (ad + 1 - pd) : (pd + 1 - ad);
    const REAL fh = mh ? (ah + 1 - ph) : (ph + 1 - ah);
    const REAL fw = mw ? (aw + 1 - pw) : (pw + 1 - aw);
    return fd * fh * fw;
}

/* d(trilinear value)/d(coordinate along dir), dir 0:d 1:h 2:w.
 * Follows dmcn_get_coordinate_weight, [cuh]:111-190. */
static REAL FN(coord_weight)(REAL ad, REAL ah, REAL aw, int D, int H, int W, const REAL *vol, int dir)
{
    if (ad <= -1 || ad >= D || ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0; /* :116-120 */
    const int d0 = (int)floor((double)ad), h0 = (int)floor((double)ah), w0 = (int)floor((double)aw);
    /* per-axis weights of the low / high corner, written as the reference writes them:
     * low: (x_low + 1 - x), high: (x - x_low)   (:135-150 etc.) */
    const REAL fd[2] = {(REAL)(d0 + 1) - ad, ad - (REAL)d0};
    const REAL fh[2] = {(REAL)(h0 + 1) - ah, ah - (REAL)h0};
    const REAL fw[2] = {(REAL)(w0 + 1) - aw, aw - (REAL)w0};
    REAL acc = 0;
    for (int cd = 0; cd < 2; ++cd)
        for (int ch = 0; ch < 2; ++ch)
            for (int cw = 0; cw < 2; ++cw) {
                const int zd = d0 + cd, zh = h0 + ch, zw = w0 + cw;
                const int ok = (cd ? zd <= D - 1 : zd >= 0) && (ch ? zh <= H - 1 : zh >= 0) &&
                               (cw ? zw <= W - 1 : zw >= 0);
                if (!ok) continue;
                const REAL v = vol[((long)zd * H + zh) * W + zw];
                REAL sgn, a, b;
                if (dir == 0) { sgn = cd ? 1 : -1; a = fh[ch]; b = fw[cw]; }
                else if (dir == 1) { sgn = ch ? 1 : -1; a = fd[cd]; b = fw[cw]; }
                else { sgn = cw ? 1 : -1; a = fd[cd]; b = fh[ch]; }
                acc += sgn * a * b * v;
            }
    return acc;
}

/* ------------------------------------------------------------------------------------------
 * The three reference kernels, as CPU loops over the same flat thread index
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    int B, C, D, H, W;          /* input  (B = images in this im2col step) */
    int Do, Ho, Wo;             /* output spatial */
    int kd, kh, kw, pd, ph, pw, sd, sh, sw, dd, dh, dw;
    int dg;                     /* deformable groups */
} FN(geom);

/* deformable_im2col_gpu_kernel, [cuh]:192-265.  col is [(c*K + tap)][b][Do*Ho*Wo]. */
static void FN(im2col)(const FN(geom) *g, const REAL *im, const REAL *off, REAL *col)
{
    const int K = g->kd * g->kh * g->kw;
    const long No = (long)g->Do * g->Ho * g->Wo, Ni = (long)g->D * g->H * g->W;
    const int cpdg = g->C / g->dg;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < g->C; ++c)
        for (int b = 0; b < g->B; ++b) {
            const REAL *vol = im + ((long)b * g->C + c) * Ni;                        /* :229 */
            const REAL *offp = off + ((long)b * g->dg + c / cpdg) * 3 * K * No;      /* :230 */
            for (int od = 0; od < g->Do; ++od)
                for (int oh = 0; oh < g->Ho; ++oh)
                    for (int ow = 0; ow < g->Wo; ++ow) {
                        const long pos = ((long)od * g->Ho + oh) * g->Wo + ow;
                        const int d_in = od * g->sd - g->pd, h_in = oh * g->sh - g->ph, w_in = ow * g->sw - g->pw; /* :223-225 */
                        int tap = 0;
                        for (int i = 0; i < g->kd; ++i)
                            for (int j = 0; j < g->kh; ++j)
                                for (int k = 0; k < g->kw; ++k, ++tap) {
                                    const REAL od_ = offp[(3 * tap + 0) * No + pos]; /* :237-242 */
                                    const REAL oh_ = offp[(3 * tap + 1) * No + pos];
                                    const REAL ow_ = offp[(3 * tap + 2) * No + pos];
                                    /* integer base first, then + float offset (:244-246) */
                                    const REAL qd = (REAL)(d_in + i * g->dd) + od_;
                                    const REAL qh = (REAL)(h_in + j * g->dh) + oh_;
                                    const REAL qw = (REAL)(w_in + k * g->dw) + ow_;
                                    REAL val = 0;
                                    if (qd > -1 && qh > -1 && qw > -1 && qd < g->D && qh < g->H && qw < g->W) /* :247 */
                                        val = FN(trilinear)(vol, g->D, g->H, g->W, qd, qh, qw);
                                    col[(((long)c * K + tap) * g->B + b) * No + pos] = val;  /* :227,258-259 */
                                }
                    }
        }
}

/* deformable_col2im_gpu_kernel, [cuh]:267-334.  The reference scatters with atomicAdd over a flat
 * index; every contribution of thread (c,tap,b,pos) lands in plane (b,c), so looping planes in
 * parallel and the rest serially gives the same sums without atomics (deterministic order).
 * q1_literal!=0 reproduces the launcher's "pad_d, pad_h, pad_h" argument slip ([cuh]:448). */
static void FN(col2im)(const FN(geom) *g, const REAL *col, const REAL *off, REAL *grad_im, int q1_literal)
{
    const int K = g->kd * g->kh * g->kw;
    const long No = (long)g->Do * g->Ho * g->Wo, Ni = (long)g->D * g->H * g->W;
    const int cpdg = g->C / g->dg;
    const int pw_used = q1_literal ? g->ph : g->pw;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < g->B; ++b)
        for (int c = 0; c < g->C; ++c) {
            REAL *gplane = grad_im + ((long)b * g->C + c) * Ni;
            const REAL *offp = off + ((long)b * g->dg + c / cpdg) * 3 * K * No;
            int tap = 0;
            for (int i = 0; i < g->kd; ++i)
                for (int j = 0; j < g->kh; ++j)
                    for (int k = 0; k < g->kw; ++k, ++tap)
                        for (int od = 0; od < g->Do; ++od)
                            for (int oh = 0; oh < g->Ho; ++oh)
                                for (int ow = 0; ow < g->Wo; ++ow) {
                                    const long pos = ((long)od * g->Ho + oh) * g->Wo + ow;
                                    const REAL qd = (REAL)(od * g->sd - g->pd + i * g->dd) + offp[(3 * tap + 0) * No + pos];
                                    const REAL qh = (REAL)(oh * g->sh - g->ph + j * g->dh) + offp[(3 * tap + 1) * No + pos];
                                    const REAL qw = (REAL)(ow * g->sw - pw_used + k * g->dw) + offp[(3 * tap + 2) * No + pos];
                                    const REAL top = col[(((long)c * K + tap) * g->B + b) * No + pos]; /* :308 */
                                    const int cd = (int)qd, chh = (int)qh, cw = (int)qw;              /* trunc, :309-311 */
                                    for (int dz = -2; dz <= 2; ++dz)
                                        for (int dy = -2; dy <= 2; ++dy)
                                            for (int dx = -2; dx <= 2; ++dx) {
                                                const int zd = cd + dz, zh = chh + dy, zw = cw + dx;
                                                if (zd >= 0 && zd < g->D && zh >= 0 && zh < g->H && zw >= 0 && zw < g->W &&
                                                    fabs((double)(qd - zd)) < 1 && fabs((double)(qh - zh)) < 1 &&
                                                    fabs((double)(qw - zw)) < 1) {                      /* :318-323 */
                                                    const REAL wt = FN(grad_weight_of_voxel)(qd, qh, qw, zd, zh, zw, g->D, g->H, g->W);
                                                    gplane[((long)zd * g->H + zh) * g->W + zw] += wt * top; /* :327 */
                                                }
                                            }
                                }
        }
}

/* deformable_col2im_coord_gpu_kernel, [cuh]:336-405.  One output per offset element. */
static void FN(col2im_coord)(const FN(geom) *g, const REAL *col, const REAL *im, const REAL *off, REAL *grad_off)
{
    const int K = g->kd * g->kh * g->kw;
    const long No = (long)g->Do * g->Ho * g->Wo, Ni = (long)g->D * g->H * g->W;
    const int offC = 3 * K * g->dg;            /* offset_channels, launcher :476 */
    const int cpdg_cols = g->C * K / g->dg;    /* "channel_per_deformable_group" of the launcher, :467 */
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < g->B; ++b)
        for (int oc = 0; oc < offC; ++oc) {
            const int dgi = oc / (3 * K);                                               /* :358 */
            const int offset_c = oc - dgi * 3 * K;                                      /* :365 */
            const int dir = offset_c % 3, tap = offset_c / 3;                            /* :370 */
            const int k = tap % g->kw, j = (tap / g->kw) % g->kh, i = tap / g->kw / g->kh;
            const REAL *colp = col + (long)dgi * cpdg_cols * g->B * No;                  /* :361 */
            const REAL *imp = im + ((long)b * g->dg + dgi) * (g->C / g->dg) * Ni;        /* :362 */
            const REAL *offp = off + ((long)b * g->dg + dgi) * 3 * K * No;               /* :363 */
            for (int od = 0; od < g->Do; ++od)
                for (int oh = 0; oh < g->Ho; ++oh)
                    for (int ow = 0; ow < g->Wo; ++ow) {
                        const long pos = ((long)od * g->Ho + oh) * g->Wo + ow;
                        REAL qd = (REAL)(od * g->sd - g->pd + i * g->dd) + offp[(3 * tap + 0) * No + pos];
                        REAL qh = (REAL)(oh * g->sh - g->ph + j * g->dh) + offp[(3 * tap + 1) * No + pos];
                        REAL qw = (REAL)(ow * g->sw - g->pw + k * g->dw) + offp[(3 * tap + 2) * No + pos];
                        if (qd <= -1 || qh <= -1 || qw <= -1 || qd >= g->D || qh >= g->H || qw >= g->W)
                            qd = qh = qw = -2;                                           /* :391-394 */
                        REAL val = 0;
                        int cnt = 0;
                        for (int col_c = tap; col_c < cpdg_cols; col_c += K, ++cnt) {    /* :368, :400 */
                            const REAL wt = FN(coord_weight)(qd, qh, qw, g->D, g->H, g->W, imp + (long)cnt * Ni, dir);
                            val += wt * colp[((long)col_c * g->B + b) * No + pos];       /* :399 */
                        }
                        grad_off[((long)b * offC + oc) * No + pos] = val;                /* :403 */
                    }
        }
}

/* ------------------------------------------------------------------------------------------
 * Host orchestration (im2col-step loop + per-group GEMMs), [cu]:18-126 and :128-285
 * ---------------------------------------------------------------------------------------- */

static int FN(out_size)(int in, int pad, int dil, int k, int stride)
{
    return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1; /* [cu]:78-80 */
}

/* Returns 0, or a negative code mirroring the reference's AT_ASSERTM checks ([cu]:61-76). */
int FN(dlka_oracle_deform_conv3d_forward)(
    const REAL *input, const REAL *weight, const REAL *bias, const REAL *offset, REAL *output,
    int B, int C, int D, int H, int W, int Cout,
    int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw,
    int group, int dg, int im2col_step)
{
    const int step = B < im2col_step ? B : im2col_step;                 /* :59 */
    if (step <= 0 || B % step != 0) return -1;                          /* :61 */
    if (C % group != 0 || Cout % group != 0) return -2;                 /* :63 */
    if (dg <= 0 || C % dg != 0) return -3;
    const int Do = FN(out_size)(D, pd, dd, kd, sd), Ho = FN(out_size)(H, ph, dh, kh, sh), Wo = FN(out_size)(W, pw, dw, kw, sw);
    if (Do <= 0 || Ho <= 0 || Wo <= 0) return -4;
    const int K = kd * kh * kw, Cg = C / group, Og = Cout / group;
    const long No = (long)Do * Ho * Wo, Ni = (long)D * H * W;
    FN(geom) g = {step, C, D, H, W, Do, Ho, Wo, kd, kh, kw, pd, ph, pw, sd, sh, sw, dd, dh, dw, dg};
    REAL *col = (REAL *)malloc(sizeof(REAL) * (size_t)C * K * step * No); /* :95 */
    if (!col) return -9;
    for (int n = 0; n < B / step; ++n) {
        FN(im2col)(&g, input + (long)n * step * C * Ni, offset + (long)n * step * dg * 3 * K * No, col);
        /* per group: out[(b,pos), co] = bias[co] + sum_k col[g][k][(b,pos)] * W[g][co][k]  (:111-119),
         * stored straight into the permuted NCDHW output (:123). */
#pragma omp parallel for collapse(2) schedule(static)
        for (int co = 0; co < Cout; ++co)
            for (int b = 0; b < step; ++b) {
                const int gi = co / Og;
                const REAL *wrow = weight + (long)co * Cg * K;
                REAL *out = output + (((long)(n * step + b)) * Cout + co) * No;
                for (long p = 0; p < No; ++p) out[p] = bias[co];
                for (int kk = 0; kk < Cg * K; ++kk) {
                    const REAL wv = wrow[kk];
                    const REAL *cp = col + (((long)gi * Cg * K + kk) * step + b) * No;
                    for (long p = 0; p < No; ++p) out[p] += cp[p] * wv;
                }
            }
    }
    free(col);
    return 0;
}

int FN(dlka_oracle_deform_conv3d_backward)(
    const REAL *input, const REAL *weight, const REAL *bias, const REAL *offset, const REAL *grad_output,
    REAL *grad_input, REAL *grad_offset, REAL *grad_weight, REAL *grad_bias,
    int B, int C, int D, int H, int W, int Cout,
    int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw,
    int group, int dg, int im2col_step, int q1_literal)
{
    (void)bias;
    const int step = B < im2col_step ? B : im2col_step;                 /* :168 */
    if (step <= 0 || B % step != 0) return -1;
    if (C % group != 0 || Cout % group != 0) return -2;
    if (dg <= 0 || C % dg != 0) return -3;
    const int Do = FN(out_size)(D, pd, dd, kd, sd), Ho = FN(out_size)(H, ph, dh, kh, sh), Wo = FN(out_size)(W, pw, dw, kw, sw);
    if (Do <= 0 || Ho <= 0 || Wo <= 0) return -4;
    const int K = kd * kh * kw, Cg = C / group, Og = Cout / group;
    const long No = (long)Do * Ho * Wo, Ni = (long)D * H * W;
    FN(geom) g = {step, C, D, H, W, Do, Ho, Wo, kd, kh, kw, pd, ph, pw, sd, sh, sw, dd, dh, dw, dg};
    memset(grad_input, 0, sizeof(REAL) * (size_t)B * C * Ni);           /* zeros_like, :202-205 */
    memset(grad_offset, 0, sizeof(REAL) * (size_t)B * dg * 3 * K * No);
    memset(grad_weight, 0, sizeof(REAL) * (size_t)Cout * Cg * K);
    memset(grad_bias, 0, sizeof(REAL) * (size_t)Cout);
    REAL *col = (REAL *)malloc(sizeof(REAL) * (size_t)C * K * step * No);
    if (!col) return -9;
    for (int n = 0; n < B / step; ++n) {
        const REAL *gout = grad_output + (long)n * step * Cout * No;
        const REAL *in_n = input + (long)n * step * C * Ni;
        const REAL *off_n = offset + (long)n * step * dg * 3 * K * No;
        /* columns[g] = W[g]^T (Cg*K x Og) * gO[g] (Og x step*No)   (:226-231) */
#pragma omp parallel for collapse(2) schedule(static)
        for (int ck = 0; ck < C * K; ++ck)
            for (int b = 0; b < step; ++b) {
                const int gi = ck / (Cg * K), kk = ck % (Cg * K);
                REAL *cp = col + ((long)ck * step + b) * No;
                for (long p = 0; p < No; ++p) cp[p] = 0;
                for (int o = 0; o < Og; ++o) {
                    const int co = gi * Og + o;
                    const REAL wv = weight[((long)co * Cg * K) + kk];
                    const REAL *gp = gout + ((long)b * Cout + co) * No;
                    for (long p = 0; p < No; ++p) cp[p] += wv * gp[p];
                }
            }
        FN(col2im_coord)(&g, col, in_n, off_n, grad_offset + (long)n * step * dg * 3 * K * No);   /* :234-242 */
        FN(col2im)(&g, col, off_n, grad_input + (long)n * step * C * Ni, q1_literal);            /* :244-251 */
        FN(im2col)(&g, in_n, off_n, col);                                                         /* :254-261 */
        /* gW[g] += gO[g] * columns[g]^T ; gB[g] += gO[g] * ones   (:270-278) */
#pragma omp parallel for schedule(static)
        for (int co = 0; co < Cout; ++co) {
            const int gi = co / Og;
            REAL bsum = 0;
            for (int b = 0; b < step; ++b) {
                const REAL *gp = gout + ((long)b * Cout + co) * No;
                for (long p = 0; p < No; ++p) bsum += gp[p];
            }
            grad_bias[co] += bsum;
            for (int kk = 0; kk < Cg * K; ++kk) {
                REAL s = 0;
                for (int b = 0; b < step; ++b) {
                    const REAL *gp = gout + ((long)b * Cout + co) * No;
                    const REAL *cp = col + (((long)gi * Cg * K + kk) * step + b) * No;
                    for (long p = 0; p < No; ++p) s += gp[p] * cp[p];
                }
                grad_weight[(long)co * Cg * K + kk] += s;
            }
        }
    }
    free(col);
    return 0;
}

/* Debug entry point for the "integer sampling indices bit-exact" requirement (BASELINE.json north_star):
 * for every (b, dg-group, tap, out voxel) emit floor(q) for the three axes and the in-range guard
 * of [cuh]:247.  idx is int32 [B][dg][K][No][3], mask is uint8 [B][dg][K][No]. */
int FN(dlka_oracle_deform_conv3d_sample_index)(
    const REAL *offset, int32_t *idx, uint8_t *mask,
    int B, int D, int H, int W,
    int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw, int dg)
{
    const int Do = FN(out_size)(D, pd, dd, kd, sd), Ho = FN(out_size)(H, ph, dh, kh, sh), Wo = FN(out_size)(W, pw, dw, kw, sw);
    if (Do <= 0 || Ho <= 0 || Wo <= 0) return -4;
    const int K = kd * kh * kw;
    const long No = (long)Do * Ho * Wo;
    for (int b = 0; b < B; ++b)
        for (int gi = 0; gi < dg; ++gi) {
            const REAL *offp = offset + ((long)b * dg + gi) * 3 * K * No;
            int tap = 0;
            for (int i = 0; i < kd; ++i)
                for (int j = 0; j < kh; ++j)
                    for (int k = 0; k < kw; ++k, ++tap)
                        for (int od = 0; od < Do; ++od)
                            for (int oh = 0; oh < Ho; ++oh)
                                for (int ow = 0; ow < Wo; ++ow) {
                                    const long pos = ((long)od * Ho + oh) * Wo + ow;
                                    const REAL qd = (REAL)(od * sd - pd + i * dd) + offp[(3 * tap + 0) * No + pos];
                                    const REAL qh = (REAL)(oh * sh - ph + j * dh) + offp[(3 * tap + 1) * No + pos];
                                    const REAL qw = (REAL)(ow * sw - pw + k * dw) + offp[(3 * tap + 2) * No + pos];
                                    const long o = (((long)b * dg + gi) * K + tap) * No + pos;
                                    idx[o * 3 + 0] = (int32_t)floor((double)qd);
                                    idx[o * 3 + 1] = (int32_t)floor((double)qh);
                                    idx[o * 3 + 2] = (int32_t)floor((double)qw);
                                    mask[o] = (qd > -1 && qh > -1 && qw > -1 && qd < D && qh < H && qw < W) ? 1 : 0;
                                }
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * 2-D deformable convolution (torchvision 0.12 deform_conv2d, mask=None)  [tv]
 * ---------------------------------------------------------------------------------------- */

/* bilinear_interpolate [tv]: zero outside (-1,size); per-corner bounds. */
static REAL FN(bilinear)(const REAL *img, int H, int W, REAL y, REAL x)
{
    if (y <= -1 || H <= y || x <= -1 || W <= x) return 0;
    const int y0 = (int)floor((double)y), x0 = (int)floor((double)x);
    const REAL ly = y - (REAL)y0, lx = x - (REAL)x0, hy = (REAL)1 - ly, hx = (REAL)1 - lx;
    const REAL v1 = (y0 >= 0 && x0 >= 0) ? img[(long)y0 * W + x0] : (REAL)0;
    const REAL v2 = (y0 >= 0 && x0 + 1 <= W - 1) ? img[(long)y0 * W + x0 + 1] : (REAL)0;
    const REAL v3 = (y0 + 1 <= H - 1 && x0 >= 0) ? img[(long)(y0 + 1) * W + x0] : (REAL)0;
    const REAL v4 = (y0 + 1 <= H - 1 && x0 + 1 <= W - 1) ? img[(long)(y0 + 1) * W + x0 + 1] : (REAL)0;
    return hy * hx * v1 + hy * lx * v2 + ly * hx * v3 + ly * lx * v4;
}

/* get_coordinate_weight [tv]: derivative of the bilinear value along y (is_y) or x; no range guard. */
static REAL FN(coord_weight2d)(const REAL *img, int H, int W, REAL y, REAL x, int is_y)
{
    const int y0 = (int)floor((double)y), x0 = (int)floor((double)x), y1 = y0 + 1, x1 = x0 + 1;
    const int vy0 = 0 <= y0 && y0 < H, vy1 = 0 <= y1 && y1 < H, vx0 = 0 <= x0 && x0 < W, vx1 = 0 <= x1 && x1 < W;
    const REAL v00 = (vy0 && vx0) ? img[(long)y0 * W + x0] : (REAL)0;
    const REAL v01 = (vy0 && vx1) ? img[(long)y0 * W + x1] : (REAL)0;
    const REAL v10 = (vy1 && vx0) ? img[(long)y1 * W + x0] : (REAL)0;
    const REAL v11 = (vy1 && vx1) ? img[(long)y1 * W + x1] : (REAL)0;
    if (is_y) { const REAL dx = x - (REAL)x0; return dx * (v11 - v01) + ((REAL)1 - dx) * (v10 - v00); }
    else      { const REAL dy = y - (REAL)y0; return dy * (v11 - v10) + ((REAL)1 - dy) * (v01 - v00); }
}

/* forward: im2col ([tv] deformable_im2col_kernel) + per-weight-group addmm + optional bias.
 * offset [B][og*2*K][Ho][Wo] with (dy,dx) per tap; weight [Cout][C/group][kh][kw]. */
int FN(dlka_oracle_deform_conv2d_forward)(
    const REAL *input, const REAL *weight, const REAL *bias /* may be NULL */, const REAL *offset, REAL *output,
    int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
    int group, int og)
{
    if (C % group != 0 || Cout % group != 0) return -2;
    if (og <= 0 || C % og != 0) return -3;
    const int Ho = FN(out_size)(H, ph, dh, kh, sh), Wo = FN(out_size)(W, pw, dw, kw, sw);
    if (Ho <= 0 || Wo <= 0) return -4;
    const int K = kh * kw, Cg = C / group, Og = Cout / group, cpog = C / og;
    const long No = (long)Ho * Wo, Ni = (long)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co) {
            const int gi = co / Og;
            REAL *out = output + ((long)b * Cout + co) * No;
            for (long p = 0; p < No; ++p) out[p] = bias ? bias[co] : (REAL)0;
            for (int cg = 0; cg < Cg; ++cg) {
                const int c = gi * Cg + cg;
                const REAL *img = input + ((long)b * C + c) * Ni;
                const REAL *offp = offset + ((long)b * og + c / cpog) * 2 * K * No;
                for (int i = 0; i < kh; ++i)
                    for (int j = 0; j < kw; ++j) {
                        const int tap = i * kw + j;
                        const REAL wv = weight[((long)co * Cg + cg) * K + tap];
                        for (int oy = 0; oy < Ho; ++oy)
                            for (int ox = 0; ox < Wo; ++ox) {
                                const long pos = (long)oy * Wo + ox;
                                const REAL y = (REAL)(oy * sh - ph + i * dh) + offp[(2 * tap + 0) * No + pos];
                                const REAL x = (REAL)(ox * sw - pw + j * dw) + offp[(2 * tap + 1) * No + pos];
                                out[pos] += wv * FN(bilinear)(img, H, W, y, x);
                            }
                    }
            }
        }
    return 0;
}

/* backward: [tv] compute_grad_input (col2im), compute_grad_offset_and_mask (col2im_coord),
 * backward_gradient_parameters (im2col recomputed, addmm into grad_weight); grad_bias = sum. */
int FN(dlka_oracle_deform_conv2d_backward)(
    const REAL *input, const REAL *weight, const REAL *offset, const REAL *grad_output,
    REAL *grad_input, REAL *grad_offset, REAL *grad_weight, REAL *grad_bias /* may be NULL */,
    int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
    int group, int og)
{
    if (C % group != 0 || Cout % group != 0) return -2;
    if (og <= 0 || C % og != 0) return -3;
    const int Ho = FN(out_size)(H, ph, dh, kh, sh), Wo = FN(out_size)(W, pw, dw, kw, sw);
    if (Ho <= 0 || Wo <= 0) return -4;
    const int K = kh * kw, Cg = C / group, Og = Cout / group, cpog = C / og;
    const long No = (long)Ho * Wo, Ni = (long)H * W;
    memset(grad_input, 0, sizeof(REAL) * (size_t)B * C * Ni);
    memset(grad_offset, 0, sizeof(REAL) * (size_t)B * og * 2 * K * No);
    memset(grad_weight, 0, sizeof(REAL) * (size_t)Cout * Cg * K);
    /* columns[c][tap][b][pos] = sum_{co in group(c)} W[co][c-c0][tap] * gO[b][co][pos] */
    REAL *col = (REAL *)malloc(sizeof(REAL) * (size_t)C * K * B * No);
    if (!col) return -9;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c)
        for (int b = 0; b < B; ++b) {
            const int gi = c / Cg, cg = c % Cg;
            for (int tap = 0; tap < K; ++tap) {
                REAL *cp = col + (((long)c * K + tap) * B + b) * No;
                for (long p = 0; p < No; ++p) cp[p] = 0;
                for (int o = 0; o < Og; ++o) {
                    const int co = gi * Og + o;
                    const REAL wv = weight[((long)co * Cg + cg) * K + tap];
                    const REAL *gp = grad_output + ((long)b * Cout + co) * No;
                    for (long p = 0; p < No; ++p) cp[p] += wv * gp[p];
                }
            }
        }
    /* grad_offset: one value per (b, offset channel, pos), summing over the channels of its offset group */
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int oc = 0; oc < og * 2 * K; ++oc) {
            const int ogi = oc / (2 * K), offset_c = oc - ogi * 2 * K;
            const int is_y = (offset_c % 2 == 0), tap = offset_c / 2, i = tap / kw, j = tap % kw;
            const REAL *offp = offset + ((long)b * og + ogi) * 2 * K * No;
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    const long pos = (long)oy * Wo + ox;
                    const REAL y = (REAL)(oy * sh - ph + i * dh) + offp[(2 * tap + 0) * No + pos];
                    const REAL x = (REAL)(ox * sw - pw + j * dw) + offp[(2 * tap + 1) * No + pos];
                    REAL val = 0;
                    for (int cc = 0; cc < cpog; ++cc) {
                        const int c = ogi * cpog + cc;
                        const REAL wt = FN(coord_weight2d)(input + ((long)b * C + c) * Ni, H, W, y, x, is_y);
                        val += wt * col[(((long)c * K + tap) * B + b) * No + pos];
                    }
                    grad_offset[((long)b * og * 2 * K + oc) * No + pos] = val;
                }
        }
    /* grad_input: scatter over the 3x3 neighbourhood of trunc(coord) with |coord - p| < 1  [tv] deformable_col2im_kernel */
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            REAL *gplane = grad_input + ((long)b * C + c) * Ni;
            const REAL *offp = offset + ((long)b * og + c / cpog) * 2 * K * No;
            for (int tap = 0; tap < K; ++tap) {
                const int i = tap / kw, j = tap % kw;
                for (int oy = 0; oy < Ho; ++oy)
                    for (int ox = 0; ox < Wo; ++ox) {
                        const long pos = (long)oy * Wo + ox;
                        const REAL y = (REAL)(oy * sh - ph + i * dh) + offp[(2 * tap + 0) * No + pos];
                        const REAL x = (REAL)(ox * sw - pw + j * dw) + offp[(2 * tap + 1) * No + pos];
                        const REAL top = col[(((long)c * K + tap) * B + b) * No + pos];
                        for (int dy = -1; dy <= 1; ++dy)
                            for (int dx = -1; dx <= 1; ++dx) {
                                const int yp = (int)y + dy, xp = (int)x + dx;
                                if (0 <= yp && yp < H && 0 <= xp && xp < W && fabs((double)(y - yp)) < 1 && fabs((double)(x - xp)) < 1) {
                                    const REAL wt = ((REAL)1 - (REAL)fabs((double)(y - yp))) * ((REAL)1 - (REAL)fabs((double)(x - xp)));
                                    gplane[(long)yp * W + xp] += wt * top;
                                }
                            }
                    }
            }
        }
    free(col);
    /* grad_weight[co][cg][tap] = sum_{b,pos} gO[b][co][pos] * S(c,tap,b,pos); grad_bias = sum gO */
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Cout; ++co) {
        const int gi = co / Og;
        if (grad_bias) {
            REAL s = 0;
            for (int b = 0; b < B; ++b) {
                const REAL *gp = grad_output + ((long)b * Cout + co) * No;
                for (long p = 0; p < No; ++p) s += gp[p];
            }
            grad_bias[co] = s;
        }
        for (int cg = 0; cg < Cg; ++cg) {
            const int c = gi * Cg + cg;
            for (int tap = 0; tap < K; ++tap) {
                const int i = tap / kw, j = tap % kw;
                REAL s = 0;
                for (int b = 0; b < B; ++b) {
                    const REAL *img = input + ((long)b * C + c) * Ni;
                    const REAL *offp = offset + ((long)b * og + c / cpog) * 2 * K * No;
                    const REAL *gp = grad_output + ((long)b * Cout + co) * No;
                    for (int oy = 0; oy < Ho; ++oy)
                        for (int ox = 0; ox < Wo; ++ox) {
                            const long pos = (long)oy * Wo + ox;
                            const REAL y = (REAL)(oy * sh - ph + i * dh) + offp[(2 * tap + 0) * No + pos];
                            const REAL x = (REAL)(ox * sw - pw + j * dw) + offp[(2 * tap + 1) * No + pos];
                            s += gp[pos] * FN(bilinear)(img, H, W, y, x);
                        }
                }
                grad_weight[((long)co * Cg + cg) * K + tap] = s;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Plain (non-deformable) N-d convolution, naive direct loops — restates what nn.Conv3d /
 * nn.Conv2d compute at the reference call sites (3D/.../synapse/transformerblock.py:637-641,
 * deform_conv.py:80-85; 2D/deformable_LKA/deformable_LKA.py:10-16,95).  2-D = D=1,kd=1.
 * ---------------------------------------------------------------------------------------- */
int FN(dlka_oracle_conv3d_forward)(
    const REAL *input, const REAL *weight, const REAL *bias /* may be NULL */, REAL *output,
    int B, int C, int D, int H, int W, int Cout,
    int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw, int group)
{
    if (C % group != 0 || Cout % group != 0) return -2;
    const int Do = FN(out_size)(D, pd, dd, kd, sd), Ho = FN(out_size)(H, ph, dh, kh, sh), Wo = FN(out_size)(W, pw, dw, kw, sw);
    if (Do <= 0 || Ho <= 0 || Wo <= 0) return -4;
    const int K = kd * kh * kw, Cg = C / group, Og = Cout / group;
    const long No = (long)Do * Ho * Wo, Ni = (long)D * H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co) {
            const int gi = co / Og;
            REAL *out = output + ((long)b * Cout + co) * No;
            for (long p = 0; p < No; ++p) out[p] = bias ? bias[co] : (REAL)0;
            for (int cg = 0; cg < Cg; ++cg) {
                const REAL *vol = input + ((long)b * C + gi * Cg + cg) * Ni;
                const REAL *wk = weight + ((long)co * Cg + cg) * K;
                for (int i = 0; i < kd; ++i)
                    for (int j = 0; j < kh; ++j)
                        for (int k = 0; k < kw; ++k) {
                            const REAL wv = wk[(i * kh + j) * kw + k];
                            for (int od = 0; od < Do; ++od) {
                                const int zd = od * sd - pd + i * dd;
                                if (zd < 0 || zd >= D) continue;
                                for (int oh = 0; oh < Ho; ++oh) {
                                    const int zh = oh * sh - ph + j * dh;
                                    if (zh < 0 || zh >= H) continue;
                                    for (int ow = 0; ow < Wo; ++ow) {
                                        const int zw = ow * sw - pw + k * dw;
                                        if (zw < 0 || zw >= W) continue;
                                        out[((long)od * Ho + oh) * Wo + ow] += wv * vol[((long)zd * H + zh) * W + zw];
                                    }
                                }
                            }
                        }
            }
        }
    return 0;
}

int FN(dlka_oracle_conv3d_backward)(
    const REAL *input, const REAL *weight, const REAL *grad_output,
    REAL *grad_input, REAL *grad_weight, REAL *grad_bias /* may be NULL */,
    int B, int C, int D, int H, int W, int Cout,
    int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw, int group)
{
    if (C % group != 0 || Cout % group != 0) return -2;
    const int Do = FN(out_size)(D, pd, dd, kd, sd), Ho = FN(out_size)(H, ph, dh, kh, sh), Wo = FN(out_size)(W, pw, dw, kw, sw);
    if (Do <= 0 || Ho <= 0 || Wo <= 0) return -4;
    const int K = kd * kh * kw, Cg = C / group, Og = Cout / group;
    const long No = (long)Do * Ho * Wo, Ni = (long)D * H * W;
    memset(grad_input, 0, sizeof(REAL) * (size_t)B * C * Ni);
    /* grad_input: parallel over (b, c) planes */
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const int gi = c / Cg, cg = c % Cg;
            REAL *gvol = grad_input + ((long)b * C + c) * Ni;
            for (int o = 0; o < Og; ++o) {
                const int co = gi * Og + o;
                const REAL *gp = grad_output + ((long)b * Cout + co) * No;
                const REAL *wk = weight + ((long)co * Cg + cg) * K;
                for (int i = 0; i < kd; ++i)
                    for (int j = 0; j < kh; ++j)
                        for (int k = 0; k < kw; ++k) {
                            const REAL wv = wk[(i * kh + j) * kw + k];
                            for (int od = 0; od < Do; ++od) {
                                const int zd = od * sd - pd + i * dd;
                                if (zd < 0 || zd >= D) continue;
                                for (int oh = 0; oh < Ho; ++oh) {
                                    const int zh = oh * sh - ph + j * dh;
                                    if (zh < 0 || zh >= H) continue;
                                    for (int ow = 0; ow < Wo; ++ow) {
                                        const int zw = ow * sw - pw + k * dw;
                                        if (zw < 0 || zw >= W) continue;
                                        gvol[((long)zd * H + zh) * W + zw] += wv * gp[((long)od * Ho + oh) * Wo + ow];
                                    }
                                }
                            }
                        }
            }
        }
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Cout; ++co) {
        const int gi = co / Og;
        if (grad_bias) {
            REAL s = 0;
            for (int b = 0; b < B; ++b) {
                const REAL *gp = grad_output + ((long)b * Cout + co) * No;
                for (long p = 0; p < No; ++p) s += gp[p];
            }
            grad_bias[co] = s;
        }
        for (int cg = 0; cg < Cg; ++cg)
            for (int i = 0; i < kd; ++i)
                for (int j = 0; j < kh; ++j)
                    for (int k = 0; k < kw; ++k) {
                        REAL s = 0;
                        for (int b = 0; b < B; ++b) {
                            const REAL *vol = input + ((long)b * C + gi * Cg + cg) * Ni;
                            const REAL *gp = grad_output + ((long)b * Cout + co) * No;
                            for (int od = 0; od < Do; ++od) {
                                const int zd = od * sd - pd + i * dd;
                                if (zd < 0 || zd >= D) continue;
                                for (int oh = 0; oh < Ho; ++oh) {
                                    const int zh = oh * sh - ph + j * dh;
                                    if (zh < 0 || zh >= H) continue;
                                    for (int ow = 0; ow < Wo; ++ow) {
                                        const int zw = ow * sw - pw + k * dw;
                                        if (zw < 0 || zw >= W) continue;
                                        s += gp[((long)od * Ho + oh) * Wo + ow] * vol[((long)zd * H + zh) * W + zw];
                                    }
                                }
                            }
                        }
                        grad_weight[((long)co * Cg + cg) * K + (i * kh + j) * kw + k] = s;
                    }
    }
    return 0;
}

#undef FN
#undef CAT
#undef CAT_
