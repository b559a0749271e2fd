// C-ABI entry points of libdlka_hip.so (declared in include/dlka.h): argument validation that mirrors the
// reference's AT_ASSERTM checks (3D/dcn/src/cuda/deform_conv_cuda.cu:41-76,193-200), workspace carving and the
// launch sequences.  No allocation, no synchronisation: every function is legal inside hipGraph capture.
#include <algorithm>

#include "dlka_kernels.h"

using namespace dlka;

namespace {

int make_geom(const dlka_conv_geom *c, bool deform, Geom &g)
{
    if (!c) return DLKA_ERR_NULL;
    if (c->B <= 0 || c->C <= 0 || c->D <= 0 || c->H <= 0 || c->W <= 0 || c->Cout <= 0) return DLKA_ERR_SHAPE;
    if (c->kd <= 0 || c->kh <= 0 || c->kw <= 0 || c->sd <= 0 || c->sh <= 0 || c->sw <= 0) return DLKA_ERR_SHAPE;
    if (c->dd <= 0 || c->dh <= 0 || c->dw <= 0 || c->pd < 0 || c->ph < 0 || c->pw < 0) return DLKA_ERR_SHAPE;
    if (c->group <= 0 || c->C % c->group != 0 || c->Cout % c->group != 0) return DLKA_ERR_GROUP;  // cu:65-66
    g.B = c->B; g.C = c->C; g.D = c->D; g.H = c->H; g.W = c->W; g.Cout = c->Cout;
    g.kd = c->kd; g.kh = c->kh; g.kw = c->kw; g.sd = c->sd; g.sh = c->sh; g.sw = c->sw;
    g.pd = c->pd; g.ph = c->ph; g.pw = c->pw; g.dd = c->dd; g.dh = c->dh; g.dw = c->dw;
    g.group = c->group;
    g.dg = deform ? c->deformable_group : 1;
    if (deform) {
        if (g.dg <= 0 || g.C % g.dg != 0) return DLKA_ERR_DEFORM_GROUP;
        // im2col_step only exists for API parity: the reference requires batch % min(batch, step) == 0 (cu:59-63)
        if (c->im2col_step > 0) {
            const int step = c->B < c->im2col_step ? c->B : c->im2col_step;
            if (c->B % step != 0) return DLKA_ERR_IM2COL_STEP;
        } else if (c->im2col_step < 0) {
            return DLKA_ERR_IM2COL_STEP;
        }
    }
    g.Do = dlka_conv_out_size(g.D, g.pd, g.dd, g.kd, g.sd);
    g.Ho = dlka_conv_out_size(g.H, g.ph, g.dh, g.kh, g.sh);
    g.Wo = dlka_conv_out_size(g.W, g.pw, g.dw, g.kw, g.sw);
    if (g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return DLKA_ERR_SHAPE;
    g.K = g.kd * g.kh * g.kw;
    g.Cg = g.C / g.group;
    g.Og = g.Cout / g.group;
    g.cpdg = g.C / g.dg;
    const long No = (long)g.Do * g.Ho * g.Wo, Ni = (long)g.D * g.H * g.W;
    // 32-bit plane indexing inside the kernels
    if (No > (1l << 30) || Ni > (1l << 30) || (long)g.B * g.C > (1l << 24) || (long)g.B * g.Cout > (1l << 24)) return DLKA_ERR_SHAPE;
    g.No = (int)No;
    g.Ni = (int)Ni;
    return DLKA_OK;
}

inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
inline size_t esz(int dtype) { return dtype == DLKA_BF16 ? 2 : 4; }

// simple bump carver over a caller-provided workspace
struct Carver {
    unsigned char *base;
    size_t cap, used;
    Carver(void *p, size_t n) : base((unsigned char *)p), cap(n), used(0) {}
    void *take(size_t n)
    {
        n = align256(n);
        if (!base || used + n > cap) { used = cap + 1; return nullptr; }
        void *r = base + used;
        used += n;
        return r;
    }
    bool ok() const { return used <= cap; }
};

#define DLKA_TRY(expr)            \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != DLKA_OK) return rc_; \
    } while (0)

// ---------------------------------------------------------------------------------------------
// typed implementations
// ---------------------------------------------------------------------------------------------
size_t deform_fwd_ws(const Geom &g) { return align256((size_t)deform_fwd_wt_floats(g) * 4); }

size_t deform_bwd_ws(const Geom &g, int dtype)
{
    size_t n = align256((size_t)deform_fwd_wt_floats(g) * 4);
    if (dtype != DLKA_F32) {
        n += align256((size_t)g.B * g.C * g.Ni * 4);          // fp32 grad_x accumulation
        n += align256((size_t)g.Cout * g.Cg * g.K * 4);       // fp32 grad_weight accumulation
    }
    return n;
}

template <typename T, int NOFF>
int deform_forward_t(const void *x, const void *off, const void *w, const void *bias, void *out, void *ws, size_t wsb,
                     const Geom &g, hipStream_t st)
{
    Carver cv(ws, wsb);
    float *wt = (float *)cv.take((size_t)deform_fwd_wt_floats(g) * 4);
    if (!cv.ok() || !wt) return DLKA_ERR_WORKSPACE;
    return launch_deform_fwd<T, NOFF>((const T *)x, (const T *)off, (const T *)w, (const T *)bias, (T *)out, wt, g, st);
}

template <typename T, int NOFF>
int deform_backward_t(const void *x, const void *off, const void *w, const void *gout, void *gx, void *goff, void *gw,
                      void *gb, void *ws, size_t wsb, const Geom &g, int dtype, hipStream_t st)
{
    Carver cv(ws, wsb);
    const int cob = deform_pick_cob(g.Og);
    const int OgP = round_up(g.Og, cob);
    float *wt = (float *)cv.take((size_t)deform_fwd_wt_floats(g) * 4);
    float *gx32 = nullptr, *gw32 = nullptr;
    const size_t nx = (size_t)g.B * g.C * g.Ni, nw = (size_t)g.Cout * g.Cg * g.K;
    if (dtype == DLKA_F32) {
        gx32 = (float *)gx;
        gw32 = (float *)gw;
    } else {
        if (gx) gx32 = (float *)cv.take(nx * 4);
        if (gw) gw32 = (float *)cv.take(nw * 4);
    }
    if (!cv.ok() || !wt) return DLKA_ERR_WORKSPACE;
    if (gx || goff) {
        DLKA_TRY(launch_relayout_weight<T>((const T *)w, wt, g.group, g.Og, g.Cg, g.K, OgP, st));
        if (gx32) { if (launch_zero(gx32, nx * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH; }
        DLKA_TRY((launch_deform_bwd_input_offset<T, NOFF>((const T *)x, (const T *)off, wt, OgP, (const T *)gout, gx32, (T *)goff, g, st)));
        if (gx && dtype != DLKA_F32) DLKA_TRY(launch_cast_from_f32<T>(gx32, (T *)gx, (long)nx, st));
    }
    if (gw) {
        if (launch_zero(gw32, nw * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
        DLKA_TRY((launch_deform_bwd_weight<T, NOFF>((const T *)x, (const T *)off, (const T *)gout, gw32, g, st)));
        if (dtype != DLKA_F32) DLKA_TRY(launch_cast_from_f32<T>(gw32, (T *)gw, (long)nw, st));
    }
    if (gb) DLKA_TRY(launch_bias_grad<T>((const T *)gout, (T *)gb, g.B, g.Cout, g.No, st));
    return DLKA_OK;
}

size_t conv_fwd_ws(const Geom &g) { return align256((size_t)conv_fwd_wt_floats(g) * 4); }
size_t conv_bwd_ws(const Geom &g, int dtype)
{
    size_t n = align256((size_t)conv_bwd_wb_floats(g) * 4);
    if (dtype != DLKA_F32) n += align256((size_t)g.Cout * g.Cg * g.K * 4);
    n += align256(conv_bwd_weight_part_floats(g, dtype == DLKA_F32 ? 4 : 2) * 4);
    return n;
}

template <typename T>
int conv_forward_t(const void *x, const void *w, const void *bias, void *out, void *ws, size_t wsb, const Geom &g, hipStream_t st)
{
    Carver cv(ws, wsb);
    float *wt = (float *)cv.take((size_t)conv_fwd_wt_floats(g) * 4);
    if (!cv.ok() || !wt) return DLKA_ERR_WORKSPACE;
    return launch_conv_fwd<T>((const T *)x, (const T *)w, (const T *)bias, (T *)out, wt, g, st);
}

template <typename T>
int conv_backward_t(const void *x, const void *w, const void *gout, void *gx, void *gw, void *gb, void *ws, size_t wsb,
                    const Geom &g, int dtype, hipStream_t st)
{
    Carver cv(ws, wsb);
    float *wb = (float *)cv.take((size_t)conv_bwd_wb_floats(g) * 4);
    const size_t nw = (size_t)g.Cout * g.Cg * g.K;
    float *gw32 = nullptr;
    if (gw) gw32 = (dtype == DLKA_F32) ? (float *)gw : (float *)cv.take(nw * 4);
    const size_t npart = conv_bwd_weight_part_floats(g, sizeof(T));
    float *part = npart ? (float *)cv.take(npart * 4) : nullptr;
    if (!cv.ok() || !wb) return DLKA_ERR_WORKSPACE;
    if (gx) DLKA_TRY(launch_conv_bwd_data<T>((const T *)gout, (const T *)w, (T *)gx, wb, g, st));
    if (gw) {
        if (launch_zero(gw32, nw * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
        DLKA_TRY(launch_conv_bwd_weight<T>((const T *)x, (const T *)gout, gw32, g, st, part));
        if (dtype != DLKA_F32) DLKA_TRY(launch_cast_from_f32<T>(gw32, (T *)gw, (long)nw, st));
    }
    if (gb) DLKA_TRY(launch_bias_grad<T>((const T *)gout, (T *)gb, g.B, g.Cout, g.No, st));
    return DLKA_OK;
}

// ---- geometry builders for the fixed convs of the two blocks --------------------------------------------
Geom conv_geom(int B, int C, int Cout, int D, int H, int W, int kd, int kh, int kw, int pd, int ph, int pw, int dil_d, int dil_h,
               int dil_w, int group)
{
    dlka_conv_geom c;
    memset(&c, 0, sizeof(c));
    c.B = B; c.C = C; c.D = D; c.H = H; c.W = W; c.Cout = Cout;
    c.kd = kd; c.kh = kh; c.kw = kw; c.sd = c.sh = c.sw = 1;
    c.pd = pd; c.ph = ph; c.pw = pw; c.dd = dil_d; c.dh = dil_h; c.dw = dil_w;
    c.group = group; c.deformable_group = 1; c.im2col_step = 64;
    Geom g;
    memset(&g, 0, sizeof(g));
    make_geom(&c, true, g);
    return g;
}

struct Lka3dGeoms {
    Geom pw, dw5, dw7, offc, dcn;
    long E, Off;
    Lka3dGeoms(int B, int C, int D, int H, int W)
    {
        pw = conv_geom(B, C, C, D, H, W, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1);
        dw5 = conv_geom(B, C, C, D, H, W, 5, 5, 5, 2, 2, 2, 1, 1, 1, C);        // transformerblock.py:637
        dw7 = conv_geom(B, C, C, D, H, W, 7, 7, 7, 9, 9, 9, 3, 3, 3, C);        // :638
        offc = conv_geom(B, C, 81, D, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 1, 1);      // synapse/deform_conv.py:80-85
        dcn = conv_geom(B, C, C, D, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 1, 1);        // transformerblock.py:639
        E = (long)B * C * D * H * W;
        Off = (long)B * 81 * D * H * W;
    }
    size_t scratch_floats() const
    {
        size_t m = 0;
        const Geom *all[5] = {&pw, &dw5, &dw7, &offc, &dcn};
        for (const Geom *g : all) {
            m = m > (size_t)conv_fwd_wt_floats(*g) ? m : (size_t)conv_fwd_wt_floats(*g);
            m = m > (size_t)conv_bwd_wb_floats(*g) ? m : (size_t)conv_bwd_wb_floats(*g);
        }
        m = m > (size_t)deform_fwd_wt_floats(dcn) ? m : (size_t)deform_fwd_wt_floats(dcn);
        return m;
    }
    size_t max_weight_elems() const
    {
        size_t m = 0;
        const Geom *all[5] = {&pw, &dw5, &dw7, &offc, &dcn};
        for (const Geom *g : all) {
            size_t n = (size_t)g->Cout * g->Cg * g->K;
            m = m > n ? m : n;
        }
        return m;
    }
};

template <typename T>
int lka3d_forward_t(const void *x, const dlka_lka3d_params *p, void *y, void *saved, size_t savedb, void *ws, size_t wsb,
                    int B, int C, int D, int H, int W, hipStream_t st)
{
    Lka3dGeoms G(B, C, D, H, W);
    const size_t e = sizeof(T);
    Carver sv(saved, savedb), cv(ws, wsb);
    T *h = (T *)sv.take(G.E * e), *a = (T *)sv.take(G.E * e), *t1 = (T *)sv.take(G.E * e), *t = (T *)sv.take(G.E * e);
    T *off = (T *)sv.take(G.Off * e), *f = (T *)sv.take(G.E * e), *g1 = (T *)sv.take(G.E * e);
    float *wt = (float *)cv.take(G.scratch_floats() * 4);
    T *m = (T *)cv.take(G.E * e);
    if (!sv.ok() || !cv.ok()) return DLKA_ERR_WORKSPACE;
    DLKA_TRY(launch_conv_fwd<T>((const T *)x, (const T *)p->proj_1_w, (const T *)p->proj_1_b, h, wt, G.pw, st));          // :667
    DLKA_TRY(launch_gelu_fwd<T>(h, a, G.E, st));                                                                          // :668
    DLKA_TRY(launch_conv_fwd<T>(a, (const T *)p->conv0_w, (const T *)p->conv0_b, t1, wt, G.dw5, st));                     // :646
    DLKA_TRY(launch_conv_fwd<T>(t1, (const T *)p->conv_spatial_w, (const T *)p->conv_spatial_b, t, wt, G.dw7, st));       // :647
    DLKA_TRY(launch_conv_fwd<T>(t, (const T *)p->offset_w, (const T *)p->offset_b, off, wt, G.offc, st));                 // deform_conv.py:94
    DLKA_TRY((launch_deform_fwd<T, 3>(t, off, (const T *)p->deform_w, (const T *)p->deform_b, f, wt, G.dcn, st)));        // deform_conv.py:95-105
    DLKA_TRY(launch_conv_fwd<T>(f, (const T *)p->conv1_w, (const T *)p->conv1_b, g1, wt, G.pw, st));                      // :650
    DLKA_TRY(launch_mul_fwd<T>(a, g1, m, G.E, st));                                                                       // :652
    DLKA_TRY(launch_conv_fwd<T>(m, (const T *)p->proj_2_w, (const T *)p->proj_2_b, (T *)y, wt, G.pw, st));                // :670
    DLKA_TRY(launch_add_fwd<T>((const T *)y, (const T *)x, (T *)y, G.E, st));                                             // :671
    return DLKA_OK;
}

template <typename T>
int conv_bwd_all(const T *xin, const T *w, const T *gout, T *gx, void *gw, void *gb, float *scratch, float *gw32buf,
                 const Geom &g, int dtype, hipStream_t st)
{
    const size_t nw = (size_t)g.Cout * g.Cg * g.K;
    if (gx) DLKA_TRY(launch_conv_bwd_data<T>(gout, w, gx, scratch, g, st));
    if (gw) {
        float *gw32 = (dtype == DLKA_F32) ? (float *)gw : gw32buf;
        if (launch_zero(gw32, nw * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
        DLKA_TRY(launch_conv_bwd_weight<T>(xin, gout, gw32, g, st));
        if (dtype != DLKA_F32) DLKA_TRY(launch_cast_from_f32<T>(gw32, (T *)gw, (long)nw, st));
    }
    if (gb) DLKA_TRY(launch_bias_grad<T>(gout, (T *)gb, g.B, g.Cout, g.No, st));
    return DLKA_OK;
}

template <typename T>
int lka3d_backward_t(const void *x_, const dlka_lka3d_params *p, const void *gy_, const void *saved, size_t savedb, void *gx_,
                     const dlka_lka3d_grads *gr, void *ws, size_t wsb, int B, int C, int D, int H, int W, int dtype, hipStream_t st)
{
    Lka3dGeoms G(B, C, D, H, W);
    const size_t e = sizeof(T);
    Carver sv((void *)saved, savedb), cv(ws, wsb);
    const T *h = (T *)sv.take(G.E * e), *a = (T *)sv.take(G.E * e), *t1 = (T *)sv.take(G.E * e), *t = (T *)sv.take(G.E * e);
    const T *off = (T *)sv.take(G.Off * e), *f = (T *)sv.take(G.E * e), *g1 = (T *)sv.take(G.E * e);
    float *scr = (float *)cv.take(G.scratch_floats() * 4);
    T *bA = (T *)cv.take(G.E * e), *bB = (T *)cv.take(G.E * e), *bC = (T *)cv.take(G.E * e), *bD = (T *)cv.take(G.E * e);
    T *bO = (T *)cv.take(G.Off * e);
    float *gw32 = nullptr, *gx32 = nullptr;
    if (dtype != DLKA_F32) {
        gw32 = (float *)cv.take(G.max_weight_elems() * 4);
        gx32 = (float *)cv.take(G.E * 4);
    }
    if (!sv.ok() || !cv.ok()) return DLKA_ERR_WORKSPACE;
    const T *x = (const T *)x_, *gy = (const T *)gy_;
    T *gx = (T *)gx_;

    // proj_2 (+ residual: d(x + ...)/dx adds gy at the very end)                           transformerblock.py:670-671
    DLKA_TRY(launch_mul_fwd<T>(a, g1, bA, G.E, st));                                        // m = a * g1 (recomputed)
    DLKA_TRY(conv_bwd_all<T>(bA, (const T *)p->proj_2_w, gy, bB, gr->proj_2_w, gr->proj_2_b, scr, gw32, G.pw, dtype, st));  // bB = gm
    // gate u * attn                                                                        :652
    DLKA_TRY(launch_mul_bwd<T>(a, g1, bB, bC, bD, G.E, st));                                // bC = ga1 = gm*g1 ; bD = gg1 = gm*a
    // conv1                                                                                :650
    DLKA_TRY(conv_bwd_all<T>(f, (const T *)p->conv1_w, bD, bB, gr->conv1_w, gr->conv1_b, scr, gw32, G.pw, dtype, st));      // bB = gf
    // deformable conv                                                                      deform_conv.py:95-105
    {
        const Geom &g = G.dcn;
        const int OgP = round_up(g.Og, deform_pick_cob(g.Og));
        const size_t nx = (size_t)G.E, nw = (size_t)g.Cout * g.Cg * g.K;
        float *gxa = (dtype == DLKA_F32) ? (float *)bA : gx32;
        float *gwa = (dtype == DLKA_F32) ? (float *)gr->deform_w : gw32;
        DLKA_TRY(launch_relayout_weight<T>((const T *)p->deform_w, scr, g.group, g.Og, g.Cg, g.K, OgP, st));
        if (launch_zero(gxa, nx * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
        DLKA_TRY((launch_deform_bwd_input_offset<T, 3>(t, off, scr, OgP, bB, gxa, bO, g, st)));   // bA = gt_a, bO = goff
        if (dtype != DLKA_F32) DLKA_TRY(launch_cast_from_f32<T>(gxa, bA, (long)nx, st));
        if (launch_zero(gwa, nw * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
        DLKA_TRY((launch_deform_bwd_weight<T, 3>(t, off, bB, gwa, g, st)));
        if (dtype != DLKA_F32) DLKA_TRY(launch_cast_from_f32<T>(gwa, (T *)gr->deform_w, (long)nw, st));
        DLKA_TRY(launch_bias_grad<T>(bB, (T *)gr->deform_b, g.B, g.Cout, g.No, st));
    }
    // offset-predict conv                                                                  deform_conv.py:94
    DLKA_TRY(conv_bwd_all<T>(t, (const T *)p->offset_w, bO, bD, gr->offset_w, gr->offset_b, scr, gw32, G.offc, dtype, st)); // bD = gt_b
    DLKA_TRY(launch_add_fwd<T>(bA, bD, bA, G.E, st));                                       // bA = gt
    // depthwise 7^3 dil 3                                                                  :647
    DLKA_TRY(conv_bwd_all<T>(t1, (const T *)p->conv_spatial_w, bA, bB, gr->conv_spatial_w, gr->conv_spatial_b, scr, gw32, G.dw7, dtype, st));  // bB = gt1
    // depthwise 5^3                                                                        :646
    DLKA_TRY(conv_bwd_all<T>(a, (const T *)p->conv0_w, bB, bD, gr->conv0_w, gr->conv0_b, scr, gw32, G.dw5, dtype, st));     // bD = ga2
    DLKA_TRY(launch_add_fwd<T>(bC, bD, bC, G.E, st));                                       // bC = ga
    // GELU                                                                                 :668
    DLKA_TRY(launch_gelu_bwd<T>(h, bC, bA, G.E, st));                                       // bA = gh
    // proj_1                                                                               :667
    DLKA_TRY(conv_bwd_all<T>(x, (const T *)p->proj_1_w, bA, bB, gr->proj_1_w, gr->proj_1_b, scr, gw32, G.pw, dtype, st));   // bB = gx1
    DLKA_TRY(launch_add_fwd<T>(bB, gy, gx, G.E, st));                                       // + shortcut
    return DLKA_OK;
}

// ---- 2-D block ---------------------------------------------------------------------------------------------
struct Lka2dGeoms {
    Geom pw, off5, dcn5, off7, dcn7;
    long E, Off5, Off7;
    Lka2dGeoms(int B, int C, int H, int W)
    {
        pw = conv_geom(B, C, C, 1, H, W, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1);
        off5 = conv_geom(B, C, 50, 1, H, W, 1, 5, 5, 0, 2, 2, 1, 1, 1, 1);     // deformable_LKA.py:10-16 via :93
        dcn5 = conv_geom(B, C, C, 1, H, W, 1, 5, 5, 0, 2, 2, 1, 1, 1, C);      // :18-25 via :93
        off7 = conv_geom(B, C, 98, 1, H, W, 1, 7, 7, 0, 9, 9, 1, 3, 3, 1);     // :94
        dcn7 = conv_geom(B, C, C, 1, H, W, 1, 7, 7, 0, 9, 9, 1, 3, 3, C);      // :94
        E = (long)B * C * H * W;
        Off5 = (long)B * 50 * H * W;
        Off7 = (long)B * 98 * H * W;
    }
    size_t scratch_floats() const
    {
        size_t m = 0;
        const Geom *all[5] = {&pw, &off5, &dcn5, &off7, &dcn7};
        for (const Geom *g : all) {
            m = m > (size_t)conv_fwd_wt_floats(*g) ? m : (size_t)conv_fwd_wt_floats(*g);
            m = m > (size_t)conv_bwd_wb_floats(*g) ? m : (size_t)conv_bwd_wb_floats(*g);
            m = m > (size_t)deform_fwd_wt_floats(*g) ? m : (size_t)deform_fwd_wt_floats(*g);
        }
        return m;
    }
    size_t max_weight_elems() const
    {
        size_t m = 0;
        const Geom *all[5] = {&pw, &off5, &dcn5, &off7, &dcn7};
        for (const Geom *g : all) {
            size_t n = (size_t)g->Cout * g->Cg * g->K;
            m = m > n ? m : n;
        }
        return m;
    }
};

template <typename T>
int lka2d_forward_t(const void *x, const dlka_lka2d_params *p, void *y, void *saved, size_t savedb, void *ws, size_t wsb,
                    int B, int C, int H, int W, hipStream_t st)
{
    Lka2dGeoms G(B, C, H, W);
    const size_t e = sizeof(T);
    Carver sv(saved, savedb), cv(ws, wsb);
    T *h = (T *)sv.take(G.E * e), *a = (T *)sv.take(G.E * e), *o5 = (T *)sv.take(G.Off5 * e), *t1 = (T *)sv.take(G.E * e);
    T *o7 = (T *)sv.take(G.Off7 * e), *t2 = (T *)sv.take(G.E * e), *g1 = (T *)sv.take(G.E * e);
    float *wt = (float *)cv.take(G.scratch_floats() * 4);
    T *m = (T *)cv.take(G.E * e);
    if (!sv.ok() || !cv.ok()) return DLKA_ERR_WORKSPACE;
    DLKA_TRY(launch_conv_fwd<T>((const T *)x, (const T *)p->proj_1_w, (const T *)p->proj_1_b, h, wt, G.pw, st));                     // :135
    DLKA_TRY(launch_gelu_fwd<T>(h, a, G.E, st));                                                                                     // :136
    DLKA_TRY(launch_conv_fwd<T>(a, (const T *)p->conv0_offset_w, (const T *)p->conv0_offset_b, o5, wt, G.off5, st));                 // :28
    DLKA_TRY((launch_deform_fwd<T, 2>(a, o5, (const T *)p->conv0_w, (const T *)nullptr, t1, wt, G.dcn5, st)));                       // :29
    DLKA_TRY(launch_conv_fwd<T>(t1, (const T *)p->conv_spatial_offset_w, (const T *)p->conv_spatial_offset_b, o7, wt, G.off7, st));  // :28
    DLKA_TRY((launch_deform_fwd<T, 2>(t1, o7, (const T *)p->conv_spatial_w, (const T *)nullptr, t2, wt, G.dcn7, st)));               // :29
    DLKA_TRY(launch_conv_fwd<T>(t2, (const T *)p->conv1_w, (const T *)p->conv1_b, g1, wt, G.pw, st));                                // :102
    DLKA_TRY(launch_mul_fwd<T>(a, g1, m, G.E, st));                                                                                  // :104
    DLKA_TRY(launch_conv_fwd<T>(m, (const T *)p->proj_2_w, (const T *)p->proj_2_b, (T *)y, wt, G.pw, st));                           // :138
    DLKA_TRY(launch_add_fwd<T>((const T *)y, (const T *)x, (T *)y, G.E, st));                                                        // :139
    return DLKA_OK;
}

// deformable depthwise conv backward inside the 2-D block: input `xin`, offsets `off`, grad_out `go`
// -> gxin (T buffer), goff (T buffer), grad weight
template <typename T>
int deform2d_bwd_all(const T *xin, const T *off, const T *w, const T *go, T *gxin, T *goff, void *gw, float *scr, float *gx32,
                     float *gw32buf, const Geom &g, int dtype, hipStream_t st)
{
    const int OgP = round_up(g.Og, deform_pick_cob(g.Og));
    const size_t nx = (size_t)g.B * g.C * g.Ni, nw = (size_t)g.Cout * g.Cg * g.K;
    float *gxa = (dtype == DLKA_F32) ? (float *)gxin : gx32;
    float *gwa = (dtype == DLKA_F32) ? (float *)gw : gw32buf;
    DLKA_TRY(launch_relayout_weight<T>(w, scr, g.group, g.Og, g.Cg, g.K, OgP, st));
    if (launch_zero(gxa, nx * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
    DLKA_TRY((launch_deform_bwd_input_offset<T, 2>(xin, off, scr, OgP, go, gxa, goff, g, st)));
    if (dtype != DLKA_F32) DLKA_TRY(launch_cast_from_f32<T>(gxa, gxin, (long)nx, st));
    if (launch_zero(gwa, nw * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
    DLKA_TRY((launch_deform_bwd_weight<T, 2>(xin, off, go, gwa, g, st)));
    if (dtype != DLKA_F32) DLKA_TRY(launch_cast_from_f32<T>(gwa, (T *)gw, (long)nw, st));
    return DLKA_OK;
}

template <typename T>
int lka2d_backward_t(const void *x_, const dlka_lka2d_params *p, const void *gy_, const void *saved, size_t savedb, void *gx_,
                     const dlka_lka2d_grads *gr, void *ws, size_t wsb, int B, int C, int H, int W, int dtype, hipStream_t st)
{
    Lka2dGeoms G(B, C, H, W);
    const size_t e = sizeof(T);
    Carver sv((void *)saved, savedb), cv(ws, wsb);
    const T *h = (T *)sv.take(G.E * e), *a = (T *)sv.take(G.E * e), *o5 = (T *)sv.take(G.Off5 * e), *t1 = (T *)sv.take(G.E * e);
    const T *o7 = (T *)sv.take(G.Off7 * e), *t2 = (T *)sv.take(G.E * e), *g1 = (T *)sv.take(G.E * e);
    float *scr = (float *)cv.take(G.scratch_floats() * 4);
    T *bA = (T *)cv.take(G.E * e), *bB = (T *)cv.take(G.E * e), *bC = (T *)cv.take(G.E * e), *bD = (T *)cv.take(G.E * e);
    T *bO = (T *)cv.take(G.Off7 * e);
    float *gw32 = nullptr, *gx32 = nullptr;
    if (dtype != DLKA_F32) {
        gw32 = (float *)cv.take(G.max_weight_elems() * 4);
        gx32 = (float *)cv.take(G.E * 4);
    }
    if (!sv.ok() || !cv.ok()) return DLKA_ERR_WORKSPACE;
    const T *x = (const T *)x_, *gy = (const T *)gy_;
    T *gx = (T *)gx_;
    DLKA_TRY(launch_mul_fwd<T>(a, g1, bA, G.E, st));                                                                        // m
    DLKA_TRY(conv_bwd_all<T>(bA, (const T *)p->proj_2_w, gy, bB, gr->proj_2_w, gr->proj_2_b, scr, gw32, G.pw, dtype, st));  // bB = gm
    DLKA_TRY(launch_mul_bwd<T>(a, g1, bB, bC, bD, G.E, st));                                                                // bC = ga1, bD = gg1
    DLKA_TRY(conv_bwd_all<T>(t2, (const T *)p->conv1_w, bD, bB, gr->conv1_w, gr->conv1_b, scr, gw32, G.pw, dtype, st));     // bB = gt2
    // conv_spatial: t2 = dcn7(t1, o7 = off7(t1))
    DLKA_TRY(deform2d_bwd_all<T>(t1, o7, (const T *)p->conv_spatial_w, bB, bA, bO, gr->conv_spatial_w, scr, gx32, gw32, G.dcn7, dtype, st));  // bA = gt1_a, bO = go7
    DLKA_TRY(conv_bwd_all<T>(t1, (const T *)p->conv_spatial_offset_w, bO, bD, gr->conv_spatial_offset_w, gr->conv_spatial_offset_b, scr, gw32, G.off7, dtype, st));  // bD = gt1_b
    DLKA_TRY(launch_add_fwd<T>(bA, bD, bA, G.E, st));                                                                       // bA = gt1
    // conv0: t1 = dcn5(a, o5 = off5(a))
    DLKA_TRY(deform2d_bwd_all<T>(a, o5, (const T *)p->conv0_w, bA, bB, bO, gr->conv0_w, scr, gx32, gw32, G.dcn5, dtype, st));  // bB = ga2_a, bO = go5
    DLKA_TRY(conv_bwd_all<T>(a, (const T *)p->conv0_offset_w, bO, bD, gr->conv0_offset_w, gr->conv0_offset_b, scr, gw32, G.off5, dtype, st));  // bD = ga2_b
    DLKA_TRY(launch_add_fwd<T>(bB, bD, bB, G.E, st));
    DLKA_TRY(launch_add_fwd<T>(bC, bB, bC, G.E, st));                                                                       // bC = ga
    DLKA_TRY(launch_gelu_bwd<T>(h, bC, bA, G.E, st));                                                                       // bA = gh
    DLKA_TRY(conv_bwd_all<T>(x, (const T *)p->proj_1_w, bA, bB, gr->proj_1_w, gr->proj_1_b, scr, gw32, G.pw, dtype, st));   // bB = gx1
    DLKA_TRY(launch_add_fwd<T>(bB, gy, gx, G.E, st));
    return DLKA_OK;
}

}  // namespace

// =================================================================================================================
// extern "C"
// =================================================================================================================
extern "C" {

int dlka_abi_version(void) { return DLKA_ABI_VERSION; }

const char *dlka_status_string(int s)
{
    switch (s) {
        case DLKA_OK: return "ok";
        case DLKA_ERR_NULL: return "null pointer argument";
        case DLKA_ERR_GROUP: return "channels and channels_out must be divisible by group";
        case DLKA_ERR_DEFORM_GROUP: return "channels must be divisible by deformable_group";
        case DLKA_ERR_SHAPE: return "invalid shape: non-positive or too large size / empty output";
        case DLKA_ERR_IM2COL_STEP: return "batch must be divisible by min(batch, im2col_step)";
        case DLKA_ERR_DTYPE: return "unsupported dtype";
        case DLKA_ERR_WORKSPACE: return "workspace / saved buffer missing or too small";
        case DLKA_ERR_UNSUPPORTED: return "unsupported parameter combination";
        case DLKA_ERR_LAUNCH: return "HIP kernel launch failed";
        default: return "unknown dlka status";
    }
}

int dlka_conv_out_size(int in, int pad, int dil, int k, int stride)
{
    if (stride <= 0) return 0;
    const int num = in + 2 * pad - (dil * (k - 1) + 1);
    if (num < 0) return 0;
    return num / stride + 1;
}

#define DLKA_DISPATCH(dtype, CALL_F32, CALL_BF16) \
    ((dtype) == DLKA_F32 ? (CALL_F32) : (dtype) == DLKA_BF16 ? (CALL_BF16) : DLKA_ERR_DTYPE)

// ---- 3-D deformable -------------------------------------------------------------------------------------------
size_t dlka_deform_conv3d_forward_workspace(const dlka_conv_geom *c, int dtype)
{
    (void)dtype;
    Geom g;
    if (make_geom(c, true, g)) return 0;
    return deform_fwd_ws(g);
}

int dlka_deform_conv3d_forward(const void *x, const void *offset, const void *weight, const void *bias, void *out,
                               void *workspace, size_t workspace_bytes, const dlka_conv_geom *c, int dtype, void *stream)
{
    if (!x || !offset || !weight || !bias || !out) return DLKA_ERR_NULL;
    Geom g;
    DLKA_TRY(make_geom(c, true, g));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DLKA_F64) return launch_deform_fwd_f64<3>((const double *)x, (const double *)offset, (const double *)weight, (const double *)bias, (double *)out, g, st);
    return DLKA_DISPATCH(dtype, (deform_forward_t<float, 3>(x, offset, weight, bias, out, workspace, workspace_bytes, g, st)),
                         (deform_forward_t<bf16_t, 3>(x, offset, weight, bias, out, workspace, workspace_bytes, g, st)));
}

size_t dlka_deform_conv3d_backward_workspace(const dlka_conv_geom *c, int dtype)
{
    Geom g;
    if (make_geom(c, true, g)) return 0;
    return deform_bwd_ws(g, dtype);
}

int dlka_deform_conv3d_backward(const void *x, const void *offset, const void *weight, const void *grad_out, void *grad_x,
                                void *grad_offset, void *grad_weight, void *grad_bias, void *workspace, size_t workspace_bytes,
                                const dlka_conv_geom *c, int dtype, void *stream)
{
    if (!x || !offset || !weight || !grad_out) return DLKA_ERR_NULL;
    Geom g;
    DLKA_TRY(make_geom(c, true, g));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DLKA_F64)
        return launch_deform_bwd_f64<3>((const double *)x, (const double *)offset, (const double *)weight, (const double *)grad_out, (double *)grad_x, (double *)grad_offset,
                                        (double *)grad_weight, (double *)grad_bias, g, st);
    return DLKA_DISPATCH(dtype,
                         (deform_backward_t<float, 3>(x, offset, weight, grad_out, grad_x, grad_offset, grad_weight, grad_bias, workspace, workspace_bytes, g, dtype, st)),
                         (deform_backward_t<bf16_t, 3>(x, offset, weight, grad_out, grad_x, grad_offset, grad_weight, grad_bias, workspace, workspace_bytes, g, dtype, st)));
}

int dlka_deform_conv3d_sample_index_path(const void *offset, int32_t *idx, uint8_t *mask, const dlka_conv_geom *c, int dtype, int path, void *stream)
{
    if (!offset || !idx || !mask) return DLKA_ERR_NULL;
    Geom g;
    DLKA_TRY(make_geom(c, true, g));
    hipStream_t st = (hipStream_t)stream;
    return DLKA_DISPATCH(dtype, launch_sample_index<float>((const float *)offset, idx, mask, g, path, st),
                         launch_sample_index<bf16_t>((const bf16_t *)offset, idx, mask, g, path, st));
}

int dlka_deform_conv3d_sample_index(const void *offset, int32_t *idx, uint8_t *mask, const dlka_conv_geom *c, int dtype, void *stream)
{
    return dlka_deform_conv3d_sample_index_path(offset, idx, mask, c, dtype, 0, stream);
}

// ---- 2-D deformable -------------------------------------------------------------------------------------------
static int check_2d(const dlka_conv_geom *c)
{
    if (!c) return DLKA_ERR_NULL;
    if (c->D != 1 || c->kd != 1 || c->sd != 1 || c->dd != 1 || c->pd != 0) return DLKA_ERR_SHAPE;
    return DLKA_OK;
}

size_t dlka_deform_conv2d_forward_workspace(const dlka_conv_geom *c, int dtype)
{
    (void)dtype;
    Geom g;
    if (check_2d(c) || make_geom(c, true, g)) return 0;
    return deform_fwd_ws(g);
}

int dlka_deform_conv2d_forward(const void *x, const void *offset, const void *weight, const void *bias, void *out,
                               void *workspace, size_t workspace_bytes, const dlka_conv_geom *c, int dtype, void *stream)
{
    if (!x || !offset || !weight || !out) return DLKA_ERR_NULL;
    DLKA_TRY(check_2d(c));
    Geom g;
    DLKA_TRY(make_geom(c, true, g));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DLKA_F64) return launch_deform_fwd_f64<2>((const double *)x, (const double *)offset, (const double *)weight, (const double *)bias, (double *)out, g, st);
    return DLKA_DISPATCH(dtype, (deform_forward_t<float, 2>(x, offset, weight, bias, out, workspace, workspace_bytes, g, st)),
                         (deform_forward_t<bf16_t, 2>(x, offset, weight, bias, out, workspace, workspace_bytes, g, st)));
}

size_t dlka_deform_conv2d_backward_workspace(const dlka_conv_geom *c, int dtype)
{
    Geom g;
    if (check_2d(c) || make_geom(c, true, g)) return 0;
    return deform_bwd_ws(g, dtype);
}

int dlka_deform_conv2d_backward(const void *x, const void *offset, const void *weight, const void *grad_out, void *grad_x,
                                void *grad_offset, void *grad_weight, void *grad_bias, void *workspace, size_t workspace_bytes,
                                const dlka_conv_geom *c, int dtype, void *stream)
{
    if (!x || !offset || !weight || !grad_out) return DLKA_ERR_NULL;
    DLKA_TRY(check_2d(c));
    Geom g;
    DLKA_TRY(make_geom(c, true, g));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DLKA_F64)
        return launch_deform_bwd_f64<2>((const double *)x, (const double *)offset, (const double *)weight, (const double *)grad_out, (double *)grad_x, (double *)grad_offset,
                                        (double *)grad_weight, (double *)grad_bias, g, st);
    return DLKA_DISPATCH(dtype,
                         (deform_backward_t<float, 2>(x, offset, weight, grad_out, grad_x, grad_offset, grad_weight, grad_bias, workspace, workspace_bytes, g, dtype, st)),
                         (deform_backward_t<bf16_t, 2>(x, offset, weight, grad_out, grad_x, grad_offset, grad_weight, grad_bias, workspace, workspace_bytes, g, dtype, st)));
}

int dlka_deform_conv2d_sample_index_path(const void *offset, int32_t *idx, uint8_t *mask, const dlka_conv_geom *c, int dtype, int path, void *stream)
{
    if (!offset || !idx || !mask) return DLKA_ERR_NULL;
    DLKA_TRY(check_2d(c));
    Geom g;
    DLKA_TRY(make_geom(c, true, g));
    hipStream_t st = (hipStream_t)stream;
    return DLKA_DISPATCH(dtype, launch_sample_index2<float>((const float *)offset, idx, mask, g, path, st),
                         launch_sample_index2<bf16_t>((const bf16_t *)offset, idx, mask, g, path, st));
}

// ---- plain conv -----------------------------------------------------------------------------------------------
size_t dlka_conv3d_forward_workspace(const dlka_conv_geom *c, int dtype)
{
    (void)dtype;
    Geom g;
    if (make_geom(c, false, g)) return 0;
    return conv_fwd_ws(g);
}

int dlka_conv3d_forward(const void *x, const void *weight, const void *bias, void *out, void *workspace, size_t workspace_bytes,
                        const dlka_conv_geom *c, int dtype, void *stream)
{
    if (!x || !weight || !out) return DLKA_ERR_NULL;
    Geom g;
    DLKA_TRY(make_geom(c, false, g));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DLKA_F64) return launch_conv_fwd_f64((const double *)x, (const double *)weight, (const double *)bias, (double *)out, g, st);
    return DLKA_DISPATCH(dtype, conv_forward_t<float>(x, weight, bias, out, workspace, workspace_bytes, g, st),
                         conv_forward_t<bf16_t>(x, weight, bias, out, workspace, workspace_bytes, g, st));
}

size_t dlka_conv3d_backward_workspace(const dlka_conv_geom *c, int dtype)
{
    Geom g;
    if (make_geom(c, false, g)) return 0;
    return conv_bwd_ws(g, dtype);
}

int dlka_conv3d_backward(const void *x, const void *weight, const void *grad_out, void *grad_x, void *grad_weight, void *grad_bias,
                         void *workspace, size_t workspace_bytes, const dlka_conv_geom *c, int dtype, void *stream)
{
    if (!x || !weight || !grad_out) return DLKA_ERR_NULL;
    Geom g;
    DLKA_TRY(make_geom(c, false, g));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DLKA_F64)
        return launch_conv_bwd_f64((const double *)x, (const double *)weight, (const double *)grad_out, (double *)grad_x, (double *)grad_weight, (double *)grad_bias, g, st);
    return DLKA_DISPATCH(dtype, conv_backward_t<float>(x, weight, grad_out, grad_x, grad_weight, grad_bias, workspace, workspace_bytes, g, dtype, st),
                         conv_backward_t<bf16_t>(x, weight, grad_out, grad_x, grad_weight, grad_bias, workspace, workspace_bytes, g, dtype, st));
}

// ---- elementwise ----------------------------------------------------------------------------------------------
int dlka_gelu_forward(const void *x, void *y, int64_t n, int dtype, void *stream)
{
    if (!x || !y) return DLKA_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    return DLKA_DISPATCH(dtype, launch_gelu_fwd<float>((const float *)x, (float *)y, (long)n, st),
                         launch_gelu_fwd<bf16_t>((const bf16_t *)x, (bf16_t *)y, (long)n, st));
}
int dlka_gelu_backward(const void *x, const void *gy, void *gx, int64_t n, int dtype, void *stream)
{
    if (!x || !gy || !gx) return DLKA_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    return DLKA_DISPATCH(dtype, launch_gelu_bwd<float>((const float *)x, (const float *)gy, (float *)gx, (long)n, st),
                         launch_gelu_bwd<bf16_t>((const bf16_t *)x, (const bf16_t *)gy, (bf16_t *)gx, (long)n, st));
}
int dlka_mul_forward(const void *a, const void *b, void *y, int64_t n, int dtype, void *stream)
{
    if (!a || !b || !y) return DLKA_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    return DLKA_DISPATCH(dtype, launch_mul_fwd<float>((const float *)a, (const float *)b, (float *)y, (long)n, st),
                         launch_mul_fwd<bf16_t>((const bf16_t *)a, (const bf16_t *)b, (bf16_t *)y, (long)n, st));
}
int dlka_mul_backward(const void *a, const void *b, const void *gy, void *ga, void *gb, int64_t n, int dtype, void *stream)
{
    if (!a || !b || !gy) return DLKA_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    return DLKA_DISPATCH(dtype, launch_mul_bwd<float>((const float *)a, (const float *)b, (const float *)gy, (float *)ga, (float *)gb, (long)n, st),
                         launch_mul_bwd<bf16_t>((const bf16_t *)a, (const bf16_t *)b, (const bf16_t *)gy, (bf16_t *)ga, (bf16_t *)gb, (long)n, st));
}
int dlka_add_forward(const void *a, const void *b, void *y, int64_t n, int dtype, void *stream)
{
    if (!a || !b || !y) return DLKA_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    return DLKA_DISPATCH(dtype, launch_add_fwd<float>((const float *)a, (const float *)b, (float *)y, (long)n, st),
                         launch_add_fwd<bf16_t>((const bf16_t *)a, (const bf16_t *)b, (bf16_t *)y, (long)n, st));
}

// ---- blocks -----------------------------------------------------------------------------------------------------
static int check_block(int B, int C, int D, int H, int W)
{
    if (B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return DLKA_ERR_SHAPE;
    if ((long)D * H * W > (1l << 30) || (long)B * (C > 98 ? C : 98) > (1l << 24)) return DLKA_ERR_SHAPE;
    return DLKA_OK;
}

size_t dlka_lka3d_saved_bytes(int B, int C, int D, int H, int W, int dtype)
{
    if (check_block(B, C, D, H, W)) return 0;
    Lka3dGeoms G(B, C, D, H, W);
    return 6 * align256(G.E * esz(dtype)) + align256(G.Off * esz(dtype));
}

size_t dlka_lka3d_workspace_bytes(int B, int C, int D, int H, int W, int dtype)
{
    if (check_block(B, C, D, H, W)) return 0;
    Lka3dGeoms G(B, C, D, H, W);
    size_t n = align256(G.scratch_floats() * 4) + 4 * align256(G.E * esz(dtype)) + align256(G.Off * esz(dtype));
    if (dtype != DLKA_F32) n += align256(G.max_weight_elems() * 4) + align256(G.E * 4);
    return n;
}

int dlka_lka3d_attention_forward(const void *x, const dlka_lka3d_params *p, void *y, void *saved, size_t saved_bytes,
                                 void *workspace, size_t workspace_bytes, int B, int C, int D, int H, int W, int dtype, void *stream)
{
    if (!x || !p || !y || !saved || !workspace) return DLKA_ERR_NULL;
    const void *const *pp = (const void *const *)p;
    for (size_t i = 0; i < sizeof(*p) / sizeof(void *); ++i) if (!pp[i]) return DLKA_ERR_NULL;
    DLKA_TRY(check_block(B, C, D, H, W));
    hipStream_t st = (hipStream_t)stream;
    return DLKA_DISPATCH(dtype, lka3d_forward_t<float>(x, p, y, saved, saved_bytes, workspace, workspace_bytes, B, C, D, H, W, st),
                         lka3d_forward_t<bf16_t>(x, p, y, saved, saved_bytes, workspace, workspace_bytes, B, C, D, H, W, st));
}

int dlka_lka3d_attention_backward(const void *x, const dlka_lka3d_params *p, const void *grad_y, const void *saved, size_t saved_bytes,
                                  void *grad_x, const dlka_lka3d_grads *grads, void *workspace, size_t workspace_bytes,
                                  int B, int C, int D, int H, int W, int dtype, void *stream)
{
    if (!x || !p || !grad_y || !saved || !grad_x || !grads || !workspace) return DLKA_ERR_NULL;
    const void *const *pp = (const void *const *)p;
    for (size_t i = 0; i < sizeof(*p) / sizeof(void *); ++i) if (!pp[i]) return DLKA_ERR_NULL;
    void *const *gp = (void *const *)grads;
    for (size_t i = 0; i < sizeof(*grads) / sizeof(void *); ++i) if (!gp[i]) return DLKA_ERR_NULL;
    DLKA_TRY(check_block(B, C, D, H, W));
    hipStream_t st = (hipStream_t)stream;
    return DLKA_DISPATCH(dtype,
                         lka3d_backward_t<float>(x, p, grad_y, saved, saved_bytes, grad_x, grads, workspace, workspace_bytes, B, C, D, H, W, dtype, st),
                         lka3d_backward_t<bf16_t>(x, p, grad_y, saved, saved_bytes, grad_x, grads, workspace, workspace_bytes, B, C, D, H, W, dtype, st));
}

// Which path a 2-D block call takes: the channels-last fast path wherever the width allows, unless the general NCHW kernels are forced.  ONE
// process-wide switch (initialised once from DLKA_LKA2D_GENERAL, changed only through dlka_lka2d_force_general), read by the size queries, the
// forward and the backward call alike — so a forward / backward pair and their buffer sizes cannot disagree because the environment changed in
// between.  Buffer sizes are the maximum over both paths: a switch between two calls never under-sizes a buffer.
static int g_lka2d_general = -1;
static bool lka2d_general()
{
    if (g_lka2d_general < 0) g_lka2d_general = getenv("DLKA_LKA2D_GENERAL") != nullptr ? 1 : 0;
    return g_lka2d_general != 0;
}
int dlka_lka2d_force_general(int on)
{
    const int old = lka2d_general() ? 1 : 0;
    g_lka2d_general = on ? 1 : 0;
    return old;
}

size_t dlka_lka2d_saved_bytes(int B, int C, int H, int W, int dtype)
{
    if (check_block(B, C, 1, H, W)) return 0;
    Lka2dGeoms G(B, C, H, W);
    size_t n = 5 * align256(G.E * esz(dtype)) + align256(G.Off5 * esz(dtype)) + align256(G.Off7 * esz(dtype));
    if (lka2d_cl_supported(B, C, H, W, dtype)) n = std::max(n, lka2d_cl_saved_bytes(B, C, H, W, dtype));   // channels-last fast path (dlka_capi_cl.hip)
    return n;
}

size_t dlka_lka2d_workspace_bytes(int B, int C, int H, int W, int dtype)
{
    if (check_block(B, C, 1, H, W)) return 0;
    Lka2dGeoms G(B, C, H, W);
    size_t n = align256(G.scratch_floats() * 4) + 4 * align256(G.E * esz(dtype)) + align256(G.Off7 * esz(dtype));
    if (dtype != DLKA_F32) n += align256(G.max_weight_elems() * 4) + align256(G.E * 4);
    if (lka2d_cl_supported(B, C, H, W, dtype)) n = std::max(n, lka2d_cl_workspace_bytes(B, C, H, W, dtype));
    return n;
}

int dlka_lka2d_attention_forward(const void *x, const dlka_lka2d_params *p, void *y, void *saved, size_t saved_bytes,
                                 void *workspace, size_t workspace_bytes, int B, int C, int H, int W, int dtype, void *stream)
{
    if (!x || !p || !y || !saved || !workspace) return DLKA_ERR_NULL;
    const void *const *pp = (const void *const *)p;
    for (size_t i = 0; i < sizeof(*p) / sizeof(void *); ++i) if (!pp[i]) return DLKA_ERR_NULL;
    DLKA_TRY(check_block(B, C, 1, H, W));
    hipStream_t st = (hipStream_t)stream;
    // channels-last fast path (MFMA offset nets, gather-layout depthwise deformable convs) wherever the width allows
    if (lka2d_cl_supported(B, C, H, W, dtype) && !lka2d_general())
        return lka2d_cl_forward(x, p, y, saved, saved_bytes, workspace, workspace_bytes, B, C, H, W, dtype, st);
    if (dtype == DLKA_BF16) return DLKA_ERR_UNSUPPORTED;   // DLKA_BF16 = bf16 activations with FP32 parameters: the channels-last path only
    return DLKA_DISPATCH(dtype, lka2d_forward_t<float>(x, p, y, saved, saved_bytes, workspace, workspace_bytes, B, C, H, W, st),
                         lka2d_forward_t<bf16_t>(x, p, y, saved, saved_bytes, workspace, workspace_bytes, B, C, H, W, st));
}

int dlka_lka2d_saved_offsets(int B, int C, int H, int W, int dtype, size_t byte_offsets[2], int *elem_bytes)
{
    if (!byte_offsets || !elem_bytes) return DLKA_ERR_NULL;
    DLKA_TRY(check_block(B, C, 1, H, W));
    if (lka2d_cl_supported(B, C, H, W, dtype) && !lka2d_general())
        return lka2d_cl_saved_offsets(B, C, H, W, dtype, byte_offsets, elem_bytes);
    if (dtype == DLKA_BF16) return DLKA_ERR_UNSUPPORTED;
    // lka2d_forward_t: h, a, o5, t1, o7, ...
    Lka2dGeoms G(B, C, H, W);
    const size_t e = esz(dtype);
    byte_offsets[0] = 2 * align256(G.E * e);
    byte_offsets[1] = 3 * align256(G.E * e) + align256(G.Off5 * e);
    *elem_bytes = (int)e;
    return DLKA_OK;
}

int dlka_lka2d_attention_backward(const void *x, const dlka_lka2d_params *p, const void *grad_y, const void *saved, size_t saved_bytes,
                                  void *grad_x, const dlka_lka2d_grads *grads, void *workspace, size_t workspace_bytes,
                                  int B, int C, int H, int W, int dtype, void *stream)
{
    if (!x || !p || !grad_y || !saved || !grad_x || !grads || !workspace) return DLKA_ERR_NULL;
    const void *const *pp = (const void *const *)p;
    for (size_t i = 0; i < sizeof(*p) / sizeof(void *); ++i) if (!pp[i]) return DLKA_ERR_NULL;
    void *const *gp = (void *const *)grads;
    for (size_t i = 0; i < sizeof(*grads) / sizeof(void *); ++i) if (!gp[i]) return DLKA_ERR_NULL;
    DLKA_TRY(check_block(B, C, 1, H, W));
    hipStream_t st = (hipStream_t)stream;
    if (lka2d_cl_supported(B, C, H, W, dtype) && !lka2d_general())
        return lka2d_cl_backward(x, p, grad_y, saved, saved_bytes, grad_x, grads, workspace, workspace_bytes, B, C, H, W, dtype, st);
    if (dtype == DLKA_BF16) return DLKA_ERR_UNSUPPORTED;
    return DLKA_DISPATCH(dtype,
                         lka2d_backward_t<float>(x, p, grad_y, saved, saved_bytes, grad_x, grads, workspace, workspace_bytes, B, C, H, W, dtype, st),
                         lka2d_backward_t<bf16_t>(x, p, grad_y, saved, saved_bytes, grad_x, grads, workspace, workspace_bytes, B, C, H, W, dtype, st));
}

}  // extern "C"
