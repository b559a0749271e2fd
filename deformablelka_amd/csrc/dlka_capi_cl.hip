// C-ABI entry points of the channels-last (NDHWC / token layout) fast path: stride-1 same-size convolutions on the
// matrix cores, register-tiled depthwise convs, the fused deformable backward, and the token-layout D-LKA block.
// Everything here is fp32; shapes the fast path does not cover return DLKA_ERR_UNSUPPORTED and the caller uses the
// general NCDHW entry points (dlka_capi.hip) instead — still HIP, never a CPU fallback.
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "cl_args.h"
#include "dlka_kernels.h"

using namespace dlka;

namespace {

inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

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
    // optional area: nullptr (and the carver stays valid) when `want` is false or the buffer has no room for it
    void *take_opt(size_t n, bool want)
    {
        n = align256(n);
        if (!want || !base || used > cap || used + n > cap) return nullptr;
        void *r = base + used;
        used += n;
        return r;
    }
    bool ok() const { return used <= cap; }
};

#define DLKA_TRY(expr)                  \
    do {                                \
        int rc_ = (expr);               \
        if (rc_ != DLKA_OK) return rc_; \
    } while (0)

// a stride-1, same-size convolution in channels-last layout
struct SameConv {
    int B, D, H, W, N, M, Cin, Cout, group;
    int kd, kh, kw, pd, ph, pw, dd, dh, dw, K;
    int act_bf16;   // 1: channels-last activation tensors are bf16 storage (DLKA_BF16 token path); planar offsets / grad_offset stay fp32
};

int make_same_conv(const dlka_conv_geom *c, SameConv &s)
{
    if (!c) return DLKA_ERR_NULL;
    if (c->B <= 0 || c->C <= 0 || c->D <= 0 || c->H <= 0 || c->W <= 0 || c->Cout <= 0) return DLKA_ERR_SHAPE;
    if (c->kd <= 0 || c->kh <= 0 || c->kw <= 0 || c->dd <= 0 || c->dh <= 0 || c->dw <= 0) return DLKA_ERR_SHAPE;
    if (c->group <= 0 || c->C % c->group || c->Cout % c->group) return DLKA_ERR_GROUP;
    if (c->sd != 1 || c->sh != 1 || c->sw != 1) return DLKA_ERR_UNSUPPORTED;
    if (dlka_conv_out_size(c->D, c->pd, c->dd, c->kd, 1) != c->D || dlka_conv_out_size(c->H, c->ph, c->dh, c->kh, 1) != c->H ||
        dlka_conv_out_size(c->W, c->pw, c->dw, c->kw, 1) != c->W)
        return DLKA_ERR_UNSUPPORTED;
    const long N = (long)c->D * c->H * c->W;
    if (N > (1l << 30) || (long)c->B * N > (1l << 30)) return DLKA_ERR_SHAPE;
    s.B = c->B; s.D = c->D; s.H = c->H; s.W = c->W; s.N = (int)N; s.M = (int)(c->B * N);
    s.Cin = c->C; s.Cout = c->Cout; s.group = c->group;
    s.kd = c->kd; s.kh = c->kh; s.kw = c->kw; s.pd = c->pd; s.ph = c->ph; s.pw = c->pw;
    s.dd = c->dd; s.dh = c->dh; s.dw = c->dw; s.K = c->kd * c->kh * c->kw;
    s.act_bf16 = 0;
    return DLKA_OK;
}

bool nt_ok(int np) { const int nt = np / 32; return nt == 1 || nt == 2 || nt == 3 || nt == 4 || nt == 8; }
bool is_depthwise(const SameConv &s) { return s.group == s.Cin && s.Cin == s.Cout; }
bool dw_supported(const SameConv &s)
{
    const bool kshape = (s.kw == 5 && s.dw == 1) || (s.kw == 7 && s.dw == 3) || (s.kw == 3 && s.dw == 1) || (s.kw == 5 && s.dw == 3) ||
                        (s.kw == 7 && s.dw == 1);
    const int cpb = s.Cin < 256 ? s.Cin : 256;
    return kshape && s.Cin % 32 == 0 && 256 % cpb == 0 && s.Cin % cpb == 0;
}
bool dense_fwd_supported(const SameConv &s) { return s.group == 1 && s.Cin % 32 == 0 && nt_ok(round_up(s.Cout, 32)); }

// Contractions with K > 1 taps (the offset-predict conv, its data gradient and its weight gradient) are MFMA-bound with fp32
// inputs; they run on the bf16 matrix cores with split operands and fp32 accumulation (cl_igemm.hip) unless DLKA_EXACT_FP32=1.
// Returns the number of bf16 terms per operand (0 = exact fp32-input MFMA):
//   gradient contractions: 2 (three products, ~1e-5 relative);
//   FORWARD offset conv:   3 (six products, fp32-equivalent).  Its output decides floor() of every sampling position: a 1e-5
//     perturbation flips the cell of the samples that sit within 1e-5 of an integer, and each flip changes that sample's
//     grad_offset by O(1) (seen as 1.5e-2 on conv_offset.weight.grad with offsets concentrated near 0), so the two-term split is
//     not used there by default; DLKA_SPLIT_FORWARD=2 forces it for A/B runs;
//   bf16 activations (DLKA_BF16): the activation is its own high term, weights are split in two, no a_lo products.
int use_split(const SameConv &s, bool forward)
{
    static const bool exact = getenv("DLKA_EXACT_FP32") != nullptr;
    static const int fwd = getenv("DLKA_SPLIT_FORWARD") ? atoi(getenv("DLKA_SPLIT_FORWARD")) : 3;
    if (s.act_bf16) return s.K > 1 ? 2 : 0;   // bf16 activations are their own high term: two-term weights, no a_lo products (cl_igemm.hip)
    if (exact || s.K <= 1) return 0;
    if (!forward) return 2;
    return (fwd == 2 || fwd == 3) ? fwd : 0;
}
inline int split_mode_flag(int terms) { return terms == 3 ? 16 : (terms == 2 ? 8 : 0); }

void fill_igemm(IgemmArgs &a, const SameConv &s)
{
    memset(&a, 0, sizeof(a));
    a.B = s.B; a.D = s.D; a.H = s.H; a.W = s.W; a.N = s.N; a.M = s.M;
    a.kd = s.kd; a.kh = s.kh; a.kw = s.kw; a.pd = s.pd; a.ph = s.ph; a.pw = s.pw; a.dd = s.dd; a.dh = s.dh; a.dw = s.dw; a.K = s.K;
    a.act_bf16 = s.act_bf16;
}

// ---- dense conv forward: out = conv(x) (+ epilogue) ---------------------------------------------------------------
// wp must hold K * Cin * round_up(Cout,32) floats
// Tap splits of a launch whose partial sums meet in fp32 atomics on a zero-filled output (> 1: the caller zero-fills, and bf16 storage goes through an fp32 staging
// buffer).  Round 6: the split-operand convs (K > 1) of small volumes split their contraction over the waves of a workgroup instead (cl_conv_kw.hip: deterministic,
// no atomics) — for those this returns 1.  The deformable conv's forward keeps its own query (deform_forward_splits).
int deform_forward_splits(const SameConv &s) { return cl_igemm_pick_splits(s.M, s.K * (s.Cin / 32), 0, s.K); }
int dense_forward_splits(const SameConv &s, int epi)
{
    const int sp = cl_igemm_pick_splits(s.M, s.K * (s.Cin / 32), epi, s.K);
    if (sp > 1 && cl_conv_kw_applies(0, s.act_bf16 ? 1 : 0, use_split(s, true), s.K, epi, round_up(s.Cout, 32), s.act_bf16 != 0, s.D > 1)) return 1;
    return sp;
}
int dense_backward_data_splits(const SameConv &s, int epi, int gout_planar = -1)
{
    const int sp = cl_igemm_pick_splits(s.M, s.K * (round_up(s.Cout, 32) / 32), epi, s.K);
    // (gout_planar < 0: a query without the layout — bf16 storage reaches here with planar gradients only, fp32 with either)
    const int amode = gout_planar < 0 ? (s.act_bf16 ? 2 : 0) : (gout_planar ? 2 : 0);
    if (sp > 1 && cl_conv_kw_applies(amode, 0, use_split(s, false), s.K, epi, s.Cin, s.act_bf16 != 0, s.D > 1)) return 1;
    return sp;
}

// zeroed: the caller has zero-filled `out` (needed when the tap split is > 1; one batched fill per block instead of one per conv)
// ride: zero fills that go out with this launch (pointwise kernel; any other kernel gets them as a launch of their own, cl_igemm.hip)
int dense_forward(const SameConv &s, const float *x, const float *w, const float *bias, float *out, int out_planar, float *wp,
                  int epi, const float *aux, float *out2, hipStream_t st, bool zeroed = false, const ZeroBatch *ride = nullptr, float *out2_f32 = nullptr)
{
    const int NP = round_up(s.Cout, 32);
    const int split = use_split(s, true);
    if (w) DLKA_TRY(launch_cl_prep_weight(w, wp, s.Cout, s.Cin, s.K, s.Cin, NP, split_mode_flag(split), st));   // w == null: wp already prepared
    IgemmArgs a;
    fill_igemm(a, s);
    a.split_bf16 = split;
    a.in = x; a.wp = wp; a.bias = bias; a.out = out; a.out2 = out2; a.aux = aux; a.epi = epi; a.out_zeroed = zeroed ? 1 : 0;
    a.Cin = s.Cin; a.CinReal = s.Cin; a.CinP = s.Cin; a.Cout = s.Cout; a.NP = NP;
    if (ride) a.zero = *ride;
    int splits = dense_forward_splits(s, epi);
    if (out_planar && cl_conv_brick3_supported(a)) splits = 1;   // (cl_conv_brick.hip writes every output itself)
    if (out2_f32) {   // only the pointwise kernel's bf16 GELU epilogue carries the fp32 side output
        if (!(s.act_bf16 && epi == 1 && s.K == 1 && splits == 1 && !split && !out_planar)) return DLKA_ERR_UNSUPPORTED;
        a.out2_f32 = out2_f32;
    }
    return launch_cl_igemm(0, out_planar ? 1 : 0, a, splits, st);
}

// ---- dense conv data gradient: gx = conv_transpose(gout) (+ epilogue) ---------------------------------------------
// wp must hold K * round_up(Cout,32) * Cin floats.  gout channels-last needs Cout % 32 == 0; planar any Cout.
// bf16 storage: `aux_f32` says the epilogue operand is fp32 all the same; `acc32` (fp32 [M][Cin], ZEROED by the caller) receives split
// partial sums and is converted into gx afterwards
int dense_backward_data(const SameConv &s, const float *gout, int gout_planar, const float *w, float *gx, float *wp, int epi,
                        const float *aux, hipStream_t st, const float *aux2 = nullptr, float *out2 = nullptr, bool zeroed = false, bool g_packed = false,
                        bool aux_f32 = false, float *acc32 = nullptr, const ZeroBatch *ride = nullptr)
{
    const int KP = round_up(s.Cout, 32), NP = s.Cin;
    if (!(nt_ok(NP) || NP == 192 || NP == 384) || s.Cin % 32) return DLKA_ERR_UNSUPPORTED;   // (192 / 384: the 2-D block's widths, 3 / 4 column tiles per workgroup)
    if (!gout_planar && s.Cout % 32) return DLKA_ERR_UNSUPPORTED;
    const int split = use_split(s, false);
    if (w) DLKA_TRY(launch_cl_prep_weight(w, wp, s.Cout, s.Cin, s.K, KP, NP, 1 | split_mode_flag(split), st));
    IgemmArgs a;
    fill_igemm(a, s);
    a.split_bf16 = split;
    a.pd = s.dd * (s.kd - 1) - s.pd; a.ph = s.dh * (s.kh - 1) - s.ph; a.pw = s.dw * (s.kw - 1) - s.pw;
    a.in = gout; a.wp = wp; a.bias = nullptr; a.out = gx; a.aux = aux; a.aux2 = aux2; a.out2 = out2; a.epi = epi; a.out_zeroed = zeroed ? 1 : 0;
    a.Cin = s.Cout; a.CinReal = s.Cout; a.CinP = KP; a.Cout = s.Cin; a.NP = NP;
    if (g_packed) {   // gout = pack_split2() words, KP zero-padded channel planes per batch (DeformBwdArgs::goff_cpad)
        if (!gout_planar || split != 2) return DLKA_ERR_UNSUPPORTED;
        a.a_packed = 1; a.CinReal = KP;
    }
    a.aux_f32 = aux_f32 ? 1 : 0;
    if (ride) a.zero = *ride;
    int splits = dense_backward_data_splits(s, epi, gout_planar);
    // cl_conv_brick.hip: no tap split; volumes too small for a workgroup per tile split the plane chunks instead (fp32 atomics into a zeroed buffer, like a tap split)
    const int bsplit = gout_planar ? cl_conv_brick_split(a) : 0;
    bool brick_only = false;   // the chunk split alone made this an accumulating launch: the caller's zero-fill decision (dense_backward_data_splits) does not know
    if (bsplit == 1) splits = 1;
    else if (bsplit > 1 && splits <= 1) { splits = 2; brick_only = true; }   // (only its being > 1 matters below: the zero-fill / fp32-accumulation route)
    if (s.act_bf16 && splits > 1) {
        if (!acc32) return DLKA_ERR_WORKSPACE;
        if (brick_only) DLKA_TRY(launch_zero(acc32, (size_t)s.M * s.Cin * 4, st));
        a.out = acc32; a.out_zeroed = 1;
        DLKA_TRY(launch_cl_igemm(gout_planar ? 2 : 0, 0, a, splits, st));
        return launch_cast_from_f32<bf16_t>(acc32, reinterpret_cast<bf16_t *>(gx), (long)s.M * s.Cin, st);
    }
    if (brick_only) {   // fp32 atomics into `gx`: zero it here, whatever the caller said
        DLKA_TRY(launch_zero(gx, (size_t)s.M * s.Cin * 4, st));
        a.out_zeroed = 1;
    }
    return launch_cl_igemm(gout_planar ? 2 : 0, 0, a, splits, st);
}

// (x 3/2 for K > 1: the three-term bf16 layout of the forward weights takes 48 instead of 32 floats per unit and column)
size_t dense_wp_floats(const SameConv &s) { return (size_t)s.K * round_up(s.Cin, 32) * round_up(s.Cout, 32) * (s.K > 1 ? 3 : 2) / 2; }

// ---- environment switches, read once (dlka_env_refresh() re-reads) ------------------------------------------------
struct ForkEnv {
    bool gx_rows_set;
    long gx_rows;       // DLKA_GX_FORK_MIN_ROWS
    int lka2d_fork;     // DLKA_LKA2D_FORK: 0 = one stream, 1 = the offset nets' weight gradients only, 2 (unset) = + grad_input beside grad_offset
    // A/B switches that used to be read with getenv() on every call (getenv racing with os.environ writes of other Python threads is undefined behaviour, and a
    // workspace-size query could disagree with the carve that follows it).  Tests that toggle them call dlka_env_refresh().
    bool wgrad_pad;     // DLKA_WGRAD_PAD != 0: the dense weight gradient from a zero-padded copy (cl_wgrad.hip)
    bool dwpair;        // DLKA_DWPAIR != 0: both depthwise convs of a small volume in one launch (cl_dwpair.hip)
    bool prep_tiled;    // DLKA_PREP_TILED != 0: weight preparation tile by tile through LDS (cl_igemm.hip prep_job_tile)
};
static std::mutex g_fork_env_mu;
static ForkEnv g_fork_env;
static std::atomic<int> g_fork_env_loaded{0};
static void fork_env_load()
{
    std::lock_guard<std::mutex> lk(g_fork_env_mu);
    ForkEnv e;
    const char *r = getenv("DLKA_GX_FORK_MIN_ROWS");
    e.gx_rows_set = r != nullptr;
    e.gx_rows = r ? atol(r) : 0;
    const char *f = getenv("DLKA_LKA2D_FORK");
    e.lka2d_fork = !f ? 2 : f[0] == '0' ? 0 : f[0] == '1' ? 1 : 2;
    auto off = [](const char *name) { const char *v = getenv(name); return v && v[0] == '0'; };
    e.wgrad_pad = !off("DLKA_WGRAD_PAD");
    e.dwpair = !off("DLKA_DWPAIR");
    e.prep_tiled = !off("DLKA_PREP_TILED");
    g_fork_env = e;
    g_fork_env_loaded.store(1, std::memory_order_release);
}
static ForkEnv fork_env()
{
    if (!g_fork_env_loaded.load(std::memory_order_acquire)) fork_env_load();
    std::lock_guard<std::mutex> lk(g_fork_env_mu);
    return g_fork_env;
}

// ---- dense conv weight gradient -----------------------------------------------------------------------------------
// padbuf (optional, dense_wgrad_pad_bytes(s) bytes): scratch for the zero-padded copy of x — selects the padded kernels (cl_wgrad.hip, round 5) where they apply
size_t dense_wgrad_pad_bytes(const SameConv &s)
{
    // (DLKA_WGRAD_PAD=0: the unpadded kernels of rounds 2 - 4.  A workspace sized with the padded copy and used without it, or the other way round, is safe — the optional
    //  carve returns null when there is no room, and null selects the unpadded kernels)
    if (!fork_env().wgrad_pad || s.K <= 1 || s.group != 1) return 0;
    const size_t n = cl_wgrad_pad_bytes(s.B, s.D, s.H, s.W, s.Cin, s.kd, s.kh, s.kw, s.dd, s.dh, s.dw, s.act_bf16);
    return n < ((size_t)1 << 31) ? align256(n) : 0;
}
int dense_backward_weight(const SameConv &s, const float *x, const float *gout, int gout_planar, float *gw, float *gb, float *part, hipStream_t st,
                          FinalizeJob *defer = nullptr, int g_cpad = 0, float *padbuf = nullptr)
{
    if (s.Cin % 32) return DLKA_ERR_UNSUPPORTED;
    if (s.K != 1 && s.K > 7 * 64) return DLKA_ERR_UNSUPPORTED;
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.g = gout; a.in = x; a.part = part;
    a.B = s.B; a.D = s.D; a.H = s.H; a.W = s.W; a.N = s.N; a.M = s.M; a.Cin = s.Cin; a.Cout = s.Cout;
    a.kd = s.kd; a.kh = s.kh; a.kw = s.kw; a.pd = s.pd; a.ph = s.ph; a.pw = s.pw; a.dd = s.dd; a.dh = s.dh; a.dw = s.dw; a.K = s.K;
    if (s.K == 1 && gout_planar) return DLKA_ERR_UNSUPPORTED;
    if (g_cpad && (!gout_planar || s.K == 1)) return DLKA_ERR_UNSUPPORTED;
    a.g_cpad = g_cpad;
    a.act_bf16 = s.act_bf16;
    a.pad = (padbuf && gout_planar && !g_cpad && dense_wgrad_pad_bytes(s)) ? padbuf : nullptr;
    return launch_cl_wgrad<float>(0, gout_planar ? 1 : 0, a, gw, gb, st, defer);
}

void fill_pw_wgrad(WgradArgs &a, const SameConv &s, const float *x, const float *gout, float *part)
{
    memset(&a, 0, sizeof(a));
    a.g = gout; a.in = x; a.part = part;
    a.B = s.B; a.D = s.D; a.H = s.H; a.W = s.W; a.N = s.N; a.M = s.M; a.Cin = s.Cin; a.Cout = s.Cout;
    a.kd = s.kd; a.kh = s.kh; a.kw = s.kw; a.pd = s.pd; a.ph = s.ph; a.pw = s.pw; a.dd = s.dd; a.dh = s.dh; a.dw = s.dw; a.K = s.K;
    a.act_bf16 = s.act_bf16;
}

// ---- depthwise ------------------------------------------------------------------------------------------------------
// The LDS-brick kernels (cl_dwconv_lds.hip) read a class-blocked fp32 copy of the input: `blk` = scratch for it (null: those kernels are never taken),
// `chain` = the depthwise conv that consumes THIS conv's output next (null: none) with `chain_blk` = scratch for ITS blocked input — when both convs take
// the LDS-brick kernel this one's epilogue writes that copy, and *chained tells the caller to pass in_blocked = true to the next call.
struct DwBlk {
    float *blk = nullptr;
    size_t blk_floats = 0;
    bool in_blocked = false;
    const SameConv *chain = nullptr;
    float *chain_blk = nullptr;
    bool *chained = nullptr;
};

static void fill_dw_args(DwArgs &a, const SameConv &s, int flip)
{
    memset(&a, 0, sizeof(a));
    a.B = s.B; a.D = s.D; a.H = s.H; a.W = s.W; a.C = s.Cin;
    a.act_bf16 = s.act_bf16; a.xcd_nx = 0;
    a.kd = s.kd; a.kh = s.kh; a.dd = s.dd; a.dh = s.dh;
    if (flip) { a.pd = s.dd * (s.kd - 1) - s.pd; a.ph = s.dh * (s.kh - 1) - s.ph; a.pw = s.dw * (s.kw - 1) - s.pw; }
    else { a.pd = s.pd; a.ph = s.ph; a.pw = s.pw; }
}

int dw_forward(const SameConv &s, const float *x, const float *w, const float *bias, float *out, float *wp, int flip, hipStream_t st,
               const float *gelu_x = nullptr, const float *gelu_add = nullptr, float *out_lo = nullptr, const DwBlk *bk = nullptr)
{
    if (w) DLKA_TRY(launch_cl_dw_prep_weight(w, wp, s.Cin, s.K, flip, st));
    if (out_lo && s.act_bf16) return DLKA_ERR_UNSUPPORTED;   // (the bf16 copy rides in the fp32 kernels only)
    DwArgs a;
    fill_dw_args(a, s, flip);
    a.in = x; a.wp = wp; a.bias = bias; a.out = out; a.out_lo = out_lo; a.gelu_x = gelu_x; a.gelu_add = gelu_add;
    if (bk && bk->blk) {
        a.blk = bk->blk; a.blk_floats = bk->blk_floats; a.in_blocked = bk->in_blocked ? 1 : 0;
        if (bk->chained) *bk->chained = false;
        if (bk->chain && bk->chain_blk && cl_dwconv_lds_selected(a, s.kw, s.dw)) {
            DwArgs n;
            fill_dw_args(n, *bk->chain, flip);
            n.blk = bk->chain_blk; n.blk_floats = bk->blk_floats;
            if (cl_dwconv_lds_selected(n, bk->chain->kw, bk->chain->dw)) {
                a.out_blk = bk->chain_blk; a.out_blk_dil = bk->chain->dw;
                if (bk->chained) *bk->chained = true;
            }
        }
    }
    return launch_cl_dwconv(a, s.kw, s.dw, st);
}

// Two chained depthwise convs of a small volume in ONE launch (cl_dwpair.hip): x -> sa -> outA -> sb -> outB, prepared weights wpA / wpB (already in the wanted form: the
// flipped one for the data gradients — "same" padding is its own mirror).  DLKA_ERR_UNSUPPORTED: not that shape (or DLKA_DWPAIR=0) — the caller runs them one by one.
static int dwpair_mode()
{
    return fork_env().dwpair ? 1 : 0;
}

int dw_pair(const SameConv &sa, const SameConv &sb, const float *x, const float *wpA, const float *biasA, float *outA, float *outA_lo, const float *wpB, const float *biasB,
            float *outB, float *outB_lo, const float *gelu_x, const float *gelu_add, hipStream_t st)
{
    if (!dwpair_mode()) return DLKA_ERR_UNSUPPORTED;
    auto cubic = [](const SameConv &s) { return s.kd == s.kw && s.kh == s.kw && s.dd == s.dw && s.dh == s.dw && s.pd == s.pw && s.ph == s.pw; };
    if (!cubic(sa) || !cubic(sb) || sa.act_bf16 != sb.act_bf16 || sa.Cin != sb.Cin) return DLKA_ERR_UNSUPPORTED;
    DwPairArgs a;
    memset(&a, 0, sizeof(a));
    a.in = x; a.wpA = wpA; a.wpB = wpB; a.biasA = biasA; a.biasB = biasB; a.outA = outA; a.outB = outB; a.outA_lo = outA_lo; a.outB_lo = outB_lo;
    a.gelu_x = gelu_x; a.gelu_add = gelu_add;
    a.B = sa.B; a.D = sa.D; a.H = sa.H; a.W = sa.W; a.C = sa.Cin;
    a.kA = sa.kw; a.dA = sa.dw; a.pA = sa.pw; a.KA = sa.K;
    a.kB = sb.kw; a.dB = sb.dw; a.pB = sb.pw; a.KB = sb.K;
    a.act_bf16 = sa.act_bf16;
    return launch_cl_dwpair_small(a, st);
}

// defer != null: gwp is a zeroed [K + 1][C] staging area (row K collects the bias sums) that the caller's fused
// finalisation kernel re-lays into gw / gb
int dw_backward_weight(const SameConv &s, const float *x, const float *gout, float *gw, float *gb, float *gwp, hipStream_t st, FinalizeJob *defer = nullptr)
{
    DwWgradArgs a;
    memset(&a, 0, sizeof(a));
    a.g = gout; a.in = x; a.gwp = gwp; a.gb = defer ? gwp + (size_t)s.K * s.Cin : gb;
    a.B = s.B; a.D = s.D; a.H = s.H; a.W = s.W; a.C = s.Cin;
    a.kd = s.kd; a.kh = s.kh; a.pd = s.pd; a.ph = s.ph; a.pw = s.pw; a.dd = s.dd; a.dh = s.dh;
    a.act_bf16 = s.act_bf16;
    if (defer) {
        DLKA_TRY(launch_cl_dwconv_wgrad(a, s.kw, s.dw, st, false));
        memset(defer, 0, sizeof(*defer));
        defer->part = gwp; defer->gw = gw; defer->gb = gb; defer->K = s.K; defer->Cin = s.Cin; defer->kind = 1; defer->chunks = 1;
        defer->n = (long)s.K * s.Cin + s.Cin;
        return DLKA_OK;
    }
    DLKA_TRY(launch_cl_dwconv_wgrad(a, s.kw, s.dw, st));
    return launch_cl_dw_unprep<float>(gwp, gw, s.Cin, s.K, st);
}

// ---- deformable (groups = deformable_groups = 1) ---------------------------------------------------------------------
// The deformable conv's contractions on the bf16 matrix cores (round 4): with DLKA_BF16 activations all of them (forward, Col of grad_offset / grad_input); with
// fp32 activations the two BACKWARD ones, grad_out split in two bf16 terms (fp32-equivalent to 1e-5; the forward pass keeps the exact fp32-input MFMA).  ONE
// process-wide switch, read once: DLKA_DEFORM_B16=0 keeps the fp32-input MFMA of rounds 2 - 3 everywhere (A/B runs).  It decides the layout of the prepared weights AND the kernel that reads them, so it must not change between
// a weight preparation and its use — hence cached.
static bool deform_b16()
{
    static const bool on = [] { const char *e = getenv("DLKA_DEFORM_B16"); return !(e && e[0] == '0') && getenv("DLKA_EXACT_FP32") == nullptr; }();   // (DLKA_EXACT_FP32: every contraction on the fp32-input MFMA)
    return on;
}

bool deform_supported(const SameConv &s) { return s.group == 1 && s.Cin % 32 == 0 && s.Cout % 32 == 0 && nt_ok(s.Cout) && nt_ok(s.Cin); }

// Small volumes split the taps over the grid: the tap ranges' partial tiles go to `slab` (fp32 [splits][M][Cout], deform_fwd_slab_floats(s) floats, every element written)
// and are summed IN SLAB ORDER by one reduce launch — deterministic, like the reference's im2col + addmm (deform_conv_cuda.cu:95-123); rounds 1 - 5 let them meet in fp32
// atomics on a zero-filled output (the order of arrival decided the last bit, and the bf16 path needed an fp32 landing zone + a cast launch: the reduce launch replaces it).
int deform_fwd_actual_splits(const SameConv &s) { return cl_deform_fwd_actual_splits(s.K, s.Cin, deform_forward_splits(s)); }
size_t deform_fwd_slab_floats(const SameConv &s)
{
    const int sp = deform_fwd_actual_splits(s);
    return sp > 1 ? (size_t)sp * s.M * s.Cout : 0;
}
int deform_forward(const SameConv &s, const float *x, const float *off, const float *w, const float *bias, float *out, float *wp, hipStream_t st,
                   float *slab = nullptr)
{
    // DLKA_BF16: the contraction runs on the bf16 matrix cores — weights as two-term bf16 records (prep mode | 8; deform_b16() = 0 keeps the fp32-input MFMA)
    const int b16 = (s.act_bf16 && deform_b16()) ? 1 : 0;
    if (w) DLKA_TRY(launch_cl_prep_weight(w, wp, s.Cout, s.Cin, s.K, s.Cin, s.Cout, b16 ? 8 : 0, st));
    IgemmArgs a;
    fill_igemm(a, s);
    a.split_bf16 = b16 ? 2 : 0;
    a.in = x; a.off = off; a.wp = wp; a.bias = bias; a.out = out; a.epi = 0; a.out_zeroed = 0;
    a.Cin = s.Cin; a.CinReal = s.Cin; a.CinP = s.Cin; a.Cout = s.Cout; a.NP = s.Cout;
    const int splits = deform_fwd_actual_splits(s);
    if (splits > 1) {
        if (!slab) return DLKA_ERR_WORKSPACE;
        a.out = slab;
        const int rc = launch_cl_deform_fwd(a, splits, st);
        if (rc == DLKA_OK) return launch_cl_slab_reduce(slab, splits, (long)s.M * s.Cout, out, s.act_bf16, st);
        if (rc != DLKA_ERR_UNSUPPORTED || s.act_bf16) return rc;
        a.out = out;   // (a width the gather kernels do not tile: the first-generation kernel, tap split with atomics on a zero fill of its own)
        return launch_cl_igemm(1, 0, a, splits, st);
    }
    const int rc = launch_cl_deform_fwd(a, 1, st);
    if (rc != DLKA_ERR_UNSUPPORTED || s.act_bf16) return rc;
    return launch_cl_igemm(1, 0, a, 1, st);
}

void fill_deform_bwd(DeformBwdArgs &a, const SameConv &s)
{
    memset(&a, 0, sizeof(a));
    a.B = s.B; a.D = s.D; a.H = s.H; a.W = s.W; a.N = s.N; a.M = s.M; a.C = s.Cin; a.Cout = s.Cout; a.CoutP = s.Cout;
    a.kd = s.kd; a.kh = s.kh; a.kw = s.kw; a.pd = s.pd; a.ph = s.ph; a.pw = s.pw; a.dd = s.dd; a.dh = s.dh; a.dw = s.dw; a.K = s.K;
    a.act_bf16 = s.act_bf16;
}

// scratch floats of the brick windows the grad_input scatter flushes (cl_deform_bwd2.hip)
size_t deform_scratch_floats(const SameConv &s)
{
    DeformBwdArgs a;
    fill_deform_bwd(a, s);
    return cl_deform_bwd2_scratch_floats(a);
}

// The stored-sample hand-over of the fp32 path keeps its samples as IEEE halves (round 6; DeformBwdArgs::samp_f16 has the reasoning and the error bound): one process-wide
// switch, read once — the grad_offset kernel that writes them and the weight-gradient kernel that reads them must agree.  DLKA_SAMP_F16=0 or DLKA_EXACT_FP32: fp32 samples.
static bool samp_f16(const SameConv &s)
{
    static const bool on = [] { const char *e = getenv("DLKA_SAMP_F16"); return !(e && e[0] == '0') && getenv("DLKA_EXACT_FP32") == nullptr; }();
    return on && !s.act_bf16;
}

int deform_bwd_variant() { return 0; }   // (two earlier generations — one fused kernel with global atomics, an fp32 LDS window — were removed in round 2)

int deform_backward(const SameConv &s, const float *x, const float *off, const float *w, const float *gout, float *gx, float *goff,
                    float *gw, float *gb, float *wp, float *part, float *scratch, hipStream_t st, FinalizeJob *defer = nullptr, bool gx_zeroed = false,
                    bool goff_zeroed = false, int goff_cpad = 0, float *samp = nullptr, const float *wp16 = nullptr)
{
    // samp ([K][M][C] fp32): a grad_offset call stores the trilinear samples there, a weight-gradient call reads them instead of gathering again
    if (gx || goff) {
        if (w) DLKA_TRY(launch_cl_prep_weight(w, wp, s.Cout, s.Cin, s.K, s.Cout, s.Cin, 2, st));
        DeformBwdArgs a;
        fill_deform_bwd(a, s);
        a.in = x; a.off = off; a.g = gout; a.wp = wp; a.gx = gx; a.goff = goff; a.gx_zeroed = gx_zeroed ? 1 : 0; a.goff_zeroed = goff_zeroed ? 1 : 0; a.goff_cpad = goff_cpad;
        a.samp = goff ? samp : nullptr;
        a.samp_f16 = (a.samp && samp_f16(s)) ? 1 : 0;
        a.wp16 = deform_b16() ? wp16 : nullptr;   // (prepared by the caller: two-term bf16 records, mode 2 | 8) — both dtypes: bf16 rows as they are, fp32 rows split
        DLKA_TRY(launch_cl_deform_bwd2(a, scratch, st));
    }
    if (gw) {
        WgradArgs a;
        memset(&a, 0, sizeof(a));
        a.g = gout; a.in = x; a.off = off; a.part = part; a.samp = samp;
        a.samp_f16 = (samp && samp_f16(s)) ? 1 : 0;
        {   // DLKA_SAMP_B16MFMA=0: the half samples widened onto fp32-input MFMAs (round 6's first form); read once
            static const bool b16 = [] { const char *e = getenv("DLKA_SAMP_B16MFMA"); return !(e && e[0] == '0'); }();
            a.samp_b16mfma = (a.samp_f16 && b16) ? 1 : 0;
        }
        a.B = s.B; a.D = s.D; a.H = s.H; a.W = s.W; a.N = s.N; a.M = s.M; a.Cin = s.Cin; a.Cout = s.Cout;
        a.kd = s.kd; a.kh = s.kh; a.kw = s.kw; a.pd = s.pd; a.ph = s.ph; a.pw = s.pw; a.dd = s.dd; a.dh = s.dh; a.dw = s.dw; a.K = s.K;
        a.act_bf16 = s.act_bf16;
        DLKA_TRY(launch_cl_wgrad<float>(1, 0, a, gw, gb, st, defer));
    } else if (gb) {
        if (s.act_bf16) return DLKA_ERR_UNSUPPORTED;
        DLKA_TRY(launch_cl_colsum(gout, gb, s.M, s.Cout, st));
    }
    return DLKA_OK;
}

// conv1 + gate -> proj_2 + shortcut (bwd = 0) / their data gradients (bwd = 1) as ONE launch where the pair kernel pays (C = 32).
// DLKA_PW_UNFUSED=1 keeps the two launches (A/B runs).  Returns DLKA_ERR_UNSUPPORTED when the caller has to issue the two convs itself.
int pointwise_pair(const SameConv &s, int bwd, const float *in, const float *wp1, const float *bias1, const float *wp2, const float *bias2,
                   const float *a, const float *b, float *out1, float *out1b, float *out2, hipStream_t st, const ZeroBatch *ride = nullptr)
{
    const bool unfused = getenv("DLKA_PW_UNFUSED") != nullptr;   // (not cached: a parity test toggles it)
    // (C = 64 exists in the kernel but loses: 26 us against 2 x 7.7 + 4.5 us at 16^3 — M / 32 = 256 waves are too few; profiles/archive/r03n notes)
    if (unfused || s.Cin != 32 || s.Cin != s.Cout || s.K != 1) return DLKA_ERR_UNSUPPORTED;
    PwPairArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.in = in; pa.wp1 = wp1; pa.bias1 = bias1; pa.wp2 = wp2; pa.bias2 = bias2; pa.a = a; pa.b = b;
    pa.out1 = out1; pa.out1b = out1b; pa.out2 = out2; pa.M = s.M; pa.C = s.Cin; pa.bwd = bwd; pa.act_bf16 = s.act_bf16;
    if (ride) pa.zero = *ride;
    return launch_cl_pointwise_pair(pa, st);
}

// ---- the token-layout 3-D block ----------------------------------------------------------------------------------------
SameConv block_conv(int B, int C, int Cout, int D, int H, int W, int k, int pad, int dil, int group, int act_bf16 = 0)
{
    SameConv s;
    s.act_bf16 = act_bf16;
    s.B = B; s.D = D; s.H = H; s.W = W; s.N = D * H * W; s.M = B * s.N; s.Cin = C; s.Cout = Cout; s.group = group;
    s.kd = s.kh = s.kw = k; s.pd = s.ph = s.pw = pad; s.dd = s.dh = s.dw = dil; s.K = k * k * k;
    return s;
}

// DLKA_BF16 is MIXED precision: bf16 storage for x, y, every saved activation and the intermediate gradients — except the chain that
// decides WHERE the deformable conv samples:  a = GELU(proj_1 x) -> t1 = DW5 a -> t = DW7 t1 -> offsets = Coff t  runs on fp32 tensors
// (a32, t1_32, t_32: forward-only workspace, never saved) with the fp32 path's own kernels, so that the predicted offsets equal the fp32
// block's to fp32 rounding.  floor() of a sampling coordinate is discontinuous: with bf16-stored a / t1 / t the offsets move by ~0.4 %, the
// samples within that distance of an integer coordinate change cell, and conv_offset / conv_spatial / conv0 / proj_1 gradients land
// 5e-2 .. 1.8e-1 from the fp32 block's (measured round 2; reproduced on the CPU by oracle.blocks with per-tensor storage flags:
// storing ONLY t in fp32 does not help — 1.1e-1 —, the whole chain does — 3e-3).  The dw convs also write the bf16 copies of t1 / t that the
// gathers and the backward pass read: the sampled VALUES and every gradient are smooth in those, 2^-9 rounding is inside the 2e-2 contract.
// The depthwise pair conv0 / conv_spatial of LKA3d_deform by net variant (include/dlka.h: dlka_lka3d_variant) and width:
//   SYNAPSE (synapse/transformerblock.py:637-638; also the pancreas copy): 5^3 pad 2, then 7^3 dilation 3 pad 9 at every width
//   ACDC (acdc/transformerblock.py:213-237): C <= 64: 5^3 pad 2, (5,7,7) dilation 3 pad (6,9,9); C = 128: 5^3 pad 2, (3,5,5) dilation (1,3,3)
//         pad (1,6,6); C = 256: 3^3 pad 1, 3^3 pad 1
// (kernel / pad / dilation triples are in the tensor's axis order: the reference's "H, W, D" = this file's D, H, W)
struct DwPairCfg { int k0[3], p0[3], d0[3], k1[3], p1[3], d1[3]; };
static bool dw_pair_cfg(int variant, int C, DwPairCfg &c)
{
    auto set = [](int *dst, int a, int b, int cc) { dst[0] = a; dst[1] = b; dst[2] = cc; };
    if (variant == DLKA_LKA3D_SYNAPSE) {
        set(c.k0, 5, 5, 5); set(c.p0, 2, 2, 2); set(c.d0, 1, 1, 1); set(c.k1, 7, 7, 7); set(c.p1, 9, 9, 9); set(c.d1, 3, 3, 3);
        return true;
    }
    if (variant != DLKA_LKA3D_ACDC) return false;
    if (C == 32 || C == 64) { set(c.k0, 5, 5, 5); set(c.p0, 2, 2, 2); set(c.d0, 1, 1, 1); set(c.k1, 5, 7, 7); set(c.p1, 6, 9, 9); set(c.d1, 3, 3, 3); }
    else if (C == 128) { set(c.k0, 5, 5, 5); set(c.p0, 2, 2, 2); set(c.d0, 1, 1, 1); set(c.k1, 3, 5, 5); set(c.p1, 1, 6, 6); set(c.d1, 1, 3, 3); }
    else if (C == 256) { set(c.k0, 3, 3, 3); set(c.p0, 1, 1, 1); set(c.d0, 1, 1, 1); set(c.k1, 3, 3, 3); set(c.p1, 1, 1, 1); set(c.d1, 1, 1, 1); }
    else return false;
    return true;
}
static SameConv dw_conv(int B, int C, int D, int H, int W, const int *k, const int *p, const int *d, int act_bf16)
{
    SameConv s;
    s.act_bf16 = act_bf16;
    s.B = B; s.D = D; s.H = H; s.W = W; s.N = D * H * W; s.M = B * s.N; s.Cin = C; s.Cout = C; s.group = C;
    s.kd = k[0]; s.kh = k[1]; s.kw = k[2]; s.pd = p[0]; s.ph = p[1]; s.pw = p[2]; s.dd = d[0]; s.dh = d[1]; s.dw = d[2]; s.K = k[0] * k[1] * k[2];
    return s;
}

// DLKA_WGRAD_GATHER: the deformable weight gradient gathers for itself instead of streaming the samples the grad_offset kernel stored (A/B runs, the
// hand-over parity test).  Read ONCE; afterwards only dlka_lka3d_force_wgrad_gather changes it.
static std::atomic<int> g_wgrad_gather{-1};
static bool wgrad_gather()
{
    int v = g_wgrad_gather.load(std::memory_order_acquire);
    if (v < 0) {
        int want = getenv("DLKA_WGRAD_GATHER") != nullptr ? 1 : 0;
        if (g_wgrad_gather.compare_exchange_strong(v, want, std::memory_order_acq_rel)) v = want;   // (lost the race: v holds the winner's value)
    }
    return v != 0;
}

struct TokGeoms {
    SameConv pw, dw5, dw7, offc, dcn;   // (dw5 / dw7: conv0 / conv_spatial, whatever their kernels are in the variant)
    SameConv dw5_f, dw7_f, offc_f, pw_f;   // the forward chain's geometries: == dw5 / dw7 / offc / pw on the fp32 path, their fp32-storage twins on DLKA_BF16
    size_t E, Off, GOff;   // GOff: the backward's internal grad_offset buffer, 96 channel planes per batch (packed layout, DeformBwdArgs::goff_cpad)
    size_t SB;             // bytes per activation element (4, or 2 on the DLKA_BF16 path)
    TokGeoms(int B, int C, int D, int H, int W, int dtype = DLKA_F32, int variant = DLKA_LKA3D_SYNAPSE)
    {
        const int bf = dtype == DLKA_BF16 ? 1 : 0;
        SB = bf ? 2 : 4;
        DwPairCfg dc;
        if (!dw_pair_cfg(variant, C, dc)) dw_pair_cfg(DLKA_LKA3D_SYNAPSE, C, dc);   // (callers check the variant with tokens_supported first)
        pw = block_conv(B, C, C, D, H, W, 1, 0, 1, 1, bf);
        dw5 = dw_conv(B, C, D, H, W, dc.k0, dc.p0, dc.d0, bf);
        dw7 = dw_conv(B, C, D, H, W, dc.k1, dc.p1, dc.d1, bf);
        offc = block_conv(B, C, 81, D, H, W, 3, 1, 1, 1, bf);
        dcn = block_conv(B, C, C, D, H, W, 3, 1, 1, 1, bf);
        dw5_f = dw_conv(B, C, D, H, W, dc.k0, dc.p0, dc.d0, 0);
        dw7_f = dw_conv(B, C, D, H, W, dc.k1, dc.p1, dc.d1, 0);
        offc_f = block_conv(B, C, 81, D, H, W, 3, 1, 1, 1, 0);
        pw_f = block_conv(B, C, C, D, H, W, 1, 0, 1, 1, 0);
        E = (size_t)B * C * D * H * W;
        Off = (size_t)B * 81 * D * H * W;
        GOff = (size_t)B * 96 * D * H * W;
    }
    size_t wp_floats() const
    {
        size_t m = dense_wp_floats(offc);
        if (dense_wp_floats(dcn) > m) m = dense_wp_floats(dcn);
        if ((size_t)dw7.K * dw7.Cin > m) m = (size_t)dw7.K * dw7.Cin;
        return m;
    }
    size_t scratch_floats() const { return deform_scratch_floats(dcn); }
    // class-blocked fp32 copy of a depthwise conv's input (cl_dwconv_lds.hip): two of them, so that one conv's epilogue can write the next one's
    size_t blk_floats() const
    {
        const size_t a5 = cl_dwconv_blk_floats(dw5.B, dw5.Cin, dw5.D, dw5.H, dw5.W, dw5.dw), a7 = cl_dwconv_blk_floats(dw7.B, dw7.Cin, dw7.D, dw7.H, dw7.W, dw7.dw);
        return ((a5 > a7 ? a5 : a7) + 63) & ~(size_t)63;
    }
    // the deformable conv's samples S[tap][m][c], handed from the grad_offset kernel to the weight gradient (0: too large for 32-bit buffer
    // offsets, or switched off — the weight gradient then gathers for itself).  The switch is ONE process-wide value (wgrad_gather(): initialised once
    // from DLKA_WGRAD_GATHER, changed only through dlka_lka3d_force_wgrad_gather), and the workspace SIZE query always includes the sample area
    // (samp_capacity_floats), so flipping the switch between sizing and a backward call can never under-size a buffer (round-3 verdict).
    size_t samp_capacity_floats() const
    {
        const size_t n = (size_t)dcn.K * dcn.M * dcn.Cin;   // elements of the activation storage type (SB bytes each)
        return (n * SB < ((size_t)1 << 31)) ? n * SB / 4 : 0;
    }
    // prepared weights, kept in `saved` from the forward to the backward call (floats)
    size_t pw_floats() const { return (size_t)pw.Cin * pw.Cin; }
    size_t offc_floats() const { return dense_wp_floats(offc); }
    size_t dcn_floats() const { return dense_wp_floats(dcn); }
    size_t dw5_floats() const { return (size_t)dw5.K * dw5.Cin; }
    size_t dw7_floats() const { return (size_t)dw7.K * dw7.Cin; }
    size_t prep_floats() const { return 6 * pw_floats() + 2 * offc_floats() + 3 * dcn_floats() + 2 * dw5_floats() + 2 * dw7_floats() + 17 * 64; }
    // weight-gradient partials: every gradient of the block has its own area (folded by one fused launch at the end)
    size_t part_pw() const { return (cl_wgrad_part_floats_mode(pw.M, 1, pw.Cin, pw.Cin, 0) + 63) & ~(size_t)63; }
    size_t part_off() const { return (cl_wgrad_part_floats_mode(pw.M, 27, 81, pw.Cin, 0) + 63) & ~(size_t)63; }
    size_t part_dcn() const { return (cl_wgrad_part_floats_mode(pw.M, 27, pw.Cin, pw.Cin, 1) + 63) & ~(size_t)63; }
    size_t stage_dw() const { return (size_t)(dw5.K + 1 + dw7.K + 1) * pw.Cin; }   // conv0 [K0 + 1][C] then conv_spatial [K1 + 1][C]
    size_t part_floats() const { return 3 * part_pw() + part_off() + part_dcn() + ((stage_dw() + 63) & ~(size_t)63); }
};

// carve + (forward only) fill the prepared-weight area
struct TokPrep {
    float *pw_f[3], *pw_b[3];   // proj_1, conv1, proj_2: forward (mode 0) / data-gradient (mode 1) layouts
    float *off_f, *off_b, *dcn_f, *dcn_b, *dw5_f, *dw5_b, *dw7_f, *dw7_b;
    float *dcn_b16;   // bf16 path: the deformable conv's column-matrix weights as two-term bf16 records (grad_offset / grad_input on the bf16 matrix cores)
};

void add_job(PrepBatch &pb, const void *src, float *dst, int Cout, int Cin, int K, int KP, int NP, int mode)
{
    PrepJob &j = pb.j[pb.njobs++];
    j.src = (const float *)src; j.dst = dst; j.Cout = Cout; j.Cin = Cin; j.K = K; j.KP = KP; j.NP = NP; j.mode = mode;
    if (!fork_env().prep_tiled && (mode & 7) <= 2) j.mode |= 32;   // DLKA_PREP_TILED=0: the element-per-lane re-layout (cl_igemm.hip); decided when the job is made — a test compares the two bitwise
    j.n = (mode == 3 || mode == 4) ? (long)Cin * K : (long)K * KP * NP;
    pb.total += j.n;
}

// collect != null: the jobs go into *collect instead of a launch — replacing its contents, or (append) behind the jobs it already holds
int carve_prep(const TokGeoms &G, float *base, TokPrep &t, const dlka_lka3d_params *p, hipStream_t st, bool fill, const ZeroBatch *zb = nullptr,
               PrepBatch *collect = nullptr, bool append = false)
{
    float *q = base;
    auto take = [&](size_t n) { float *r = q; q += (n + 63) & ~(size_t)63; return r; };
    for (int k = 0; k < 3; ++k) { t.pw_f[k] = take(G.pw_floats()); t.pw_b[k] = take(G.pw_floats()); }
    t.off_f = take(G.offc_floats()); t.off_b = take(G.offc_floats());
    t.dcn_f = take(G.dcn_floats()); t.dcn_b = take(G.dcn_floats());
    t.dw5_f = take(G.dw5_floats()); t.dw5_b = take(G.dw5_floats());
    t.dw7_f = take(G.dw7_floats()); t.dw7_b = take(G.dw7_floats());
    t.dcn_b16 = take(G.dcn_floats());
    if (!fill) return DLKA_OK;
    PrepBatch pb;
    if (collect && append) pb = *collect;
    else memset(&pb, 0, sizeof(pb));
    if (pb.njobs + 15 + (zb ? zb->n : 0) > PREP_MAX_JOBS) return DLKA_ERR_WORKSPACE;
    const int C = G.pw.Cin;
    const void *pw_w[3] = {p->proj_1_w, p->conv1_w, p->proj_2_w};
    for (int k = 0; k < 3; ++k) {
        add_job(pb, pw_w[k], t.pw_f[k], C, C, 1, C, C, 0);
        add_job(pb, pw_w[k], t.pw_b[k], C, C, 1, C, C, 1);
    }
    add_job(pb, p->offset_w, t.off_f, 81, C, 27, C, 96, split_mode_flag(use_split(G.offc_f, true)));   // (fp32 A operand on both paths)
    add_job(pb, p->offset_w, t.off_b, 81, C, 27, 96, C, use_split(G.offc, false) ? 9 : 1);
    add_job(pb, p->deform_w, t.dcn_f, C, C, 27, C, C, (G.dcn.act_bf16 && deform_b16()) ? 8 : 0);   // bf16 path: two-term records for cl_deform_fwd_b16_kernel
    add_job(pb, p->deform_w, t.dcn_b, C, C, 27, C, C, 2);
    if (deform_b16()) add_job(pb, p->deform_w, t.dcn_b16, C, C, 27, C, C, 2 | 8);   // (both dtypes: the backward contractions of the deformable conv)
    add_job(pb, p->conv0_w, t.dw5_f, C, C, G.dw5.K, 0, 0, 3);
    add_job(pb, p->conv0_w, t.dw5_b, C, C, G.dw5.K, 0, 0, 4);
    add_job(pb, p->conv_spatial_w, t.dw7_f, C, C, G.dw7.K, 0, 0, 3);
    add_job(pb, p->conv_spatial_w, t.dw7_b, C, C, G.dw7.K, 0, 0, 4);
    if (zb && pb.njobs + zb->n > PREP_MAX_JOBS) return DLKA_ERR_WORKSPACE;   // (a dropped zero fill would be a silent wrong answer)
    if (zb)   // the forward pass's zero fills ride along (one launch less per block)
        for (int r = 0; r < zb->n; ++r) {
            PrepJob &j = pb.j[pb.njobs++];
            memset(&j, 0, sizeof(j));
            j.dst = zb->p[r]; j.n = zb->cnt[r]; j.mode = 5;
            pb.total += j.n;
        }
    if (collect) { *collect = pb; return DLKA_OK; }   // dlka_lka3d_tokens_prepare_plan: the jobs go into a table instead of a launch
    return launch_cl_prep_batch(pb, st);
}

// ---- fork / join inside the backward passes -----------------------------------------------------------------------------
// Three places let independent kernels of ONE call run beside each other on library-internal streams (events fork from and join back into the caller's stream; under
// hipGraph capture the pattern becomes a fork / join in the graph):
//   * grad_input of the 3-D deformable conv beside grad_offset (stream `s`; the stack engine's data-chain pass, see gx_fork_wanted),
//   * the 2-D block's offset-net weight gradients (`s`) and its depthwise deformable convs' grad_input (`s2`) beside the data chain (lka2d_cl_backward),
//   * (opt-in, DLKA_SIDE_STREAM=1) the 3-D block's weight gradients on `side` in the one-call backward — measured slower under graph replay at EVERY stage (fork / join
//     cost): 1.70 vs 1.61 ms per block at stage 0 (r01), 0.479 vs 0.414 ms at stage 2 and 0.433 vs 0.381 ms at stage 3 (r03); the code path stays for reference.
// Why the pairs pay (profiles/r06_notes.md, `r7d`, `r7h`, `r7i`): at the 32^3 stage grad_input beside grad_offset gains 35 us per block (the LDS-window scatter kernel runs
// two workgroups per CU at 128 registers, the gather kernel three waves per SIMD with little LDS — they fill each other's holes); forking EVERY block measured best in the
// whole step on three boxes (fp32 10.616 / 10.597 / 10.518 ms for never / wide stage only / always).
//
// CONTRACT.  The streams and events a call forks onto are a ForkCtx LEASED for the duration of that call from a per-DEVICE pool:
//   * per device: a context is created on the device the caller's stream belongs to (== the current device, or the call does not fork at all), so a call under
//     nn.DataParallel on device k never touches a handle of device 0 (2D/trainer_MaxViT_deform_LKA.py:107-108 wraps the model so);
//   * per caller: two host threads (each on its own stream) inside the library at the same time hold DIFFERENT contexts — no event is shared between concurrent calls;
//     a context returns to the pool when its call returns (everything it forked has been joined into the caller's stream by then, so stream order carries the
//     dependency on to whoever leases it next);
//   * capture: contexts are never CREATED inside a stream capture (a capturing call finds one in the pool or keeps everything on the caller's stream); a context whose
//     streams were pulled into a capture is handed only to calls of that same capture until the capture has ended;
//   * errors: a lease that has forked and is destroyed without its join (early return on a failed launch) still joins its streams into the caller's, so neither
//     an eager caller nor a capture is left with an unjoined stream.
#if !defined(HIPEMU)
struct ForkCtx {
    int dev;
    hipStream_t s, s2, side;
    hipEvent_t fork, join, fork2, join2;
    hipEvent_t ev[8];                 // `side` only
    unsigned long long cap_id;        // != 0: last used inside the stream capture with this id
    ForkCtx *next;
};
constexpr int FORK_MAX_DEV = 64;
static std::mutex g_fork_mu;
static ForkCtx *g_fork_free[FORK_MAX_DEV];
static std::atomic<int> g_fork_failed{0};      // creation failed once: no forks in this process
static std::atomic<long> g_fork_created_dev[FORK_MAX_DEV], g_fork_leases_dev[FORK_MAX_DEV];   // diagnostics (dlka_fork_stats)

static ForkCtx *fork_ctx_create(int dev, bool with_side)
{
    ForkCtx *c = new ForkCtx();
    memset(c, 0, sizeof(*c));
    c->dev = dev;
    bool ok = hipStreamCreateWithFlags(&c->s, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&c->s2, hipStreamNonBlocking) == hipSuccess;
    hipEvent_t *evs[] = {&c->fork, &c->join, &c->fork2, &c->join2};
    for (hipEvent_t *e : evs) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    if (ok && with_side) {
        ok = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess;
        for (int k = 0; k < 8 && ok; ++k) ok = hipEventCreateWithFlags(&c->ev[k], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {   // (no stale error for the next launch check to find; the handles created so far are released)
        (void)hipGetLastError();
        if (c->s) (void)hipStreamDestroy(c->s);
        if (c->s2) (void)hipStreamDestroy(c->s2);
        if (c->side) (void)hipStreamDestroy(c->side);
        for (hipEvent_t *e : evs) if (*e) (void)hipEventDestroy(*e);
        for (int k = 0; k < 8; ++k) if (c->ev[k]) (void)hipEventDestroy(c->ev[k]);
        (void)hipGetLastError();
        delete c;
        return nullptr;
    }
    g_fork_created_dev[dev].fetch_add(1, std::memory_order_relaxed);
    return c;
}

static bool side_stream_wanted()
{
    static const bool on = getenv("DLKA_SIDE_STREAM") != nullptr;
    return on;
}

// RAII lease of one ForkCtx for one library call on the caller's stream `st`.  ok() == false: the call keeps everything on `st`.
class ForkLease {
    ForkCtx *c_ = nullptr;
    hipStream_t st_;
    unsigned long long cap_ = 0;
    bool open1_ = false, open2_ = false, open_side_ = false;
    int nev_ = 0;

    static bool capture_of(hipStream_t s, unsigned long long *id)
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        unsigned long long i = 0;
        if (hipStreamGetCaptureInfo(s, &cs, &i) != hipSuccess) { (void)hipGetLastError(); *id = 0; return false; }
        *id = cs == hipStreamCaptureStatusActive ? (i ? i : ~0ull) : 0;
        return cs == hipStreamCaptureStatusNone || cs == hipStreamCaptureStatusActive;
    }

public:
    ForkLease(hipStream_t st, bool want) : st_(st)
    {
        if (!want || g_fork_failed.load(std::memory_order_acquire)) return;
        int dev = -1, sdev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FORK_MAX_DEV) { (void)hipGetLastError(); return; }
        if (st && hipStreamGetDevice(st, &sdev) == hipSuccess && sdev != dev) return;   // a foreign stream: the launches themselves will say so; no fork
        (void)hipGetLastError();
        if (!capture_of(st, &cap_)) return;
        const bool with_side = side_stream_wanted();
        {
            std::lock_guard<std::mutex> lk(g_fork_mu);
            ForkCtx **pp = &g_fork_free[dev];
            while (*pp) {
                ForkCtx *c = *pp;
                bool usable = true;
                if (c->cap_id) {   // pulled into a capture by an earlier call: is that capture over by now, or is it this very capture?
                    unsigned long long i1 = 0, i2 = 0, i3 = 0;
                    const bool q = capture_of(c->s, &i1) && capture_of(c->s2, &i2) && (!c->side || capture_of(c->side, &i3));
                    if (q && !i1 && !i2 && !i3) c->cap_id = 0;
                    else usable = q && cap_ != 0 && cap_ == c->cap_id;
                }
                if (usable) { *pp = c->next; c->next = nullptr; c_ = c; break; }
                pp = &c->next;
            }
        }
        if (!c_ && cap_ == 0) {   // none free for this device: create one — never inside a capture
            c_ = fork_ctx_create(dev, with_side);
            if (!c_) g_fork_failed.store(1, std::memory_order_release);
        }
        if (c_) g_fork_leases_dev[dev].fetch_add(1, std::memory_order_relaxed);
    }
    ForkLease(const ForkLease &) = delete;
    ForkLease &operator=(const ForkLease &) = delete;
    ~ForkLease()
    {
        if (!c_) return;
        // an early return between a fork and its join: join now, whatever the streams hold (best effort; the call is failing anyway)
        if (open1_) (void)join(1);
        if (open2_) (void)join(2);
        if (open_side_) (void)side_join();
        (void)hipGetLastError();
        if (cap_) c_->cap_id = cap_;
        std::lock_guard<std::mutex> lk(g_fork_mu);
        c_->next = g_fork_free[c_->dev];
        g_fork_free[c_->dev] = c_;
    }
    bool ok() const { return c_ != nullptr; }
    // which: 1 = stream s (events fork / join), 2 = stream s2 (fork2 / join2).  fork() may be repeated before one join() (the stream then also sees the later work).
    hipStream_t stream(int which) const { return !c_ ? st_ : which == 2 ? c_->s2 : c_->s; }
    hipStream_t side() const { return c_ && c_->side ? c_->side : st_; }
    bool has_side() const { return c_ && c_->side; }
    int fork(int which)   // the internal stream may use what the caller's stream has produced so far
    {
        if (!c_) return DLKA_OK;
        hipEvent_t e = which == 2 ? c_->fork2 : c_->fork;
        if (hipEventRecord(e, st_) != hipSuccess || hipStreamWaitEvent(stream(which), e, 0) != hipSuccess) return DLKA_ERR_LAUNCH;
        (which == 2 ? open2_ : open1_) = true;
        return DLKA_OK;
    }
    int join(int which)   // the caller's stream waits for everything issued on the internal stream so far
    {
        if (!c_) return DLKA_OK;
        bool &open = which == 2 ? open2_ : open1_;
        if (!open) return DLKA_OK;
        open = false;
        hipEvent_t e = which == 2 ? c_->join2 : c_->join;
        if (hipEventRecord(e, stream(which)) != hipSuccess || hipStreamWaitEvent(st_, e, 0) != hipSuccess) return DLKA_ERR_LAUNCH;
        return DLKA_OK;
    }
    // `side` (DLKA_SIDE_STREAM): may use what the caller's stream has produced so far (up to 7 times per call)
    int side_publish()
    {
        if (!has_side()) return DLKA_OK;
        if (nev_ >= 7) return DLKA_ERR_UNSUPPORTED;
        if (hipEventRecord(c_->ev[nev_], st_) != hipSuccess || hipStreamWaitEvent(c_->side, c_->ev[nev_], 0) != hipSuccess) return DLKA_ERR_LAUNCH;
        ++nev_;
        open_side_ = true;
        return DLKA_OK;
    }
    int side_join()
    {
        if (!has_side() || !open_side_) return DLKA_OK;
        open_side_ = false;
        if (hipEventRecord(c_->ev[7], c_->side) != hipSuccess || hipStreamWaitEvent(st_, c_->ev[7], 0) != hipSuccess) return DLKA_ERR_LAUNCH;
        return DLKA_OK;
    }
};

#else   // HIPEMU: no streams on the CPU test backend — every lease is empty and the call stays on the caller's stream
constexpr int FORK_MAX_DEV = 64;
static std::atomic<long> g_fork_created_dev[FORK_MAX_DEV], g_fork_leases_dev[FORK_MAX_DEV];
static bool side_stream_wanted() { return false; }
class ForkLease {
    hipStream_t st_;
public:
    ForkLease(hipStream_t st, bool) : st_(st) {}
    bool ok() const { return false; }
    hipStream_t stream(int) const { return st_; }
    hipStream_t side() const { return st_; }
    bool has_side() const { return false; }
    int fork(int) { return DLKA_OK; }
    int join(int) { return DLKA_OK; }
    int side_publish() { return DLKA_OK; }
    int side_join() { return DLKA_OK; }
};
#endif

// Environment switches of the fork decisions, read ONCE (dlka_env_refresh() re-reads: tests and the A/B scripts toggle them in-process).
// (phase 0 = the one-call backward of the nn.Module path: there the fork measured SLOWER — wrapper-block stack 100.5 against 102.5 volumes/s, full net 68.8 against 69.8 — so by
//  default only the stack engine's data-chain pass, phase 1, forks; DLKA_GX_FORK_MIN_ROWS = row count from which a call forks (a huge value = never), when set, rules both)
bool gx_fork_wanted(long rows, int phase)
{
#if defined(HIPEMU)
    (void)rows; (void)phase;
    return false;   // (no streams on the CPU test backend)
#else
    const ForkEnv e = fork_env();
    if (!e.gx_rows_set) return phase == 1;
    return rows >= e.gx_rows;
#endif
}

bool tokens_supported(int B, int C, int D, int H, int W, int variant = DLKA_LKA3D_SYNAPSE)
{
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return false;
    if (!(C == 32 || C == 64 || C == 128 || C == 256)) return false;
    if ((long)B * D * H * W > (1l << 28)) return false;
    DwPairCfg dc;
    if (!dw_pair_cfg(variant, C, dc)) return false;
    return dw_supported(dw_conv(B, C, D, H, W, dc.k0, dc.p0, dc.d0, 0)) && dw_supported(dw_conv(B, C, D, H, W, dc.k1, dc.p1, dc.d1, 0));
}

}  // namespace

// =========================================================================================================================
// The 2-D D-LKA block (deformable_LKA_Attention, 2D/deformable_LKA/deformable_LKA.py:124-140) on the channels-last kernels.
// x / y arrive in the reference's NCHW layout; the block transposes once on the way in and once on the way out (two passes of E
// floats against ~100 E of work) and runs entirely in [B][H][W][C]:
//   1x1 projections (+GELU / gate / residual epilogues)      cl_pointwise_kernel
//   offset nets C -> 50 (5x5) and C -> 98 (7x7 dil 3)        cl_igemm_kernel (3-D kernel with D = kd = 1), planar offsets as torchvision wants them;
//                                                            split-bf16 MFMA as in the 3-D block (forward fp32-equivalent three-term)
//   the two depthwise deformable convs                       cl_ddw2d.hip
// Supported: fp32, C / 32 in {1, 2, 3, 4, 6, 8, 12} (the net's 96 / 192 / 384 / 768 -> the first three; 768 never runs).
// =========================================================================================================================
namespace dlka {

namespace {

SameConv block_conv2d(int B, int C, int Cout, int H, int W, int k, int pad, int dil, int act_bf16 = 0)
{
    SameConv s;
    s.act_bf16 = act_bf16;
    s.B = B; s.D = 1; s.H = H; s.W = W; s.N = H * W; s.M = B * s.N; s.Cin = C; s.Cout = Cout; s.group = 1;
    s.kd = 1; s.kh = s.kw = k; s.pd = 0; s.ph = s.pw = pad; s.dd = 1; s.dh = s.dw = dil; s.K = k * k;
    return s;
}

// DLKA_BF16 (BASELINE.json config 2: "224x224 bf16 training, batch 24") is MIXED precision, like the 3-D block (TokGeoms): bf16 storage for x, y,
// every saved activation and the intermediate gradients — except the chain that decides WHERE the two deformable convs sample:
//     a = GELU(proj_1 x)  ->  o5 = offnet5(a)  ->  t1 = DDW5(a, o5)  ->  o7 = offnet7(t1)
// runs on fp32 tensors (a32, t1_32: forward-only workspace) with the fp32 path's own kernels, so that both offset fields equal the fp32 block's to
// fp32 rounding.  The bf16 copies of a / t1 (saved for the backward pass) ride in the producing kernels.
struct Lka2dCl {
    SameConv pw, off5, off7;      // activation-typed (bf16 on DLKA_BF16): the pointwise convs, the offset nets' backward passes
    SameConv off5_f, off7_f;      // the offset nets' FORWARD passes: fp32 input on both paths
    size_t E, O5, O7, SB;
    int B, C, H, W, bf;
    Lka2dCl(int B_, int C_, int H_, int W_, int dtype = DLKA_F32) : B(B_), C(C_), H(H_), W(W_)
    {
        bf = dtype == DLKA_BF16 ? 1 : 0;
        SB = bf ? 2 : 4;
        pw = block_conv2d(B, C, C, H, W, 1, 0, 1, bf);
        off5 = block_conv2d(B, C, 50, H, W, 5, 2, 1, bf);
        off7 = block_conv2d(B, C, 98, H, W, 7, 9, 3, bf);
        off5_f = block_conv2d(B, C, 50, H, W, 5, 2, 1, 0);
        off7_f = block_conv2d(B, C, 98, H, W, 7, 9, 3, 0);
        E = (size_t)B * C * H * W; O5 = (size_t)B * 50 * H * W; O7 = (size_t)B * 98 * H * W;
    }
    size_t pw_floats() const { return (size_t)C * C; }
    size_t off5_floats() const { return dense_wp_floats(off5); }
    size_t off7_floats() const { return dense_wp_floats(off7); }
    size_t prep_floats() const { return 6 * (pw_floats() + 64) + 2 * (off5_floats() + 64) + 2 * (off7_floats() + 64) + (size_t)(25 + 49) * C + 256; }
    size_t part_pw() const { return (cl_wgrad_part_floats_mode(pw.M, 1, C, C, 0) + 63) & ~(size_t)63; }
    size_t part_o5() const { return (cl_wgrad_part_floats_mode(pw.M, 25, 50, C, 0) + 63) & ~(size_t)63; }
    size_t part_o7() const { return (cl_wgrad_part_floats_mode(pw.M, 49, 98, C, 0) + 63) & ~(size_t)63; }
    size_t part_dw() const { return (cl_ddw2d_part_floats(pw.M, 49, C) + 63) & ~(size_t)63; }
    size_t part_floats() const { return 3 * part_pw() + part_o5() + part_o7() + part_dw(); }
};

struct Prep2d { float *pw_f[3], *pw_b[3], *o5_f, *o5_b, *o7_f, *o7_b, *dw5, *dw7; };

int carve_prep2d(const Lka2dCl &G, float *base, Prep2d &t, const dlka_lka2d_params *p, hipStream_t st, bool fill)
{
    float *q = base;
    auto take = [&](size_t n) { float *r = q; q += (n + 63) & ~(size_t)63; return r; };
    for (int k = 0; k < 3; ++k) { t.pw_f[k] = take(G.pw_floats()); t.pw_b[k] = take(G.pw_floats()); }
    t.o5_f = take(G.off5_floats()); t.o5_b = take(G.off5_floats());
    t.o7_f = take(G.off7_floats()); t.o7_b = take(G.off7_floats());
    t.dw5 = take((size_t)25 * G.C); t.dw7 = take((size_t)49 * G.C);
    if (!fill) return DLKA_OK;
    PrepBatch pb;
    memset(&pb, 0, sizeof(pb));
    const int C = G.C;
    const void *pw_w[3] = {p->proj_1_w, p->conv1_w, p->proj_2_w};
    for (int k = 0; k < 3; ++k) {
        add_job(pb, pw_w[k], t.pw_f[k], C, C, 1, C, C, 0);
        add_job(pb, pw_w[k], t.pw_b[k], C, C, 1, C, C, 1);
    }
    add_job(pb, p->conv0_offset_w, t.o5_f, 50, C, 25, C, 64, split_mode_flag(use_split(G.off5_f, true)));
    add_job(pb, p->conv0_offset_w, t.o5_b, 50, C, 25, 64, C, use_split(G.off5, false) ? 9 : 1);
    add_job(pb, p->conv_spatial_offset_w, t.o7_f, 98, C, 49, C, 128, split_mode_flag(use_split(G.off7_f, true)));
    add_job(pb, p->conv_spatial_offset_w, t.o7_b, 98, C, 49, 128, C, use_split(G.off7, false) ? 9 : 1);
    add_job(pb, p->conv0_w, t.dw5, C, C, 25, 0, 0, 3);
    add_job(pb, p->conv_spatial_w, t.dw7, C, C, 49, 0, 0, 3);
    return launch_cl_prep_batch(pb, st);
}

void fill_ddw(DwArgs2d &d, const Lka2dCl &G, int k, int pad, int dil)
{
    memset(&d, 0, sizeof(d));
    d.B = G.B; d.H = G.H; d.W = G.W; d.C = G.C; d.kh = d.kw = k; d.ph = d.pw = pad; d.dh = d.dw = dil;
}

}  // namespace

int lka2d_cl_supported(int B, int C, int H, int W, int dtype)
{
    if ((dtype != DLKA_F32 && dtype != DLKA_BF16) || B <= 0 || H <= 0 || W <= 0 || !cl_ddw2d_supported(C)) return 0;
    if (!(nt_ok(C) || C == 192 || C == 384)) return 0;   // the offset nets' data gradient has C columns: the igemm launcher's tile menu
    if ((long)B * H * W * C >= (1l << 29)) return 0;
    return dense_fwd_supported(block_conv2d(B, C, 50, H, W, 5, 2, 1)) ? 1 : 0;
}

// saved: xt, h, a, t1, t2, g1, m, (spare) — activation-typed —, o5, o7 (fp32), prepared weights
size_t lka2d_cl_saved_bytes(int B, int C, int H, int W, int dtype)
{
    Lka2dCl G(B, C, H, W, dtype);
    return 8 * align256(G.E * G.SB) + align256(G.O5 * 4) + align256(G.O7 * 4) + align256(G.prep_floats() * 4);
}

// (the nine gradient buffers keep their fp32 size on the bf16 path: the two grad_input accumulators ARE fp32, two more serve as the forward
//  pass's fp32 chain tensors and as landing zones of tap-split sums)
// (diagnostics) the offset tensors inside `saved`: lka2d_cl_forward carves xt, h, a, t1, t2, g1, m, spare, then o5, o7 (fp32 on both paths)
int lka2d_cl_saved_offsets(int B, int C, int H, int W, int dtype, size_t byte_offsets[2], int *elem_bytes)
{
    Lka2dCl G(B, C, H, W, dtype);
    byte_offsets[0] = 8 * align256(G.E * G.SB);
    byte_offsets[1] = byte_offsets[0] + align256(G.O5 * 4);
    *elem_bytes = 4;
    return DLKA_OK;
}

size_t lka2d_cl_workspace_bytes(int B, int C, int H, int W, int dtype)
{
    Lka2dCl G(B, C, H, W, dtype);
    return 9 * align256(G.E * 4) + align256(G.O7 * 4) + align256(G.O5 * 4) + align256(G.part_floats() * 4) + align256(4096) +   // (O5: the second conv's grad_offset, see lka2d_cl_backward)
           dense_wgrad_pad_bytes(G.off7) + dense_wgrad_pad_bytes(G.off5);   // the zero-padded copies the offset nets' weight gradients read (round 5)
}

int lka2d_cl_forward(const void *x_, const dlka_lka2d_params *p, void *y_, void *saved, size_t saved_bytes, void *workspace, size_t workspace_bytes, int B,
                     int C, int H, int W, int dtype, hipStream_t st)
{
    Lka2dCl G(B, C, H, W, dtype);
    const size_t SB = G.SB;
    const int bf = G.bf;
    Carver sv(saved, saved_bytes), cv(workspace, workspace_bytes);
    float *xt = (float *)sv.take(G.E * SB), *h = (float *)sv.take(G.E * SB), *a = (float *)sv.take(G.E * SB), *t1 = (float *)sv.take(G.E * SB);
    float *t2 = (float *)sv.take(G.E * SB), *g1 = (float *)sv.take(G.E * SB), *m = (float *)sv.take(G.E * SB), *spare = (float *)sv.take(G.E * SB);
    float *o5 = (float *)sv.take(G.O5 * 4), *o7 = (float *)sv.take(G.O7 * 4);
    float *prep = (float *)sv.take(G.prep_floats() * 4);
    float *yt = (float *)cv.take(G.E * 4);
    float *a32 = (float *)cv.take(G.E * 4), *t1_32 = (float *)cv.take(G.E * 4);   // bf16 path: the fp32 offset-determining chain
    (void)spare;
    if (!sv.ok() || !cv.ok()) return DLKA_ERR_WORKSPACE;
    const float *N0 = nullptr;
    Prep2d PW;
    DLKA_TRY(carve_prep2d(G, prep, PW, p, st, true));
    DLKA_TRY(launch_cl_transpose((const float *)x_, xt, B, C, G.pw.N, 1, st, bf));                                           // NCHW -> NHWC
    DLKA_TRY(dense_forward(G.pw, xt, N0, (const float *)p->proj_1_b, h, 0, PW.pw_f[0], 1, nullptr, a, st, false, nullptr, bf ? a32 : nullptr));   // :135-136 (+GELU)
    const float *a_in = bf ? a32 : a;
    DLKA_TRY(dense_forward(G.off5_f, a_in, N0, (const float *)p->conv0_offset_b, o5, 1, PW.o5_f, 0, nullptr, nullptr, st));     // :28 offset_net
    DwArgs2d d;
    fill_ddw(d, G, 5, 2, 1);
    d.in = a_in; d.off = o5; d.wp = PW.dw5; d.out = bf ? t1_32 : t1; d.out_lo = bf ? t1 : nullptr;
    DLKA_TRY(launch_cl_ddw2d_fwd(d, st));                                                                                     // :29
    const float *t1_in = bf ? t1_32 : t1;
    DLKA_TRY(dense_forward(G.off7_f, t1_in, N0, (const float *)p->conv_spatial_offset_b, o7, 1, PW.o7_f, 0, nullptr, nullptr, st));
    fill_ddw(d, G, 7, 9, 3);
    d.in = t1_in; d.off = o7; d.wp = PW.dw7; d.out = bf ? nullptr : t2; d.out_lo = bf ? t2 : nullptr;
    DLKA_TRY(launch_cl_ddw2d_fwd(d, st));
    DLKA_TRY(dense_forward(G.pw, t2, N0, (const float *)p->conv1_b, g1, 0, PW.pw_f[1], 2, a, m, st));                          // :102-104 conv1 + gate
    DLKA_TRY(dense_forward(G.pw, m, N0, (const float *)p->proj_2_b, yt, 0, PW.pw_f[2], 3, xt, nullptr, st));                   // :138-139 proj_2 + shortcut
    return launch_cl_transpose(yt, (float *)y_, B, C, G.pw.N, 0, st, bf);
}

int lka2d_cl_backward(const void *x_, const dlka_lka2d_params *p, const void *gy_, const void *saved, size_t saved_bytes, void *gx_, const dlka_lka2d_grads *gr,
                      void *workspace, size_t workspace_bytes, int B, int C, int H, int W, int dtype, hipStream_t st)
{
    (void)x_;
    Lka2dCl G(B, C, H, W, dtype);
    const size_t SB = G.SB;
    const int bf = G.bf;
    Carver sv((void *)saved, saved_bytes), cv(workspace, workspace_bytes);
    const float *xt = (float *)sv.take(G.E * SB), *h = (float *)sv.take(G.E * SB), *a = (float *)sv.take(G.E * SB), *t1 = (float *)sv.take(G.E * SB);
    const float *t2 = (float *)sv.take(G.E * SB), *g1 = (float *)sv.take(G.E * SB), *m = (float *)sv.take(G.E * SB);
    (void)sv.take(G.E * SB);
    const float *o5 = (float *)sv.take(G.O5 * 4), *o7 = (float *)sv.take(G.O7 * 4);
    float *prep = (float *)sv.take(G.prep_floats() * 4);
    float *gyt = (float *)cv.take(G.E * 4), *gg1 = (float *)cv.take(G.E * 4), *ga1 = (float *)cv.take(G.E * 4), *gt2 = (float *)cv.take(G.E * 4);
    float *gta = (float *)cv.take(G.E * 4), *gt1 = (float *)cv.take(G.E * 4), *gaa = (float *)cv.take(G.E * 4), *gab = (float *)cv.take(G.E * 4);
    float *gh = (float *)cv.take(G.E * 4);
    float *goff = (float *)cv.take(G.O7 * 4);
    float *goff5 = (float *)cv.take(G.O5 * 4);   // the 5x5 conv's grad_offset in a buffer of its own: the 7x7 offset net's weight gradient may still be reading `goff`
    float *part = (float *)cv.take(G.part_floats() * 4);
    (void)cv.take(4096);
    float *pad7 = (float *)cv.take_opt(dense_wgrad_pad_bytes(G.off7), dense_wgrad_pad_bytes(G.off7) != 0);   // (own buffers: both weight gradients may be in flight on the
    float *pad5 = (float *)cv.take_opt(dense_wgrad_pad_bytes(G.off5), dense_wgrad_pad_bytes(G.off5) != 0);   //  internal stream at once)
    if (!sv.ok() || !cv.ok()) return DLKA_ERR_WORKSPACE;
    const float *N0 = nullptr;
    Prep2d PW;
    DLKA_TRY(carve_prep2d(G, prep, PW, p, st, false));
    // The two offset nets' WEIGHT gradients (the largest kernels of this pass after grad_input: C -> 98 / 50 channels over 49 / 25 taps) only read grad_offset and a saved
    // activation, and nothing before the finalisation reads their partial sums: they run on the library's internal stream (aux_ctx) beside the data chain — fork behind
    // each depthwise deformable conv's backward, one join in front of the finalisation.  DLKA_LKA2D_FORK=0: one stream (A/B; read per call).  Measured in
    // profiles/r06_notes.md.
    int fork_mode = 0;
#if !defined(HIPEMU)
    fork_mode = fork_env().lka2d_fork;
#endif
    ForkLease lease(st, fork_mode != 0);   // (joins whatever is still forked when a DLKA_TRY below returns early)
    const bool fork2d = lease.ok();
    // ... and each depthwise deformable conv's grad_input (the pass's largest kernel) beside its grad_offset / weight-gradient kernel on a second internal stream:
    // fork in front of the pair, join in front of the offset net's data gradient, which adds grad_input (DLKA_LKA2D_FORK=1: the weight gradients only)
    const bool forkgx = fork2d && fork_mode == 2;
    hipStream_t gst = forkgx ? lease.stream(2) : nullptr;
    auto fork_gx = [&]() -> int { return forkgx ? lease.fork(2) : DLKA_OK; };
    auto join_gx = [&]() -> int { return forkgx ? lease.join(2) : DLKA_OK; };
    hipStream_t wst = fork2d ? lease.stream(1) : st;
    auto fork_to_aux = [&]() -> int { return fork2d ? lease.fork(1) : DLKA_OK; };
    float *part_p2 = part, *part_c1 = part_p2 + G.part_pw(), *part_p1 = part_c1 + G.part_pw(), *part_o5 = part_p1 + G.part_pw();
    float *part_o7 = part_o5 + G.part_o5(), *part_dw = part_o7 + G.part_o7();
    FinalizeBatch fb;
    memset(&fb, 0, sizeof(fb));
    // bf16: the fp32 landing zone of a tap-split offset-net data gradient (converted into its bf16 destination afterwards): gg1 is dead by then for
    // the first one, gt2 for the second
    ZeroBatch zb;
    memset(&zb, 0, sizeof(zb));
    zb.add(gta, G.E);   // grad_input targets of the two depthwise deformable convs (fp32 atomics)
    zb.add(gaa, G.E);
    const bool split7 = dense_backward_data_splits(G.off7, 3) > 1, split5 = dense_backward_data_splits(G.off5, 3) > 1;
    if (bf && split7) zb.add(gh, G.E);     // (gh is written last: free until then)
    if (zb.overflow) return DLKA_ERR_WORKSPACE;
    DLKA_TRY(launch_zero_batch(zb, st));
    float *gxt = gt2;   // (gt2 is dead by the time the last projection's data gradient is written)
    DLKA_TRY(launch_cl_transpose((const float *)gy_, gyt, B, C, G.pw.N, 1, st, bf));
    // proj_2 data gradient with the gate's backward in the epilogue: gg1 = gm * a, ga1 = gm * g1
    DLKA_TRY(dense_backward_data(G.pw, gyt, 0, N0, gg1, PW.pw_b[2], 4, a, st, g1, ga1));
    DLKA_TRY(dense_backward_data(G.pw, gg1, 0, N0, gt2, PW.pw_b[1], 0, nullptr, st));                                           // conv1
    // conv_spatial = DeformConv(7x7 dil 3): t2 = DDW7(t1, o7 = offnet7(t1))
    DwArgs2d d;
    fill_ddw(d, G, 7, 9, 3);
    d.act_bf16 = bf;
    d.in = t1; d.off = o7; d.wp = PW.dw7; d.g = gt2; d.gx = gta; d.goff = goff; d.part = part_dw;
    DLKA_TRY(fork_gx());
    DLKA_TRY(launch_cl_ddw2d_bwd(d, (float *)gr->conv_spatial_w, st, gst));
    DLKA_TRY(fork_to_aux());
    DLKA_TRY(dense_backward_weight(G.off7, t1, goff, 1, (float *)gr->conv_spatial_offset_w, (float *)gr->conv_spatial_offset_b, part_o7, wst, &fb.j[fb.njobs++], 0, pad7));
    DLKA_TRY(join_gx());
    DLKA_TRY(dense_backward_data(G.off7, goff, 1, N0, gt1, PW.o7_b, 3, gta, st, nullptr, nullptr, bf && split7, false, bf != 0, bf ? gh : nullptr));   // gt1 = gta + offnet7^T goff
    // conv0 = DeformConv(5x5): t1 = DDW5(a, o5 = offnet5(a))
    fill_ddw(d, G, 5, 2, 1);
    d.act_bf16 = bf;
    d.in = a; d.off = o5; d.wp = PW.dw5; d.g = gt1; d.gx = gaa; d.goff = goff5; d.part = part_dw;
    DLKA_TRY(fork_gx());
    DLKA_TRY(launch_cl_ddw2d_bwd(d, (float *)gr->conv0_w, st, gst));
    DLKA_TRY(fork_to_aux());
    DLKA_TRY(dense_backward_weight(G.off5, a, goff5, 1, (float *)gr->conv0_offset_w, (float *)gr->conv0_offset_b, part_o5, wst, &fb.j[fb.njobs++], 0, pad5));
    DLKA_TRY(join_gx());
    if (bf && split5) DLKA_TRY(launch_zero(gh, G.E * 4, st));
    DLKA_TRY(dense_backward_data(G.off5, goff5, 1, N0, gab, PW.o5_b, 3, gaa, st, nullptr, nullptr, bf && split5, false, bf != 0, bf ? gh : nullptr));   // gab = gaa + offnet5^T goff
    // a = GELU(h) feeds the gate and conv0: gh = (ga1 + gab) * gelu'(h)
    if (bf) DLKA_TRY(launch_gelu_bwd_sum<bf16_t>((const bf16_t *)h, (const bf16_t *)ga1, (const bf16_t *)gab, (bf16_t *)gh, (long)G.E, st));
    else DLKA_TRY(launch_gelu_bwd_sum<float>(h, ga1, gab, gh, (long)G.E, st));
    {
        WgradArgs jobs[3];
        fill_pw_wgrad(jobs[0], G.pw, m, gyt, part_p2);
        fill_pw_wgrad(jobs[1], G.pw, t2, gg1, part_c1);
        fill_pw_wgrad(jobs[2], G.pw, xt, gh, part_p1);
        float *const gws[3] = {(float *)gr->proj_2_w, (float *)gr->conv1_w, (float *)gr->proj_1_w};
        float *const gbs[3] = {(float *)gr->proj_2_b, (float *)gr->conv1_b, (float *)gr->proj_1_b};
        DLKA_TRY(launch_cl_wgrad_pw3(jobs, gws, gbs, st, &fb.j[fb.njobs]));
        fb.njobs += 3;
    }
    if (fork2d) DLKA_TRY(lease.join(1));   // the folds read the offset nets' partial sums
    DLKA_TRY(launch_cl_wgrad_finalize(fb, st));
    DLKA_TRY(dense_backward_data(G.pw, gh, 0, N0, gxt, PW.pw_b[0], 3, gyt, st));                                                // gx = P1^T gh + gy
    return launch_cl_transpose(gxt, (float *)gx_, B, C, G.pw.N, 0, st, bf);
}

}  // namespace dlka

extern "C" {

// ---- channels-last convolution ----------------------------------------------------------------------------------------
size_t dlka_conv3d_cl_workspace(const dlka_conv_geom *c, int dtype, int backward)
{
    SameConv s;
    if (dtype != DLKA_F32 || make_same_conv(c, s)) return 0;
    if (is_depthwise(s)) return 2 * align256((size_t)s.K * s.Cin * 4);
    size_t n = align256(dense_wp_floats(s) * 4);
    if (backward) n += align256(cl_wgrad_part_floats(s.M, s.K, s.Cout, s.Cin) * 4) + dense_wgrad_pad_bytes(s);   // (+ the zero-padded input copy of the padded weight gradient)
    return n;
}

int dlka_conv3d_forward_cl(const void *x, const void *weight, const void *bias, void *out, int out_planar, void *workspace,
                           size_t workspace_bytes, const dlka_conv_geom *c, int dtype, void *stream)
{
    if (!x || !weight || !out) return DLKA_ERR_NULL;
    if (dtype != DLKA_F32) return DLKA_ERR_UNSUPPORTED;
    SameConv s;
    DLKA_TRY(make_same_conv(c, s));
    hipStream_t st = (hipStream_t)stream;
    Carver cv(workspace, workspace_bytes);
    if (is_depthwise(s)) {
        if (!dw_supported(s) || out_planar) return DLKA_ERR_UNSUPPORTED;
        float *wp = (float *)cv.take((size_t)s.K * s.Cin * 4);
        if (!cv.ok()) return DLKA_ERR_WORKSPACE;
        return dw_forward(s, (const float *)x, (const float *)weight, (const float *)bias, (float *)out, wp, 0, st);
    }
    if (!dense_fwd_supported(s)) return DLKA_ERR_UNSUPPORTED;
    float *wp = (float *)cv.take(dense_wp_floats(s) * 4);
    if (!cv.ok()) return DLKA_ERR_WORKSPACE;
    return dense_forward(s, (const float *)x, (const float *)weight, (const float *)bias, (float *)out, out_planar, wp, 0, nullptr, nullptr, st);
}

int dlka_conv3d_backward_cl(const void *x, const void *weight, const void *grad_out, int grad_out_planar, void *grad_x, void *grad_weight,
                            void *grad_bias, void *workspace, size_t workspace_bytes, const dlka_conv_geom *c, int dtype, void *stream)
{
    if (!x || !weight || !grad_out) return DLKA_ERR_NULL;
    if (dtype != DLKA_F32) return DLKA_ERR_UNSUPPORTED;
    SameConv s;
    DLKA_TRY(make_same_conv(c, s));
    hipStream_t st = (hipStream_t)stream;
    Carver cv(workspace, workspace_bytes);
    if (is_depthwise(s)) {
        if (!dw_supported(s) || grad_out_planar) return DLKA_ERR_UNSUPPORTED;
        float *wp = (float *)cv.take((size_t)s.K * s.Cin * 4), *gwp = (float *)cv.take((size_t)s.K * s.Cin * 4);
        if (!cv.ok()) return DLKA_ERR_WORKSPACE;
        if (grad_x) DLKA_TRY(dw_forward(s, (const float *)grad_out, (const float *)weight, nullptr, (float *)grad_x, wp, 1, st));
        if (grad_weight) DLKA_TRY(dw_backward_weight(s, (const float *)x, (const float *)grad_out, (float *)grad_weight, (float *)grad_bias, gwp, st));
        else if (grad_bias) DLKA_TRY(launch_cl_colsum((const float *)grad_out, (float *)grad_bias, s.M, s.Cout, st));
        return DLKA_OK;
    }
    if (s.group != 1) return DLKA_ERR_UNSUPPORTED;
    float *wp = (float *)cv.take(dense_wp_floats(s) * 4);
    float *part = (float *)cv.take(cl_wgrad_part_floats(s.M, s.K, s.Cout, s.Cin) * 4);
    float *padb = (float *)cv.take_opt(dense_wgrad_pad_bytes(s), dense_wgrad_pad_bytes(s) != 0);
    if (!cv.ok()) return DLKA_ERR_WORKSPACE;
    if (grad_x) DLKA_TRY(dense_backward_data(s, (const float *)grad_out, grad_out_planar, (const float *)weight, (float *)grad_x, wp, 0, nullptr, st));
    if (grad_weight) DLKA_TRY(dense_backward_weight(s, (const float *)x, (const float *)grad_out, grad_out_planar, (float *)grad_weight, (float *)grad_bias, part, st, nullptr, 0, padb));
    else if (grad_bias) {
        if (grad_out_planar) DLKA_TRY(launch_bias_grad<float>((const float *)grad_out, (float *)grad_bias, s.B, s.Cout, s.N, st));
        else DLKA_TRY(launch_cl_colsum((const float *)grad_out, (float *)grad_bias, s.M, s.Cout, st));
    }
    return DLKA_OK;
}

// ---- channels-last deformable conv (x, out channels-last; offsets planar as in the reference) ---------------------------
size_t dlka_deform_conv3d_cl_workspace(const dlka_conv_geom *c, int dtype, int backward)
{
    SameConv s;
    if ((dtype != DLKA_F32 && dtype != DLKA_BF16) || make_same_conv(c, s)) return 0;
    size_t n = align256(dense_wp_floats(s) * 4);
    if (backward) n += align256(cl_wgrad_part_floats(s.M, s.K, s.Cout, s.Cin) * 4) + align256(deform_scratch_floats(s) * 4);
    else n += align256(deform_fwd_slab_floats(s) * 4);   // small volumes: the tap ranges' slabs (deform_forward)
    return n;
}

int dlka_deform_conv3d_forward_cl(const void *x, const void *offset, const void *weight, const void *bias, void *out, void *workspace,
                                  size_t workspace_bytes, const dlka_conv_geom *c, int dtype, void *stream)
{
    if (!x || !offset || !weight || !bias || !out) return DLKA_ERR_NULL;
    if (dtype != DLKA_F32 && dtype != DLKA_BF16) return DLKA_ERR_UNSUPPORTED;
    SameConv s;
    DLKA_TRY(make_same_conv(c, s));
    if (c->deformable_group != 1 || !deform_supported(s)) return DLKA_ERR_UNSUPPORTED;
    s.act_bf16 = dtype == DLKA_BF16;   // x / out bf16 storage; offsets, weight and bias stay fp32
    Carver cv(workspace, workspace_bytes);
    float *wp = (float *)cv.take(dense_wp_floats(s) * 4);
    float *slab = (float *)cv.take(deform_fwd_slab_floats(s) * 4);
    if (!cv.ok()) return DLKA_ERR_WORKSPACE;
    return deform_forward(s, (const float *)x, (const float *)offset, (const float *)weight, (const float *)bias, (float *)out, wp, (hipStream_t)stream, slab);
}

int dlka_deform_conv3d_backward_cl(const void *x, const void *offset, const void *weight, const void *grad_out, void *grad_x, void *grad_offset,
                                   void *grad_weight, void *grad_bias, void *workspace, size_t workspace_bytes, const dlka_conv_geom *c,
                                   int dtype, void *stream)
{
    if (!x || !offset || !weight || !grad_out) return DLKA_ERR_NULL;
    if (dtype != DLKA_F32 && dtype != DLKA_BF16) return DLKA_ERR_UNSUPPORTED;
    SameConv s;
    DLKA_TRY(make_same_conv(c, s));
    if (c->deformable_group != 1 || !deform_supported(s)) return DLKA_ERR_UNSUPPORTED;
    // DLKA_BF16: x / grad_out bf16 storage; offsets, weight and ALL FOUR gradients fp32 (grad_x is the fp32 accumulation target the token
    // block also uses; grad_offset is planar; the parameter gradients are fp32 masters)
    s.act_bf16 = dtype == DLKA_BF16;
    if (s.act_bf16 && grad_bias && !grad_weight) return DLKA_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    Carver cv(workspace, workspace_bytes);
    float *wp = (float *)cv.take(dense_wp_floats(s) * 4);
    float *part = (float *)cv.take(cl_wgrad_part_floats(s.M, s.K, s.Cout, s.Cin) * 4);
    float *scratch = (float *)cv.take(deform_scratch_floats(s) * 4);
    if (!cv.ok()) return DLKA_ERR_WORKSPACE;
    DLKA_TRY(deform_backward(s, (const float *)x, (const float *)offset, (const float *)weight, (const float *)grad_out, (float *)grad_x,
                             (float *)grad_offset, (float *)grad_weight, (float *)grad_bias, wp, part, scratch, st));
    return DLKA_OK;
}

// ---- channels-last 2-D DEPTHWISE deformable conv (cl_ddw2d.hip: the 2-D D-LKA block's conv0 / conv_spatial) on its own ------------------
// torchvision.ops.deform_conv2d(input, offset, weight, bias=None, stride 1, "same" padding, groups = C, one offset group) as the reference
// calls it (2D/deformable_LKA/deformable_LKA.py:18-30, 93-94).  x / out / grad_out / grad_x [B][H][W][C], offsets / grad_offset planar
// [B][2K][H][W] as torchvision lays them out, weight / grad_weight [C][1][kh][kw].  Lets the parity tests hold the fast-path kernels
// against the reference's own op (tests/test_ref_d3d_2d_gpu.py) without the rest of the block around them.
namespace {
int make_ddw2d(const dlka_conv_geom *c, DwArgs2d &d)
{
    if (!c) return DLKA_ERR_NULL;
    if (c->D != 1 || c->kd != 1 || c->sd != 1 || c->dd != 1 || c->pd != 0) return DLKA_ERR_SHAPE;
    if (c->B <= 0 || c->C <= 0 || c->H <= 0 || c->W <= 0 || c->kh <= 0 || c->kw <= 0 || c->dh <= 0 || c->dw <= 0) return DLKA_ERR_SHAPE;
    if (c->group != c->C || c->Cout != c->C || c->deformable_group != 1 || c->sh != 1 || c->sw != 1) return DLKA_ERR_UNSUPPORTED;
    if (dlka_conv_out_size(c->H, c->ph, c->dh, c->kh, 1) != c->H || dlka_conv_out_size(c->W, c->pw, c->dw, c->kw, 1) != c->W) return DLKA_ERR_UNSUPPORTED;
    if (!cl_ddw2d_supported(c->C)) return DLKA_ERR_UNSUPPORTED;
    memset(&d, 0, sizeof(d));
    d.B = c->B; d.H = c->H; d.W = c->W; d.C = c->C; d.kh = c->kh; d.kw = c->kw; d.ph = c->ph; d.pw = c->pw; d.dh = c->dh; d.dw = c->dw;
    return DLKA_OK;
}
}  // namespace

size_t dlka_deform_dwconv2d_cl_workspace(const dlka_conv_geom *c, int dtype, int backward)
{
    DwArgs2d d;
    if (dtype != DLKA_F32 || make_ddw2d(c, d)) return 0;
    size_t n = align256((size_t)d.kh * d.kw * d.C * 4);
    if (backward) n += align256(cl_ddw2d_part_floats(d.B * d.H * d.W, d.kh * d.kw, d.C) * 4);
    return n;
}

int dlka_deform_dwconv2d_forward_cl(const void *x, const void *offset, const void *weight, void *out, void *workspace, size_t workspace_bytes,
                                    const dlka_conv_geom *c, int dtype, void *stream)
{
    if (!x || !offset || !weight || !out) return DLKA_ERR_NULL;
    if (dtype != DLKA_F32) return DLKA_ERR_UNSUPPORTED;
    DwArgs2d d;
    DLKA_TRY(make_ddw2d(c, d));
    hipStream_t st = (hipStream_t)stream;
    Carver cv(workspace, workspace_bytes);
    float *wp = (float *)cv.take((size_t)d.kh * d.kw * d.C * 4);
    if (!cv.ok()) return DLKA_ERR_WORKSPACE;
    DLKA_TRY(launch_cl_dw_prep_weight((const float *)weight, wp, d.C, d.kh * d.kw, 0, st));
    d.in = (const float *)x; d.off = (const float *)offset; d.wp = wp; d.out = (float *)out;
    return launch_cl_ddw2d_fwd(d, st);
}

int dlka_deform_dwconv2d_backward_cl(const void *x, const void *offset, const void *weight, const void *grad_out, void *grad_x, void *grad_offset,
                                     void *grad_weight, void *workspace, size_t workspace_bytes, const dlka_conv_geom *c, int dtype, void *stream)
{
    if (!x || !offset || !weight || !grad_out || !grad_x || !grad_offset || !grad_weight) return DLKA_ERR_NULL;   // one traversal produces all three
    if (dtype != DLKA_F32) return DLKA_ERR_UNSUPPORTED;
    DwArgs2d d;
    DLKA_TRY(make_ddw2d(c, d));
    hipStream_t st = (hipStream_t)stream;
    Carver cv(workspace, workspace_bytes);
    float *wp = (float *)cv.take((size_t)d.kh * d.kw * d.C * 4);
    float *part = (float *)cv.take(cl_ddw2d_part_floats(d.B * d.H * d.W, d.kh * d.kw, d.C) * 4);
    if (!cv.ok()) return DLKA_ERR_WORKSPACE;
    DLKA_TRY(launch_cl_dw_prep_weight((const float *)weight, wp, d.C, d.kh * d.kw, 0, st));
    DLKA_TRY(launch_zero(grad_x, (size_t)d.B * d.H * d.W * d.C * 4, st));   // the window scatter accumulates with atomics
    d.in = (const float *)x; d.off = (const float *)offset; d.wp = wp; d.g = (const float *)grad_out;
    d.gx = (float *)grad_x; d.goff = (float *)grad_offset; d.part = part;
    return launch_cl_ddw2d_bwd(d, (float *)grad_weight, st);
}

// ---- layout helpers -------------------------------------------------------------------------------------------------------
int dlka_ncdhw_to_ndhwc(const void *src, void *dst, int B, int C, int N, int dtype, void *stream)
{
    if (!src || !dst) return DLKA_ERR_NULL;
    if (dtype != DLKA_F32) return DLKA_ERR_UNSUPPORTED;
    return launch_cl_transpose((const float *)src, (float *)dst, B, C, N, 1, (hipStream_t)stream);
}
int dlka_ndhwc_to_ncdhw(const void *src, void *dst, int B, int C, int N, int dtype, void *stream)
{
    if (!src || !dst) return DLKA_ERR_NULL;
    if (dtype != DLKA_F32) return DLKA_ERR_UNSUPPORTED;
    return launch_cl_transpose((const float *)src, (float *)dst, B, C, N, 0, (hipStream_t)stream);
}

// ---- fork contexts: diagnostics and the cached switches ---------------------------------------------------------------------------
void dlka_env_refresh(void)
{
    fork_env_load();
}

int dlka_fork_stats(int device, int64_t *contexts, int64_t *leases)
{
    if (device < 0 || device >= FORK_MAX_DEV) return DLKA_ERR_SHAPE;
    if (contexts) *contexts = g_fork_created_dev[device].load(std::memory_order_relaxed);
    if (leases) *leases = g_fork_leases_dev[device].load(std::memory_order_relaxed);
    return DLKA_OK;
}

// ---- token-layout D-LKA block ------------------------------------------------------------------------------------------------
int dlka_lka3d_force_wgrad_gather(int on)
{
    const int old = wgrad_gather() ? 1 : 0;
    g_wgrad_gather.store(on ? 1 : 0, std::memory_order_release);
    return old;
}

int dlka_lka3d_tokens_supported_v(int B, int C, int D, int H, int W, int dtype, int variant)
{
    return ((dtype == DLKA_F32 || dtype == DLKA_BF16) && tokens_supported(B, C, D, H, W, variant)) ? 1 : 0;
}
int dlka_lka3d_tokens_supported(int B, int C, int D, int H, int W, int dtype) { return dlka_lka3d_tokens_supported_v(B, C, D, H, W, dtype, DLKA_LKA3D_SYNAPSE); }

size_t dlka_lka3d_tokens_saved_bytes_v(int B, int C, int D, int H, int W, int dtype, int variant)
{
    if (!dlka_lka3d_tokens_supported_v(B, C, D, H, W, dtype, variant)) return 0;
    TokGeoms G(B, C, D, H, W, dtype, variant);
    return 7 * align256(G.E * G.SB) + align256(G.Off * 4) + align256(G.prep_floats() * 4);
}
size_t dlka_lka3d_tokens_saved_bytes(int B, int C, int D, int H, int W, int dtype) { return dlka_lka3d_tokens_saved_bytes_v(B, C, D, H, W, dtype, DLKA_LKA3D_SYNAPSE); }

size_t dlka_lka3d_tokens_workspace_bytes(int B, int C, int D, int H, int W, int dtype) { return dlka_lka3d_tokens_workspace_bytes_v(B, C, D, H, W, dtype, DLKA_LKA3D_SYNAPSE); }
size_t dlka_lka3d_tokens_workspace_bytes_v(int B, int C, int D, int H, int W, int dtype, int variant)
{
    if (!dlka_lka3d_tokens_supported_v(B, C, D, H, W, dtype, variant)) return 0;
    TokGeoms G(B, C, D, H, W, dtype, variant);   // (the eight gradient buffers keep their fp32 size on the bf16 path: gta and the split scratch ARE fp32)
    return align256(G.wp_floats() * 4) + align256(G.part_floats() * 4) + 8 * align256(G.E * 4) + align256(G.GOff * 4) +
           align256(G.scratch_floats() * 4) + align256(G.samp_capacity_floats() * 4) + (cl_dwconv_lds_mode() ? 2 * align256(G.blk_floats() * 4) : 0) + align256(4096) +
           dense_wgrad_pad_bytes(G.offc) +   // the zero-padded copy of t the offset conv's weight gradient reads (round 5)
           align256(deform_fwd_slab_floats(G.dcn) * 4);   // forward pass, small stages: the deformable conv's tap-range slabs, at the END of the workspace (round 6)
}

// x_f32 (DLKA_BF16 only, optional): the caller's UNROUNDED fp32 twin of the bf16 input x.  The chain that decides where the deformable conv samples then starts
// from it (a32 = GELU(proj_1 x_f32) by one extra fp32 pointwise launch) instead of from the bf16 tensor: the wrapper block's mixed mode rounds LayerNorm's output
// INSIDE the block, and 2^-9 of input rounding in front of floor() would flip sampling cells against the fp32 block (tests/parity.py::check_tblock3d_mixed_bf16).
static int tokens_forward_impl(const void *x_, const dlka_lka3d_params *p, void *y_, void *saved, size_t saved_bytes, void *workspace,
                               size_t workspace_bytes, int B, int C, int D, int H, int W, int dtype, void *stream, bool prepared,
                               int variant = DLKA_LKA3D_SYNAPSE, const float *x_f32 = nullptr)
{
    if (!x_ || !p || !y_ || !saved || !workspace) return DLKA_ERR_NULL;
    const void *const *pp = (const void *const *)p;
    for (size_t k = 0; k < sizeof(*p) / sizeof(void *); ++k) if (!pp[k]) return DLKA_ERR_NULL;
    if (!dlka_lka3d_tokens_supported_v(B, C, D, H, W, dtype, variant)) return DLKA_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // DLKA_BF16: x, y and every saved activation are bf16 storage (`float *` below is then just an address: the kernels reinterpret it);
    // offsets, prepared weights, parameters and all accumulators are fp32
    TokGeoms G(B, C, D, H, W, dtype, variant);
    const size_t SB = G.SB;
    Carver sv(saved, saved_bytes), cv(workspace, workspace_bytes);
    float *h = (float *)sv.take(G.E * SB), *a = (float *)sv.take(G.E * SB), *t1 = (float *)sv.take(G.E * SB), *t = (float *)sv.take(G.E * SB);
    float *off = (float *)sv.take(G.Off * 4), *f = (float *)sv.take(G.E * SB), *g1 = (float *)sv.take(G.E * SB);
    float *prep = (float *)sv.take(G.prep_floats() * 4);
    float *m = (float *)sv.take(G.E * SB);   // gate output, kept: proj_2's weight gradient needs it
    (void)cv.take(G.wp_floats() * 4);
    (void)cv.take(G.part_floats() * 4);
    (void)cv.take(G.E * 4);   // (layout kept: the first of the backward pass's eight gradient buffers)
    // bf16 path: the fp32 offset-determining chain (TokGeoms): a32, t1_32, t_32 — the next three of the backward pass's gradient buffers
    float *a32 = (float *)cv.take(G.E * 4), *t1_32 = (float *)cv.take(G.E * 4), *t_32 = (float *)cv.take(G.E * 4);
    // blocked inputs of the opt-in LDS-brick depthwise convs (DLKA_DW_LDS): sized and carved only when that mode is on — and only when both fit, so a mode
    // switched on between the size query and this call keeps the register-row kernels instead of overrunning the workspace
    const bool want_blk = cl_dwconv_lds_mode() != 0;
    float *blkA = (float *)cv.take_opt(G.blk_floats() * 4, want_blk), *blkB = (float *)cv.take_opt(G.blk_floats() * 4, want_blk && blkA);
    if (!blkB) blkA = nullptr;
    if (!sv.ok() || !cv.ok()) return DLKA_ERR_WORKSPACE;
    // the deformable conv's slabs (small stages) live at the END of the workspace: behind everything either pass carves from the front
    const size_t slab_bytes = align256(deform_fwd_slab_floats(G.dcn) * 4);
    if (slab_bytes && workspace_bytes < dlka_lka3d_tokens_workspace_bytes_v(B, C, D, H, W, dtype, variant)) return DLKA_ERR_WORKSPACE;
    float *slab = slab_bytes ? (float *)((char *)workspace + dlka_lka3d_tokens_workspace_bytes_v(B, C, D, H, W, dtype, variant) - slab_bytes) : nullptr;
    const bool bf = dtype == DLKA_BF16;
    const float *x = (const float *)x_;
    float *y = (float *)y_;
    const float *N0 = nullptr;
    // every weight re-layout of the block (forward and backward forms) in one launch; the backward call reuses them
    // outputs of tap-split convs (small stages) collect partial sums with atomics: their zero fills ride in the weight-preparation launch
    ZeroBatch zb;
    memset(&zb, 0, sizeof(zb));
    if (dense_forward_splits(G.offc_f, 0) > 1) zb.add(off, G.Off);
    if (dense_forward_splits(G.pw, 3) > 1) zb.add(y, G.E);
    TokPrep PW;
    const ZeroBatch *ride = nullptr;
    if (prepared) {   // the prepared weights are already in `saved` (dlka_lka3d_tokens_prepare_run): the zero fills ride in the first kernel
        DLKA_TRY(carve_prep(G, prep, PW, p, st, false));
        if (zb.n) ride = &zb;
    } else {
        DLKA_TRY(carve_prep(G, prep, PW, p, st, true, &zb));
    }
    // proj_1 + GELU (transformerblock.py:667-668): h kept for the GELU gradient, a = GELU(h)   (bf16: + the unrounded a for the fp32 chain)
    if (bf && x_f32) {   // (t_32 is free until the dilated conv writes it: the fp32 pre-activation lands there and is never read)
        DLKA_TRY(dense_forward(G.pw_f, x_f32, N0, (const float *)p->proj_1_b, t_32, 0, PW.pw_f[0], 1, nullptr, a32, st, false, ride));
        ride = nullptr;
        DLKA_TRY(dense_forward(G.pw, x, N0, (const float *)p->proj_1_b, h, 0, PW.pw_f[0], 1, nullptr, a, st));
    } else
        DLKA_TRY(dense_forward(G.pw, x, N0, (const float *)p->proj_1_b, h, 0, PW.pw_f[0], 1, nullptr, a, st, false, ride, bf ? a32 : nullptr));
    // depthwise 5^3 then 7^3 dilation 3 (:646-647)   (bf16: fp32 in / out, the bf16 copies t1 / t ride in the same kernels)
    const float *a_in = bf ? a32 : a;
    float *t1_out = bf ? t1_32 : t1, *t_out = bf ? t_32 : t;
    bool chained = false;
    DwBlk bk5, bk7;
    bk5.blk = blkA; bk5.blk_floats = G.blk_floats(); bk5.chain = &G.dw7_f; bk5.chain_blk = blkB; bk5.chained = &chained;
    const int pair = dw_pair(G.dw5_f, G.dw7_f, a_in, PW.dw5_f, (const float *)p->conv0_b, t1_out, bf ? t1 : nullptr, PW.dw7_f, (const float *)p->conv_spatial_b, t_out,
                             bf ? t : nullptr, nullptr, nullptr, st);   // 8^3 / 4^3 stages: both in one launch
    if (pair != DLKA_ERR_UNSUPPORTED) DLKA_TRY(pair);
    else {
        DLKA_TRY(dw_forward(G.dw5_f, a_in, N0, (const float *)p->conv0_b, t1_out, PW.dw5_f, 0, st, nullptr, nullptr, bf ? t1 : nullptr, &bk5));
        bk7.blk = blkB; bk7.blk_floats = G.blk_floats(); bk7.in_blocked = chained;
        DLKA_TRY(dw_forward(G.dw7_f, t1_out, N0, (const float *)p->conv_spatial_b, t_out, PW.dw7_f, 0, st, nullptr, nullptr, bf ? t : nullptr, &bk7));
    }
    // offset-predict conv C -> 81 (synapse/deform_conv.py:94) on the fp32 t; offsets stay in the reference's planar layout
    DLKA_TRY(dense_forward(G.offc_f, t_out, N0, (const float *)p->offset_b, off, 1, PW.off_f, 0, nullptr, nullptr, st, true));
    // deformable 3^3 conv (deform_conv.py:95-105)   (bf16: samples the bf16 copy of t)
    DLKA_TRY(deform_forward(G.dcn, t, off, N0, (const float *)p->deform_b, f, PW.dcn_f, st, slab));
    // conv1 + gate u*attn (:650-652): g1 kept, m = a * g1
    // ... and proj_2 + shortcut (:670-671) — one launch at C <= 64 (cl_pointwise_pair_kernel)
    const int prc = pointwise_pair(G.pw, 0, f, PW.pw_f[1], (const float *)p->conv1_b, PW.pw_f[2], (const float *)p->proj_2_b, a, x, g1, m, y, st);
    if (prc != DLKA_ERR_UNSUPPORTED) return prc;
    DLKA_TRY(dense_forward(G.pw, f, N0, (const float *)p->conv1_b, g1, 0, PW.pw_f[1], 2, a, m, st));
    DLKA_TRY(dense_forward(G.pw, m, N0, (const float *)p->proj_2_b, y, 0, PW.pw_f[2], 3, x, nullptr, st, true));
    return DLKA_OK;
}

int dlka_lka3d_attention_tokens_forward(const void *x, const dlka_lka3d_params *p, void *y, void *saved, size_t saved_bytes, void *workspace,
                                        size_t workspace_bytes, int B, int C, int D, int H, int W, int dtype, void *stream)
{
    return tokens_forward_impl(x, p, y, saved, saved_bytes, workspace, workspace_bytes, B, C, D, H, W, dtype, stream, false);
}

int dlka_lka3d_attention_tokens_forward_v(const void *x, const dlka_lka3d_params *p, void *y, void *saved, size_t saved_bytes, void *workspace,
                                          size_t workspace_bytes, int B, int C, int D, int H, int W, int dtype, int variant, void *stream)
{
    return tokens_forward_impl(x, p, y, saved, saved_bytes, workspace, workspace_bytes, B, C, D, H, W, dtype, stream, false, variant);
}

int dlka_lka3d_attention_tokens_forward_prepared(const void *x, const dlka_lka3d_params *p, void *y, void *saved, size_t saved_bytes, void *workspace,
                                                 size_t workspace_bytes, int B, int C, int D, int H, int W, int dtype, void *stream)
{
    return tokens_forward_impl(x, p, y, saved, saved_bytes, workspace, workspace_bytes, B, C, D, H, W, dtype, stream, true);
}

// ---- weight preparation of MANY blocks in one launch --------------------------------------------------------------------------------
// plan (host, then copied to the device by the caller): [header][int first[MAXJ + 1]: first workgroup of job j][int blkjob[nblocks + 1]: first job of block k][PrepJob jobs[MAXJ]]
namespace {
constexpr int PLAN_JOBS_PER_BLOCK = 16;   // (14 today: 6 pointwise, 2 offset conv, 2 deformable, 4 depthwise forms)
struct PlanHeader { int njobs, nblocks, pad0, pad1; };
size_t plan_first_off() { return sizeof(PlanHeader); }
size_t plan_blkjob_off(int nb) { return sizeof(PlanHeader) + ((size_t)nb * PLAN_JOBS_PER_BLOCK + 1) * sizeof(int); }
size_t plan_jobs_off(int nb) { return align256(plan_blkjob_off(nb) + (size_t)(nb + 1) * sizeof(int)); }
}  // namespace

size_t dlka_lka3d_tokens_prepare_plan_bytes(int nblocks)
{
    if (nblocks <= 0) return 0;
    return plan_jobs_off(nblocks) + (size_t)nblocks * PLAN_JOBS_PER_BLOCK * sizeof(PrepJob);
}

int dlka_lka3d_tokens_prepare_plan(int nblocks, const dlka_lka3d_params *params, void *const *saved, const size_t *saved_bytes, const int *dims5,
                                   int dtype, void *plan_host, size_t plan_bytes)
{
    if (!params || !saved || !saved_bytes || !dims5 || !plan_host) return DLKA_ERR_NULL;
    if (nblocks <= 0 || plan_bytes < dlka_lka3d_tokens_prepare_plan_bytes(nblocks)) return DLKA_ERR_WORKSPACE;
    unsigned char *base = (unsigned char *)plan_host;
    PlanHeader *hd = (PlanHeader *)base;
    int *first = (int *)(base + plan_first_off());
    PrepJob *jobs = (PrepJob *)(base + plan_jobs_off(nblocks));
    int *blkjob = (int *)(base + plan_blkjob_off(nblocks));
    int nj = 0, nb = 0;
    for (int k = 0; k < nblocks; ++k) {
        blkjob[k] = nj;
        const int B = dims5[5 * k], C = dims5[5 * k + 1], D = dims5[5 * k + 2], H = dims5[5 * k + 3], W = dims5[5 * k + 4];
        if (!dlka_lka3d_tokens_supported(B, C, D, H, W, dtype)) return DLKA_ERR_UNSUPPORTED;
        TokGeoms G(B, C, D, H, W, dtype);
        Carver sv(saved[k], saved_bytes[k]);
        for (int e = 0; e < 4; ++e) (void)sv.take(G.E * G.SB);   // h, a, t1, t
        (void)sv.take(G.Off * 4);
        (void)sv.take(G.E * G.SB); (void)sv.take(G.E * G.SB);     // f, g1
        float *prep = (float *)sv.take(G.prep_floats() * 4);
        if (!sv.ok()) return DLKA_ERR_WORKSPACE;
        TokPrep PW;
        PrepBatch pb;
        DLKA_TRY(carve_prep(G, prep, PW, &params[k], nullptr, true, nullptr, &pb));
        if (pb.njobs > PLAN_JOBS_PER_BLOCK) return DLKA_ERR_UNSUPPORTED;
        for (int j = 0; j < pb.njobs; ++j) {
            first[nj] = nb;
            jobs[nj] = pb.j[j];
            nb += cl_prep_table_blocks(pb.j[j]);
            ++nj;
        }
    }
    blkjob[nblocks] = nj;
    first[nj] = nb;   // (end marker: the workgroups of job j are first[j] .. first[j + 1])
    hd->njobs = nj; hd->nblocks = nb; hd->pad0 = hd->pad1 = 0;
    return DLKA_OK;
}

int dlka_lka3d_tokens_prepare_run_range(const void *plan_device, const void *plan_host, int nblocks, int block_lo, int block_hi, void *stream)
{
    if (!plan_device || !plan_host) return DLKA_ERR_NULL;
    if (block_lo < 0 || block_hi > nblocks || block_lo >= block_hi) return DLKA_ERR_SHAPE;
    const unsigned char *hst = (const unsigned char *)plan_host;   // (the counts are read from the host copy: no device round trip)
    const int *first = (const int *)(hst + plan_first_off()), *blkjob = (const int *)(hst + plan_blkjob_off(nblocks));
    const int jlo = blkjob[block_lo], jhi = blkjob[block_hi];
    const unsigned char *dev = (const unsigned char *)plan_device;
    return launch_cl_prep_table((const PrepJob *)(dev + plan_jobs_off(nblocks)), (const int *)(dev + plan_first_off()), jlo, jhi, first[jhi] - first[jlo],
                                (hipStream_t)stream);
}

int dlka_lka3d_tokens_prepare_run(const void *plan_device, const void *plan_host, int nblocks, void *stream)
{
    return dlka_lka3d_tokens_prepare_run_range(plan_device, plan_host, nblocks, 0, nblocks, stream);
}

int dlka_lka3d_attention_tokens_backward(const void *x_, const dlka_lka3d_params *p, const void *gy_, const void *saved, size_t saved_bytes,
                                         void *gx_, const dlka_lka3d_grads *gr, void *workspace, size_t workspace_bytes, int B, int C,
                                         int D, int H, int W, int dtype, void *stream)
{
    return dlka_lka3d_attention_tokens_backward_v(x_, p, gy_, saved, saved_bytes, gx_, gr, workspace, workspace_bytes, B, C, D, H, W, dtype, DLKA_LKA3D_SYNAPSE, stream);
}

// ---- weight-gradient finalisation of MANY blocks in one launch (include/dlka.h: dlka_wgrad_finalize_*) --------------------------------
// plan (host; the caller copies it to the device once): [FinPlanHeader][int first_job[nblocks + 1]][FinalizeJob jobs[FIN_JOBS_PER_BLOCK * nblocks]]
// jobs are stored block after block (block k: first_job[k] .. first_job[k + 1]), block0 = running workgroup count.
namespace {
constexpr int FIN_JOBS_PER_BLOCK = 8;
struct FinPlanHeader { int nblocks, sealed, pad0, pad1; long total_blocks; long pad2; };
size_t fin_first_off() { return sizeof(FinPlanHeader); }
size_t fin_jobs_off(int nb) { return align256(sizeof(FinPlanHeader) + (size_t)(nb + 1) * sizeof(int)); }
int tokens_backward_impl(const void *x_, const dlka_lka3d_params *p, const void *gy_, const void *saved, size_t saved_bytes, void *gx_,
                         const dlka_lka3d_grads *gr, void *workspace, size_t workspace_bytes, int B, int C, int D, int H, int W, int dtype, int variant,
                         void *stream, void *partials, size_t partials_bytes, FinalizeJob *jobs_out, int *njobs_out, int phase = 0,
                         const FinalizeBatch *extra = nullptr);
}  // namespace

size_t dlka_lka3d_tokens_partials_bytes_v(int B, int C, int D, int H, int W, int dtype, int variant)
{
    if (!dlka_lka3d_tokens_supported_v(B, C, D, H, W, dtype, variant)) return 0;
    TokGeoms G(B, C, D, H, W, dtype, variant);
    return align256(G.part_floats() * 4);
}

size_t dlka_wgrad_finalize_plan_bytes(int nblocks)
{
    if (nblocks <= 0) return 0;
    return fin_jobs_off(nblocks) + (size_t)nblocks * FIN_JOBS_PER_BLOCK * sizeof(FinalizeJob);
}

int dlka_wgrad_finalize_plan_init(void *plan_host, size_t plan_bytes, int nblocks)
{
    if (!plan_host) return DLKA_ERR_NULL;
    if (nblocks <= 0 || plan_bytes < dlka_wgrad_finalize_plan_bytes(nblocks)) return DLKA_ERR_WORKSPACE;
    memset(plan_host, 0, dlka_wgrad_finalize_plan_bytes(nblocks));
    ((FinPlanHeader *)plan_host)->nblocks = nblocks;
    return DLKA_OK;
}

int dlka_lka3d_attention_tokens_backward_deferred_v(const void *x, const dlka_lka3d_params *p, const void *grad_y, const void *saved, size_t saved_bytes,
                                                    void *grad_x, const dlka_lka3d_grads *grads, void *workspace, size_t workspace_bytes, void *partials,
                                                    size_t partials_bytes, void *plan_host, int plan_slot, int B, int C, int D, int H, int W, int dtype,
                                                    int variant, void *stream)
{
    if (!partials) return DLKA_ERR_NULL;
    FinalizeJob jobs[FIN_JOBS_PER_BLOCK];
    int nj = 0;
    DLKA_TRY(tokens_backward_impl(x, p, grad_y, saved, saved_bytes, grad_x, grads, workspace, workspace_bytes, B, C, D, H, W, dtype, variant, stream, partials,
                                  partials_bytes, jobs, &nj));
    if (plan_host) {   // record this block's jobs (slots must be filled in ascending order, every slot once, before _plan_seal)
        FinPlanHeader *hd = (FinPlanHeader *)plan_host;
        if (plan_slot < 0 || plan_slot >= hd->nblocks || hd->sealed) return DLKA_ERR_SHAPE;
        int *first = (int *)((unsigned char *)plan_host + fin_first_off());
        FinalizeJob *all = (FinalizeJob *)((unsigned char *)plan_host + fin_jobs_off(hd->nblocks));
        // slots may be recorded in any order: block k owns the fixed window [k * FIN_JOBS_PER_BLOCK, ..); sealing compacts them
        for (int j = 0; j < nj; ++j) all[plan_slot * FIN_JOBS_PER_BLOCK + j] = jobs[j];
        first[plan_slot] = nj;   // (count until sealed)
    }
    return DLKA_OK;
}

int dlka_lka3d_attention_tokens_backward_phase_v(const void *x, const dlka_lka3d_params *p, const void *grad_y, const void *saved, size_t saved_bytes,
                                                 void *grad_x, const dlka_lka3d_grads *grads, void *workspace, size_t workspace_bytes, void *partials,
                                                 size_t partials_bytes, void *plan_host, int plan_slot, int phase, int B, int C, int D, int H, int W, int dtype,
                                                 int variant, void *stream)
{
    if (!partials) return DLKA_ERR_NULL;
    if (phase != 1 && phase != 2) return DLKA_ERR_SHAPE;
    FinalizeJob jobs[FIN_JOBS_PER_BLOCK];
    int nj = 0;
    DLKA_TRY(tokens_backward_impl(x, p, grad_y, saved, saved_bytes, grad_x, grads, workspace, workspace_bytes, B, C, D, H, W, dtype, variant, stream, partials,
                                  partials_bytes, jobs, &nj, phase));
    if (plan_host && phase == 2) {
        FinPlanHeader *hd = (FinPlanHeader *)plan_host;
        if (plan_slot < 0 || plan_slot >= hd->nblocks || hd->sealed) return DLKA_ERR_SHAPE;
        int *first = (int *)((unsigned char *)plan_host + fin_first_off());
        FinalizeJob *all = (FinalizeJob *)((unsigned char *)plan_host + fin_jobs_off(hd->nblocks));
        for (int j = 0; j < nj; ++j) all[plan_slot * FIN_JOBS_PER_BLOCK + j] = jobs[j];
        first[plan_slot] = nj;
    }
    return DLKA_OK;
}

int dlka_wgrad_finalize_run_slot(const void *plan_host, int plan_slot, void *stream)
{
    if (!plan_host) return DLKA_ERR_NULL;
    const FinPlanHeader *hd = (const FinPlanHeader *)plan_host;
    if (hd->sealed || plan_slot < 0 || plan_slot >= hd->nblocks) return DLKA_ERR_SHAPE;
    const int *first = (const int *)((const unsigned char *)plan_host + fin_first_off());
    const FinalizeJob *all = (const FinalizeJob *)((const unsigned char *)plan_host + fin_jobs_off(hd->nblocks));
    const int cnt = first[plan_slot];
    if (cnt <= 0 || cnt > FIN_JOBS_PER_BLOCK) return DLKA_ERR_SHAPE;
    FinalizeBatch fb;
    memset(&fb, 0, sizeof(fb));
    for (int j = 0; j < cnt; ++j) fb.j[j] = all[plan_slot * FIN_JOBS_PER_BLOCK + j];
    fb.njobs = cnt;
    return launch_cl_wgrad_finalize(fb, (hipStream_t)stream);
}

int dlka_wgrad_finalize_plan_seal(void *plan_host)
{
    if (!plan_host) return DLKA_ERR_NULL;
    FinPlanHeader *hd = (FinPlanHeader *)plan_host;
    if (hd->sealed) return DLKA_OK;
    const int nb = hd->nblocks;
    int *first = (int *)((unsigned char *)plan_host + fin_first_off());
    FinalizeJob *all = (FinalizeJob *)((unsigned char *)plan_host + fin_jobs_off(nb));
    int nj = 0;
    long blk = 0;
    for (int k = 0; k < nb; ++k) {
        const int cnt = first[k];
        if (cnt <= 0 || cnt > FIN_JOBS_PER_BLOCK) return DLKA_ERR_SHAPE;   // a slot was never recorded
        first[k] = nj;
        for (int j = 0; j < cnt; ++j) {
            FinalizeJob jb = all[k * FIN_JOBS_PER_BLOCK + j];
            jb.block0 = blk;
            blk += cl_wgrad_finalize_plan_job(jb);
            all[nj++] = jb;   // (nj <= k * FIN_JOBS_PER_BLOCK + j: compaction never overtakes the reads)
        }
    }
    first[nb] = nj;
    hd->total_blocks = blk;
    hd->sealed = 1;
    return DLKA_OK;
}

int dlka_wgrad_finalize_run(const void *plan_device, const void *plan_host, int block_lo, int block_hi, void *stream)
{
    if (!plan_device || !plan_host) return DLKA_ERR_NULL;
    const FinPlanHeader *hd = (const FinPlanHeader *)plan_host;   // (counts are read from the host copy: no device round trip)
    if (!hd->sealed || block_lo < 0 || block_hi > hd->nblocks || block_lo >= block_hi) return DLKA_ERR_SHAPE;
    const int *first = (const int *)((const unsigned char *)plan_host + fin_first_off());
    const FinalizeJob *all_h = (const FinalizeJob *)((const unsigned char *)plan_host + fin_jobs_off(hd->nblocks));
    const int jlo = first[block_lo], jhi = first[block_hi];
    const long b_lo = all_h[jlo].block0;
    const long b_hi = block_hi == hd->nblocks ? hd->total_blocks : all_h[jhi].block0;
    const FinalizeJob *all_d = (const FinalizeJob *)((const unsigned char *)plan_device + fin_jobs_off(hd->nblocks));
    return launch_cl_wgrad_finalize_table(all_d, jlo, jhi, b_hi - b_lo, (hipStream_t)stream);
}

int dlka_lka3d_attention_tokens_backward_v(const void *x_, const dlka_lka3d_params *p, const void *gy_, const void *saved, size_t saved_bytes,
                                           void *gx_, const dlka_lka3d_grads *gr, void *workspace, size_t workspace_bytes, int B, int C,
                                           int D, int H, int W, int dtype, int variant, void *stream)
{
    return tokens_backward_impl(x_, p, gy_, saved, saved_bytes, gx_, gr, workspace, workspace_bytes, B, C, D, H, W, dtype, variant, stream, nullptr, 0, nullptr,
                                nullptr);
}

namespace {
// partials != nullptr: the weight gradients' partial sums go to that (block-private) area and the finalisation is NOT launched — its jobs are
// returned in jobs_out / njobs_out for dlka_wgrad_finalize_run
int tokens_backward_impl(const void *x_, const dlka_lka3d_params *p, const void *gy_, const void *saved, size_t saved_bytes, void *gx_,
                         const dlka_lka3d_grads *gr, void *workspace, size_t workspace_bytes, int B, int C, int D, int H, int W, int dtype, int variant,
                         void *stream, void *partials, size_t partials_bytes, FinalizeJob *jobs_out, int *njobs_out, int phase, const FinalizeBatch *extra)
{
    if (!x_ || !p || !gy_ || !saved || !gx_ || !gr || !workspace) return DLKA_ERR_NULL;
    if (phase < 0 || phase > 2) return DLKA_ERR_SHAPE;
    const void *const *pp = (const void *const *)p;
    for (size_t k = 0; k < sizeof(*p) / sizeof(void *); ++k) if (!pp[k]) return DLKA_ERR_NULL;
    void *const *gp = (void *const *)gr;
    for (size_t k = 0; k < sizeof(*gr) / sizeof(void *); ++k) if (!gp[k]) return DLKA_ERR_NULL;
    if (!dlka_lka3d_tokens_supported_v(B, C, D, H, W, dtype, variant)) return DLKA_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // DLKA_BF16: x, gy, gx, the saved activations and the intermediate gradients gg1, ga1, gf, gt, gt1, gh are bf16 storage; grad_offset, the
    // deformable conv's grad_input accumulator gta (atomics), the weight-gradient partials and the parameter gradients are fp32
    TokGeoms G(B, C, D, H, W, dtype, variant);
    const size_t SB = G.SB;
    const bool bf = dtype == DLKA_BF16;
    Carver sv((void *)saved, saved_bytes), cv(workspace, workspace_bytes);
    const float *h = (float *)sv.take(G.E * SB), *a = (float *)sv.take(G.E * SB), *t1 = (float *)sv.take(G.E * SB), *t = (float *)sv.take(G.E * SB);
    const float *off = (float *)sv.take(G.Off * 4), *f = (float *)sv.take(G.E * SB), *g1 = (float *)sv.take(G.E * SB);
    float *prep = (float *)sv.take(G.prep_floats() * 4);   // written by the matching forward call
    const float *m = (const float *)sv.take(G.E * SB);
    (void)cv.take(G.wp_floats() * 4);
    float *part = (float *)cv.take(G.part_floats() * 4);
    if (partials) {
        if (partials_bytes < G.part_floats() * 4) return DLKA_ERR_WORKSPACE;
        part = (float *)partials;
    }
    // every intermediate gradient has its own buffer: the weight-gradient stream reads them while the data-gradient chain moves on
    float *gg1 = (float *)cv.take(G.E * 4), *ga1 = (float *)cv.take(G.E * 4), *gf = (float *)cv.take(G.E * 4), *gta = (float *)cv.take(G.E * 4);
    float *gt = (float *)cv.take(G.E * 4), *gt1 = (float *)cv.take(G.E * 4), *ga2 = (float *)cv.take(G.E * 4), *gh = (float *)cv.take(G.E * 4);
    float *goff = (float *)cv.take(G.GOff * 4);
    float *scratch = (float *)cv.take(G.scratch_floats() * 4);
    // the sample area is ALWAYS carved at its capacity (the layout behind it does not depend on the gather switch, which is read once here)
    float *samp = G.samp_capacity_floats() ? (float *)cv.take(G.samp_capacity_floats() * 4) : nullptr;
    if (wgrad_gather()) samp = nullptr;
    const bool want_blk = cl_dwconv_lds_mode() != 0;   // (see the forward pass)
    float *blkA = (float *)cv.take_opt(G.blk_floats() * 4, want_blk), *blkB = (float *)cv.take_opt(G.blk_floats() * 4, want_blk && blkA);
    if (!blkB) blkA = nullptr;
    float *padt = (float *)cv.take_opt(dense_wgrad_pad_bytes(G.offc), dense_wgrad_pad_bytes(G.offc) != 0);   // (null when the workspace was sized without it: the unpadded kernels)
    if (!sv.ok() || !cv.ok()) return DLKA_ERR_WORKSPACE;
    const float *x = (const float *)x_, *gy = (const float *)gy_;
    float *gx = (float *)gx_;
    const long E = (long)G.E;
    const float *N0 = nullptr;
    TokPrep PW;
    DLKA_TRY(carve_prep(G, prep, PW, p, st, false));
    // partial-sum areas, one per weight gradient; folded by ONE launch at the end
    float *part_p2 = part, *part_c1 = part_p2 + G.part_pw(), *part_p1 = part_c1 + G.part_pw(), *part_off = part_p1 + G.part_pw();
    float *part_dcn = part_off + G.part_off(), *stage5 = part_dcn + G.part_dcn(), *stage7 = stage5 + (size_t)(G.dw5.K + 1) * C;
    FinalizeBatch fb;
    memset(&fb, 0, sizeof(fb));

    // one lease for the call: `side` (opt-in) for the weight gradients, `s` for grad_input beside grad_offset
    const bool want_gx_fork = phase != 2 && gx_fork_wanted((long)G.dcn.M, phase);
    ForkLease lease(st, want_gx_fork || side_stream_wanted());
    const bool fork = lease.has_side();
    hipStream_t ws_ = lease.side();   // stream of the weight gradients (== st without the side stream)
    // `ws_` may use what the main stream has produced so far
    auto publish = [&]() -> int { return fork ? lease.side_publish() : DLKA_OK; };
    // everything that is accumulated into with atomics, zero-filled by ONE launch: the depthwise weight-gradient staging, the
    // deformable conv's grad_input (halo overflow of the LDS windows) and the outputs of tap-split data gradients
    ZeroBatch zb;
    memset(&zb, 0, sizeof(zb));
    zb.add(stage5, G.stage_dw());
    zb.add(gta, G.E);
    // grad_offset has ONE producer and two 27-tap consumers that contract it on the bf16 matrix cores (offset conv data / weight gradient):
    // the producer stores it already split (pack_split2 words, 96 channel planes) unless its channel-sliced variant needs fp32 atomics
    int goff_cpad = 0;
    {
        DeformBwdArgs da;
        fill_deform_bwd(da, G.dcn);
        // Measured with the workgroup-tiled consumers (profiles/archive/r01v): no gain — they are bound by per-unit latency (barrier + staging per
        // 192 MFMA cycles), not by the split arithmetic (grad_offset +14 us, weight gradient +18 us, data gradient unchanged at 32^3) — so the
        // packed hand-over is opt-in (DLKA_GOFF_PACKED=1) until the consumers are wave-granular.
        const bool fp32_goff = bf || getenv("DLKA_GOFF_PACKED") == nullptr;   // (not cached: tests toggle it)
        const bool sliced = (bf || deform_bwd_variant() == 0) && cl_deform_goff_ccsplit(da) > 1;
        if (sliced) zb.add(goff, G.Off);
        // (N % 16: the weight-gradient kernel's split variant exists for 16-voxel-aligned volumes only, cl_wgrad.hip)
        else if (!fp32_goff && deform_bwd_variant() == 0 && use_split(G.offc, false) == 2 && (G.offc.N & 15) == 0) goff_cpad = 96;
    }
    if (dense_backward_data_splits(G.pw, 0) > 1) zb.add(gf, G.E);
    if (dense_backward_data_splits(G.offc, 3) > 1) zb.add(bf ? ga2 : gt, G.E);   // bf16: split partial sums land in the fp32 scratch ga2
    if (dense_backward_data_splits(G.pw, 3) > 1) zb.add(gx, G.E);
    if (zb.overflow) return DLKA_ERR_WORKSPACE;
    // (the zero fills ride in the first kernel below: one dependent node less per block)

    // proj_2:  y = P2 m + x.   Its data gradient gm = P2^T gy feeds only the gate  m = a * g1, whose backward is fused into
    // the epilogue:  gg1 = gm * a,  ga1 = gm * g1
    // (the three pointwise weight gradients run as ONE launch at the end: their operands m/gy, f/gg1, x/gh all stay live)
    // ... and conv1:  g1 = P0 f,  gf = P0^T gg1 — one launch at C <= 64
    // phase: 0 = the whole backward pass; 1 = the DATA-gradient chain only (everything the next block needs: gx; and what the weight gradients read:
    // the intermediate gradients and the stored samples, all in `workspace`); 2 = the five WEIGHT-gradient launches only, reading those — a caller that
    // runs phase 2 on another stream lets them overlap the next block's data chain (DLKABlockStack: two alternating workspaces).
#define DLKA_P1(call) do { if (phase != 2) DLKA_TRY(call); } while (0)
#define DLKA_P2(call) do { if (phase != 1) DLKA_TRY(call); } while (0)
    if (phase != 2) {
        const int prc = dense_backward_data_splits(G.pw, 0) > 1 ? DLKA_ERR_UNSUPPORTED
                                                                : pointwise_pair(G.pw, 1, gy, PW.pw_b[2], nullptr, PW.pw_b[1], nullptr, a, g1, gg1, ga1, gf, st, &zb);
        if (prc != DLKA_ERR_UNSUPPORTED) DLKA_TRY(prc);
        else {
            DLKA_TRY(dense_backward_data(G.pw, gy, 0, N0, gg1, PW.pw_b[2], 4, a, st, g1, ga1, false, false, false, nullptr, &zb));
            DLKA_TRY(publish());   // fork: everything issued so far (gy, saved activations, the zero fills, the previous block's use of the workspace)
            DLKA_TRY(dense_backward_data(G.pw, gg1, 0, N0, gf, PW.pw_b[1], 0, nullptr, st, nullptr, nullptr, true));
        }
    }
    DLKA_TRY(publish());
    // deformable conv:  f = DCN(t, off):  grad_offset and grad_input on the main stream, the weight gradient on the side one — after
    // grad_offset when that kernel hands over the samples it interpolated (samp), else at once with its own gather
    if (!samp)
        DLKA_P2(deform_backward(G.dcn, t, off, N0, gf, nullptr, nullptr, (float *)gr->deform_w, (float *)gr->deform_b, PW.dcn_b, part_dcn, scratch, ws_,
                                &fb.j[fb.njobs++]));
    bool gx_forked = false;
    if (want_gx_fork && lease.ok()) {   // grad_input on the internal stream, beside grad_offset (see ForkCtx)
        DLKA_TRY(lease.fork(1));
        DLKA_TRY(deform_backward(G.dcn, t, off, N0, gf, gta, nullptr, nullptr, nullptr, PW.dcn_b, nullptr, scratch, lease.stream(1), nullptr, true, false, 0, nullptr, PW.dcn_b16));
        gx_forked = true;
    }
    DLKA_P1(deform_backward(G.dcn, t, off, N0, gf, nullptr, goff, nullptr, nullptr, PW.dcn_b, nullptr, scratch, st, nullptr, false, true, goff_cpad, samp, PW.dcn_b16));
    DLKA_TRY(publish());
    if (samp)
        DLKA_P2(deform_backward(G.dcn, t, off, N0, gf, nullptr, nullptr, (float *)gr->deform_w, (float *)gr->deform_b, PW.dcn_b, part_dcn, scratch, ws_,
                                &fb.j[fb.njobs++], false, false, 0, samp));
    // offset-predict conv:  off = Coff t      (gt = gt_a + Coff^T goff fused in the epilogue)
    DLKA_P2(dense_backward_weight(G.offc, t, goff, 1, (float *)gr->offset_w, (float *)gr->offset_b, part_off, ws_, &fb.j[fb.njobs++], goff_cpad, padt));
    if (gx_forked) DLKA_TRY(lease.join(1));   // the offset conv's data gradient adds gta
    else
        DLKA_P1(deform_backward(G.dcn, t, off, N0, gf, gta, nullptr, nullptr, nullptr, PW.dcn_b, nullptr, scratch, st, nullptr, true, false, 0, nullptr, PW.dcn_b16));
    DLKA_P1(dense_backward_data(G.offc, goff, 1, N0, gt, PW.off_b, 3, gta, st, nullptr, nullptr, true, goff_cpad != 0, true, ga2));
    DLKA_TRY(publish());
    // depthwise 7^3 dil 3:  t = DW7 t1
    DLKA_P2(dw_backward_weight(G.dw7, t1, gt, (float *)gr->conv_spatial_w, (float *)gr->conv_spatial_b, stage7, ws_, &fb.j[fb.njobs++]));
    bool chained = false;
    DwBlk bk7, bk5;
    bk7.blk = blkA; bk7.blk_floats = G.blk_floats(); bk7.chain = &G.dw5; bk7.chain_blk = blkB; bk7.chained = &chained;
    // (8^3 / 4^3 stages: this conv's and the next one's data gradients, GELU' included, in one launch)
    int pair = DLKA_ERR_UNSUPPORTED;
    if (phase != 2) pair = dw_pair(G.dw7, G.dw5, gt, PW.dw7_b, nullptr, gt1, nullptr, PW.dw5_b, nullptr, gh, nullptr, h, ga1, st);
    if (pair != DLKA_ERR_UNSUPPORTED) DLKA_TRY(pair);
    else DLKA_P1(dw_forward(G.dw7, gt, N0, nullptr, gt1, PW.dw7_b, 1, st, nullptr, nullptr, nullptr, &bk7));
    DLKA_TRY(publish());
    // depthwise 5^3:  t1 = DW5 a
    DLKA_P2(dw_backward_weight(G.dw5, a, gt1, (float *)gr->conv0_w, (float *)gr->conv0_b, stage5, ws_, &fb.j[fb.njobs++]));
    // ... with the GELU backward in its epilogue:  a = GELU(h),  gh = (ga1 + DW5^T gt1) * gelu'(h)
    (void)E;
    bk5.blk = blkB; bk5.blk_floats = G.blk_floats(); bk5.in_blocked = chained;
    if (pair == DLKA_ERR_UNSUPPORTED) DLKA_P1(dw_forward(G.dw5, gt1, N0, nullptr, gh, PW.dw5_b, 1, st, h, ga1, nullptr, &bk5));
    DLKA_TRY(publish());
    // proj_1:  h = P1 x ;  gx = P1^T gh + gy (shortcut)
    if (phase != 1) {
        WgradArgs jobs[3];
        fill_pw_wgrad(jobs[0], G.pw, m, gy, part_p2);
        fill_pw_wgrad(jobs[1], G.pw, f, gg1, part_c1);
        fill_pw_wgrad(jobs[2], G.pw, x, gh, part_p1);
        float *const gws[3] = {(float *)gr->proj_2_w, (float *)gr->conv1_w, (float *)gr->proj_1_w};
        float *const gbs[3] = {(float *)gr->proj_2_b, (float *)gr->conv1_b, (float *)gr->proj_1_b};
        DLKA_TRY(launch_cl_wgrad_pw3(jobs, gws, gbs, ws_, &fb.j[fb.njobs]));
        fb.njobs += 3;
    }
    if (partials) {
        if (fb.njobs > FIN_JOBS_PER_BLOCK || !jobs_out || !njobs_out) return DLKA_ERR_UNSUPPORTED;
        for (int k = 0; k < fb.njobs; ++k) jobs_out[k] = fb.j[k];
        *njobs_out = fb.njobs;
    } else {
        if (phase != 0) return DLKA_ERR_UNSUPPORTED;   // the split passes need the deferred finalisation (block-private partial sums)
        if (extra) {   // the caller's own folds (the wrapper block's three conv weight gradients) ride in this launch
            if (fb.njobs + extra->njobs > (int)(sizeof(fb.j) / sizeof(fb.j[0]))) return DLKA_ERR_UNSUPPORTED;
            for (int k = 0; k < extra->njobs; ++k) fb.j[fb.njobs++] = extra->j[k];
        }
        DLKA_TRY(launch_cl_wgrad_finalize(fb, ws_));
    }
    DLKA_P1(dense_backward_data(G.pw, gh, 0, N0, gx, PW.pw_b[0], 3, gy, st, nullptr, nullptr, true));
#undef DLKA_P1
#undef DLKA_P2
    if (fork) DLKA_TRY(lease.side_join());
    return DLKA_OK;
}
}  // namespace

// ---- the wrapper block's non-convolutional pieces (cl_norm.hip) ------------------------------------------------------------
int dlka_layernorm_tokens_forward(const void *x, int x_planar, const void *pos, const void *w, const void *b, void *xt, void *xn, void *stats, int B,
                                  int N, int C, float eps, int dtype, void *stream)
{
    if (!x || !w || !b || !xt || !xn || !stats) return DLKA_ERR_NULL;
    if (dtype != DLKA_F32) return DLKA_ERR_UNSUPPORTED;
    if (B <= 0 || N <= 0 || C <= 0) return DLKA_ERR_SHAPE;
    return launch_cl_layernorm_fwd((const float *)x, x_planar, (const float *)pos, (const float *)w, (const float *)b, (float *)xt, (float *)xn,
                                   (float *)stats, B, N, C, eps, (hipStream_t)stream);
}

int dlka_layernorm_tokens_backward(const void *g_xn, const void *g_res, const void *xt, const void *stats, const void *w, void *gxt, void *gw,
                                   void *gb, void *gpos, int B, int N, int C, int dtype, void *stream)
{
    if (!g_xn || !xt || !stats || !w || !gxt || !gw || !gb) return DLKA_ERR_NULL;
    if (dtype != DLKA_F32) return DLKA_ERR_UNSUPPORTED;
    if (B <= 0 || N <= 0 || C <= 0) return DLKA_ERR_SHAPE;
    return launch_cl_layernorm_bwd((const float *)g_xn, (const float *)g_res, (const float *)xt, (const float *)stats, (const float *)w, (float *)gxt,
                                   (float *)gw, (float *)gb, (float *)gpos, B, N, C, (hipStream_t)stream);
}

int dlka_scale_residual_forward(const void *xt, const void *e, const void *gamma, void *out, int64_t M, int C, int dtype, void *stream)
{
    if (!xt || !e || !gamma || !out) return DLKA_ERR_NULL;
    if (dtype != DLKA_F32) return DLKA_ERR_UNSUPPORTED;
    if (M <= 0 || C <= 0) return DLKA_ERR_SHAPE;
    return launch_cl_scale_residual_fwd((const float *)xt, (const float *)e, (const float *)gamma, (float *)out, (long)M, C, (hipStream_t)stream);
}

int dlka_scale_residual_backward(const void *g, const void *e, const void *gamma, void *ge, void *ggamma, int64_t M, int C, int dtype, void *stream)
{
    if (!g || !e || !gamma || !ge || !ggamma) return DLKA_ERR_NULL;
    if (dtype != DLKA_F32) return DLKA_ERR_UNSUPPORTED;
    if (M <= 0 || C <= 0 || C > 1024 || C % 32) return DLKA_ERR_SHAPE;
    return launch_cl_scale_residual_bwd((const float *)g, (const float *)e, (const float *)gamma, (float *)ge, (float *)ggamma, (long)M, C, (hipStream_t)stream);
}

int dlka_batchnorm_cl_forward(const void *x, const void *res, const void *w, const void *b, void *stats, int training, void *y, void *scratch, int64_t M,
                              int C, float eps, float slope, int dtype, void *stream)
{
    if (!x || !w || !b || !stats || !y || !scratch) return DLKA_ERR_NULL;
    if (dtype != DLKA_F32) return DLKA_ERR_UNSUPPORTED;
    if (M <= 0 || C <= 0 || C > 1024 || C % 32) return DLKA_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (training) DLKA_TRY(launch_cl_bn_stats((const float *)x, (float *)scratch, (float *)stats, (long)M, C, eps, st));
    return launch_cl_bn_apply((const float *)x, (const float *)res, (const float *)w, (const float *)b, (const float *)stats, nullptr, (float *)y, (long)M, (long)M, C, slope, st);
}

int dlka_batchnorm_cl_backward(const void *g, const void *x, const void *y, const void *w, const void *stats, int training, void *gx, void *gres,
                               void *gw, void *gb, void *scratch, int64_t M, int C, float slope, int dtype, void *stream)
{
    if (!g || !x || !y || !w || !stats || !gx || !gw || !gb || !scratch) return DLKA_ERR_NULL;
    if (dtype != DLKA_F32) return DLKA_ERR_UNSUPPORTED;
    if (M <= 0 || C <= 0 || C > 1024 || C % 32) return DLKA_ERR_SHAPE;
    return launch_cl_bn_bwd((const float *)g, nullptr, (const float *)x, (const float *)y, (const float *)w, (const float *)stats, (float *)scratch, (float *)gx,
                            (float *)gres, nullptr, (float *)gw, (float *)gb, (long)M, (long)M, C, slope, training, (hipStream_t)stream);
}

// ---- planar (NCDHW) plumbing of the full net (planar_ops.hip) -----------------------------------------------------------------------
int dlka_batchnorm_planar_forward(const void *x, const void *w, const void *b, void *stats, void *y, void *scratch, int B, int C, int64_t N, float eps,
                                  void *stream)
{
    if (!x || !stats || !y || !scratch) return DLKA_ERR_NULL;
    return launch_pl_bn_forward((const float *)x, (const float *)w, (const float *)b, (float *)stats, (float *)y, (float *)scratch, B, C, (long)N, eps,
                                (hipStream_t)stream);
}

int dlka_batchnorm_planar_backward(const void *g, const void *x, const void *w, const void *stats, void *gx, void *gw, void *gb, void *scratch, int B, int C,
                                   int64_t N, void *stream)
{
    if (!g || !x || !stats || !gx || !scratch) return DLKA_ERR_NULL;
    return launch_pl_bn_backward((const float *)g, (const float *)x, (const float *)w, (const float *)stats, (float *)gx, (float *)gw, (float *)gb,
                                 (float *)scratch, B, C, (long)N, (hipStream_t)stream);
}

int dlka_pointwise_planar_forward(const void *x, const void *w, const void *bias, void *y, int B, int Cin, int Cout, int64_t N, void *stream)
{
    if (!x || !w || !y) return DLKA_ERR_NULL;
    return launch_pl_pw_forward((const float *)x, (const float *)w, (const float *)bias, (float *)y, B, Cin, Cout, (long)N, (hipStream_t)stream);
}

int dlka_pointwise_planar_backward(const void *x, const void *w, const void *g, void *gx, void *gw, void *gb, int B, int Cin, int Cout, int64_t N, void *stream)
{
    if (!x || !w || !g) return DLKA_ERR_NULL;
    return launch_pl_pw_backward((const float *)x, (const float *)w, (const float *)g, (float *)gx, (float *)gw, (float *)gb, B, Cin, Cout, (long)N,
                                 (hipStream_t)stream);
}

int dlka_channel_scale(const void *x, const void *mask, void *y, int B, int64_t N, int C, int dtype, void *stream)
{
    if (!x || !mask || !y) return DLKA_ERR_NULL;
    if (dtype != DLKA_F32) return DLKA_ERR_UNSUPPORTED;
    if (B <= 0 || N <= 0 || C <= 0) return DLKA_ERR_SHAPE;
    return launch_cl_channel_scale((const float *)x, (const float *)mask, (float *)y, B, (long)N, C, (hipStream_t)stream);
}


// ---- TransformerBlock_3D_single_deform_LKA, one call per direction -----------------------------------------------------------
// Reference: 3D/d_lka_former/network_architecture/synapse/transformerblock.py:617-630 (forward), dynunet_block.py:66-80
// (UnetResBlock.forward).  Everything stays in token layout [M][C]; the only strided access is the read of an NCDHW input.
namespace {

struct TBlockGeoms {
    SameConv c3, pw;   // the 3^3 dense convs of UnetResBlock, the 1x1x1 conv of conv8
    size_t E, M;
    TBlockGeoms(int B, int C, int D, int H, int W)
    {
        dlka_conv_geom g;
        memset(&g, 0, sizeof(g));
        g.B = B; g.C = C; g.D = D; g.H = H; g.W = W; g.Cout = C;
        g.kd = g.kh = g.kw = 3; g.sd = g.sh = g.sw = 1; g.pd = g.ph = g.pw = 1; g.dd = g.dh = g.dw = 1; g.group = 1; g.deformable_group = 1; g.im2col_step = 64;
        make_same_conv(&g, c3);
        g.kd = g.kh = g.kw = 1; g.pd = g.ph = g.pw = 0;
        make_same_conv(&g, pw);
        M = (size_t)c3.M;
        E = M * C;
    }
    size_t wp_floats() const { return dense_wp_floats(c3); }
    size_t part_floats() const { return cl_wgrad_part_floats(c3.M, c3.K, c3.Cout, c3.Cin); }
};

bool tblock_supported(int B, int C, int D, int H, int W, int variant = DLKA_LKA3D_SYNAPSE) { return tokens_supported(B, C, D, H, W, variant) && (long)D * H * W < (1l << 31); }

struct TBlockSaved {
    float *xt, *xn, *e, *attn, *c1, *a1, *c2, *rd, *lnstats;
    float *w1_f, *w1_b, *w2_f, *w2_b, *w8_f, *w8_b;   // prepared weights (forward / data-gradient forms), written by ONE launch in forward
    void *lka;
    size_t lka_bytes;
};

bool carve_tblock_saved(Carver &sv, const TBlockGeoms &G, int B, int C, int D, int H, int W, TBlockSaved &S, int variant = DLKA_LKA3D_SYNAPSE, int dtype = DLKA_F32)
{
    S.xt = (float *)sv.take(G.E * 4); S.xn = (float *)sv.take(G.E * 4); S.e = (float *)sv.take(G.E * 4); S.attn = (float *)sv.take(G.E * 4);
    S.c1 = (float *)sv.take(G.E * 4); S.a1 = (float *)sv.take(G.E * 4); S.c2 = (float *)sv.take(G.E * 4); S.rd = (float *)sv.take(G.E * 4);
    S.lnstats = (float *)sv.take(G.M * 2 * 4);
    S.w1_f = (float *)sv.take(dense_wp_floats(G.c3) * 4); S.w1_b = (float *)sv.take(dense_wp_floats(G.c3) * 4);
    S.w2_f = (float *)sv.take(dense_wp_floats(G.c3) * 4); S.w2_b = (float *)sv.take(dense_wp_floats(G.c3) * 4);
    S.w8_f = (float *)sv.take(dense_wp_floats(G.pw) * 4); S.w8_b = (float *)sv.take(dense_wp_floats(G.pw) * 4);
    S.lka_bytes = dlka_lka3d_tokens_saved_bytes_v(B, C, D, H, W, dtype, variant);
    S.lka = sv.take(S.lka_bytes);
    return sv.ok();
}

}  // namespace

// dtype = DLKA_BF16 on the wrapper block is MIXED precision: x, y, the residual stream, LayerNorm / BatchNorm statistics, the 3^3 convs of UnetResBlock and
// every wrapper gradient stay fp32 (pointers are fp32 on both dtypes); the D-LKA attention inside (transformerblock.py:624) runs DLKA_BF16 — its input xn, output
// e and their gradients are bf16 storage, with the mixed-precision rule of the token path (fp32 offset-determining chain, fp32 parameters / accumulation).
int dlka_tblock3d_supported_v(int B, int C, int D, int H, int W, int dtype, int variant)
{
    return ((dtype == DLKA_F32 || dtype == DLKA_BF16) && tblock_supported(B, C, D, H, W, variant)) ? 1 : 0;
}
int dlka_tblock3d_supported(int B, int C, int D, int H, int W, int dtype) { return dlka_tblock3d_supported_v(B, C, D, H, W, dtype, DLKA_LKA3D_SYNAPSE); }

size_t dlka_tblock3d_saved_bytes(int B, int C, int D, int H, int W, int dtype) { return dlka_tblock3d_saved_bytes_v(B, C, D, H, W, dtype, DLKA_LKA3D_SYNAPSE); }
size_t dlka_tblock3d_saved_bytes_v(int B, int C, int D, int H, int W, int dtype, int variant)
{
    if (!dlka_tblock3d_supported_v(B, C, D, H, W, dtype, variant)) return 0;
    TBlockGeoms G(B, C, D, H, W);
    return 8 * align256(G.E * 4) + align256(G.M * 2 * 4) + 4 * align256(dense_wp_floats(G.c3) * 4) + 2 * align256(dense_wp_floats(G.pw) * 4) +
           align256(dlka_lka3d_tokens_saved_bytes_v(B, C, D, H, W, dtype, variant));
}

int dlka_lka3d_tokens_saved_offsets_v(int B, int C, int D, int H, int W, int dtype, int variant, size_t *byte_offset)
{
    if (!byte_offset) return DLKA_ERR_NULL;
    if (!dlka_lka3d_tokens_supported_v(B, C, D, H, W, dtype, variant)) return DLKA_ERR_UNSUPPORTED;
    TokGeoms G(B, C, D, H, W, dtype, variant);
    *byte_offset = 4 * align256(G.E * G.SB);   // tokens_forward_impl carves h, a, t1, t, then the offsets
    return DLKA_OK;
}

int dlka_tblock3d_saved_offsets_v(int B, int C, int D, int H, int W, int dtype, int variant, size_t *byte_offset)
{
    if (!byte_offset) return DLKA_ERR_NULL;
    if (!dlka_tblock3d_supported_v(B, C, D, H, W, dtype, variant)) return DLKA_ERR_UNSUPPORTED;
    TBlockGeoms G(B, C, D, H, W);
    size_t inner = 0;
    DLKA_TRY(dlka_lka3d_tokens_saved_offsets_v(B, C, D, H, W, dtype, variant, &inner));
    // carve_tblock_saved: eight activation tensors, the LayerNorm statistics, six prepared weight forms, then the D-LKA block's own `saved`
    *byte_offset = 8 * align256(G.E * 4) + align256(G.M * 2 * 4) + 4 * align256(dense_wp_floats(G.c3) * 4) + 2 * align256(dense_wp_floats(G.pw) * 4) + inner;
    return DLKA_OK;
}

int dlka_tblock3d_saved_activations_v(int B, int C, int D, int H, int W, int dtype, int variant, size_t byte_offsets[2])
{
    if (!byte_offsets) return DLKA_ERR_NULL;
    if (!dlka_tblock3d_supported_v(B, C, D, H, W, dtype, variant)) return DLKA_ERR_UNSUPPORTED;
    TBlockGeoms G(B, C, D, H, W);
    // carve_tblock_saved: xt, xn, e, attn, c1, a1, c2, rd
    byte_offsets[0] = 5 * align256(G.E * 4);
    byte_offsets[1] = 7 * align256(G.E * 4);
    return DLKA_OK;
}

size_t dlka_tblock3d_workspace_bytes(int B, int C, int D, int H, int W, int dtype) { return dlka_tblock3d_workspace_bytes_v(B, C, D, H, W, dtype, DLKA_LKA3D_SYNAPSE); }
size_t dlka_tblock3d_workspace_bytes_v(int B, int C, int D, int H, int W, int dtype, int variant)
{
    if (!dlka_tblock3d_supported_v(B, C, D, H, W, dtype, variant)) return 0;
    TBlockGeoms G(B, C, D, H, W);
    return align256(dlka_lka3d_tokens_workspace_bytes_v(B, C, D, H, W, dtype, variant)) + align256(G.wp_floats() * 4) + 2 * align256(G.part_floats() * 4) +
           align256(cl_wgrad_part_floats(G.pw.M, 1, G.pw.Cout, G.pw.Cin) * 4) + 6 * align256(G.E * 4) + align256(4096) +
           align256(dlka_lka3d_tokens_partials_bytes_v(B, C, D, H, W, dtype, variant));   // (the phased backward's partial sums of the attention: dlka_tblock3d_backward_phase_v)
}

int dlka_tblock3d_forward(const void *x, int x_planar, const dlka_tblock3d_params *p, const dlka_lka3d_params *lka, const void *drop_mask, int training,
                          void *bn_stats, void *y, void *saved, size_t saved_bytes, void *workspace, size_t workspace_bytes, int B, int C, int D, int H, int W,
                          float ln_eps, float bn_eps, int dtype, void *stream)
{
    return dlka_tblock3d_forward_v(x, x_planar, p, lka, drop_mask, training, bn_stats, y, saved, saved_bytes, workspace, workspace_bytes, B, C, D, H, W, ln_eps, bn_eps,
                                   dtype, DLKA_LKA3D_SYNAPSE, stream);
}

int dlka_tblock3d_forward_v(const void *x, int x_planar, const dlka_tblock3d_params *p, const dlka_lka3d_params *lka, const void *drop_mask, int training,
                            void *bn_stats, void *y, void *saved, size_t saved_bytes, void *workspace, size_t workspace_bytes, int B, int C, int D, int H, int W,
                            float ln_eps, float bn_eps, int dtype, int variant, void *stream)
{
    if (!x || !p || !lka || !bn_stats || !y || !saved || !workspace) return DLKA_ERR_NULL;
    if (!p->norm_w || !p->norm_b || !p->gamma || !p->conv51_conv1_w || !p->conv51_conv2_w || !p->conv51_norm1_w || !p->conv51_norm1_b ||
        !p->conv51_norm2_w || !p->conv51_norm2_b || !p->conv8_w || !p->conv8_b)
        return DLKA_ERR_NULL;
    if (!dlka_tblock3d_supported_v(B, C, D, H, W, dtype, variant)) return DLKA_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    TBlockGeoms G(B, C, D, H, W);
    Carver sv(saved, saved_bytes), cv(workspace, workspace_bytes);
    TBlockSaved S;
    if (!carve_tblock_saved(sv, G, B, C, D, H, W, S, variant, dtype)) return DLKA_ERR_WORKSPACE;
    const size_t lka_ws_bytes = dlka_lka3d_tokens_workspace_bytes_v(B, C, D, H, W, dtype, variant);
    void *lka_ws = cv.take(lka_ws_bytes);
    float *wp = (float *)cv.take(G.wp_floats() * 4);
    float *sums = (float *)cv.take(4096);
    const int lo = dtype == DLKA_BF16 ? 1 : 0;   // the D-LKA attention runs DLKA_BF16: xn / e are bf16 storage
    // mixed mode: LayerNorm's unrounded output, for the offset-determining chain of the attention (in the region the backward call uses for its six gradient buffers)
    static const bool xn32_on = [] { const char *e = getenv("DLKA_MIXED_XN32"); return !(e && e[0] == '0'); }();   // (A/B: 0 = the chain starts from the bf16 tensor, as in round 4)
    float *xn32 = (lo && xn32_on) ? (float *)cv.take(G.E * 4) : nullptr;
    if (!cv.ok()) return DLKA_ERR_WORKSPACE;
    const long M = (long)G.M, N = G.c3.N;
    float *st1 = (float *)bn_stats, *st2 = st1 + 3 * C;
    const float slope = 0.01f;   // UnetResBlock's act_name default (dynunet_block.py:41)
    (void)wp;
    // ONE launch prepares the wrapper's six weight forms (kept in `saved` for the backward call) and zero-fills what this direction accumulates
    // into with atomics (BatchNorm sums, tap-split conv outputs)
    {
        PrepBatch pb;
        memset(&pb, 0, sizeof(pb));
        const int f3 = split_mode_flag(use_split(G.c3, true)), b3 = 1 | split_mode_flag(use_split(G.c3, false));
        add_job(pb, p->conv51_conv1_w, S.w1_f, C, C, 27, C, C, f3);
        add_job(pb, p->conv51_conv1_w, S.w1_b, C, C, 27, C, C, b3);
        add_job(pb, p->conv51_conv2_w, S.w2_f, C, C, 27, C, C, f3);
        add_job(pb, p->conv51_conv2_w, S.w2_b, C, C, 27, C, C, b3);
        add_job(pb, p->conv8_w, S.w8_f, C, C, 1, C, C, 0);
        add_job(pb, p->conv8_w, S.w8_b, C, C, 1, C, C, 1);
        auto add_zero = [&](float *ptr, size_t n) { PrepJob &j = pb.j[pb.njobs++]; memset(&j, 0, sizeof(j)); j.dst = ptr; j.n = (long)n; j.mode = 5; pb.total += j.n; };
        add_zero(sums, 1024);
        if (dense_forward_splits(G.c3, 0) > 1) { add_zero(S.c1, G.E); add_zero(S.c2, G.E); }
        if (dense_forward_splits(G.pw, 3) > 1) add_zero((float *)y, G.E);
        // ... and the attention's own fifteen forms go out with them (round 5: one launch per block less in the path the trainers call; the attention's zero fills ride
        // in its first kernel, as they do behind the engine's hoisted preparation)
        const void *const *pp = (const void *const *)lka;
        for (size_t k = 0; k < sizeof(*lka) / sizeof(void *); ++k) if (!pp[k]) return DLKA_ERR_NULL;
        TokGeoms TG(B, C, D, H, W, dtype, variant);
        Carver lsv(S.lka, S.lka_bytes);   // (the layout tokens_forward_impl carves: h, a, t1, t, offsets, f, g1, then the prepared weights)
        for (int e = 0; e < 4; ++e) (void)lsv.take(TG.E * TG.SB);
        (void)lsv.take(TG.Off * 4);
        (void)lsv.take(TG.E * TG.SB); (void)lsv.take(TG.E * TG.SB);
        float *lprep = (float *)lsv.take(TG.prep_floats() * 4);
        if (!lsv.ok()) return DLKA_ERR_WORKSPACE;
        TokPrep PWl;
        DLKA_TRY(carve_prep(TG, lprep, PWl, lka, st, true, nullptr, &pb, true));
        DLKA_TRY(launch_cl_prep_batch(pb, st));
    }
    // tokens (+ pos_embed) and LayerNorm (:620-624)
    DLKA_TRY(launch_cl_layernorm_fwd((const float *)x, x_planar, (const float *)p->pos_embed, (const float *)p->norm_w, (const float *)p->norm_b, S.xt, S.xn,
                                     S.lnstats, B, (int)N, C, ln_eps, st, lo, xn32));
    // epa_block = the D-LKA block (:624)
    DLKA_TRY(tokens_forward_impl(S.xn, lka, S.e, S.lka, S.lka_bytes, lka_ws, lka_ws_bytes, B, C, D, H, W, dtype, stream, true, variant, xn32));   // (prepared above)
    // attn = x + gamma * epa (:624); attn IS attn_skip in channels-last memory (:626 is a view here)
    DLKA_TRY(launch_cl_scale_residual_fwd(S.xt, S.e, (const float *)p->gamma, S.attn, M, C, st, lo));
    // conv51 = UnetResBlock (dynunet_block.py:66-80)
    DLKA_TRY(dense_forward(G.c3, S.attn, nullptr, nullptr, S.c1, 0, S.w1_f, 0, nullptr, nullptr, st, true));
    // (batch statistics in their deterministic form: the attention's workspace is free again and serves as the per-workgroup partial-sum scratch)
    if (training) DLKA_TRY(launch_cl_bn_stats(S.c1, sums, st1, M, C, bn_eps, st, true, (float *)lka_ws, lka_ws_bytes / 4));
    DLKA_TRY(launch_cl_bn_apply(S.c1, nullptr, (const float *)p->conv51_norm1_w, (const float *)p->conv51_norm1_b, st1, nullptr, S.a1, M, N, C, slope, st));
    DLKA_TRY(dense_forward(G.c3, S.a1, nullptr, nullptr, S.c2, 0, S.w2_f, 0, nullptr, nullptr, st, true));
    if (training) DLKA_TRY(launch_cl_bn_stats(S.c2, sums + 512, st2, M, C, bn_eps, st, true, (float *)lka_ws, lka_ws_bytes / 4));
    // ... + residual, LeakyReLU, and conv8[0] = Dropout3d folded into the same pass (:611)
    DLKA_TRY(launch_cl_bn_apply(S.c2, S.attn, (const float *)p->conv51_norm2_w, (const float *)p->conv51_norm2_b, st2, (const float *)drop_mask, S.rd, M, N, C, slope, st));
    // x = attn_skip + conv8(attn) (:628)
    DLKA_TRY(dense_forward(G.pw, S.rd, nullptr, (const float *)p->conv8_b, (float *)y, 0, S.w8_f, 3, S.attn, nullptr, st, true));
    return DLKA_OK;
}

int dlka_tblock3d_backward(const dlka_tblock3d_params *p, const dlka_lka3d_params *lka, const void *drop_mask, int training, const void *bn_stats,
                           const void *grad_y, const void *saved, size_t saved_bytes, void *grad_x, const dlka_tblock3d_grads *gr,
                           const dlka_lka3d_grads *glka, void *workspace, size_t workspace_bytes, int B, int C, int D, int H, int W, int dtype, void *stream)
{
    return dlka_tblock3d_backward_v(p, lka, drop_mask, training, bn_stats, grad_y, saved, saved_bytes, grad_x, gr, glka, workspace, workspace_bytes, B, C, D, H, W, dtype,
                                    DLKA_LKA3D_SYNAPSE, stream);
}

int dlka_tblock3d_backward_v(const dlka_tblock3d_params *p, const dlka_lka3d_params *lka, const void *drop_mask, int training, const void *bn_stats,
                             const void *grad_y, const void *saved, size_t saved_bytes, void *grad_x, const dlka_tblock3d_grads *gr,
                             const dlka_lka3d_grads *glka, void *workspace, size_t workspace_bytes, int B, int C, int D, int H, int W, int dtype, int variant,
                             void *stream)
{
    return dlka_tblock3d_backward_phase_v(p, lka, drop_mask, training, bn_stats, grad_y, saved, saved_bytes, grad_x, gr, glka, workspace, workspace_bytes, B, C, D, H, W,
                                          dtype, variant, 0, stream);
}

// phase 0: the whole backward pass (dlka_tblock3d_backward_v).  phase 1: the DATA-gradient chain only — grad_x and every gradient a data kernel produces on its
// way (LayerNorm / BatchNorm affine parameters, gamma, pos_embed) — leaving in `workspace` what phase 2 reads; phase 2: the wrapper's three conv weight gradients, the
// attention's weight gradients and the fold of their partial sums, reading `workspace` as phase 1 left it.  A caller that issues phase 2 on another stream (behind an
// event recorded after phase 1, same `workspace`, which must stay untouched until phase 2 has run) lets a block's weight gradients overlap the NEXT block's data
// chain: what the block-stack engine does for the bare attention (dlka_lka3d_attention_tokens_backward_phase_v), here for the block the trainers call.
int dlka_tblock3d_backward_phase_v(const dlka_tblock3d_params *p, const dlka_lka3d_params *lka, const void *drop_mask, int training, const void *bn_stats,
                                   const void *grad_y, const void *saved, size_t saved_bytes, void *grad_x, const dlka_tblock3d_grads *gr,
                                   const dlka_lka3d_grads *glka, void *workspace, size_t workspace_bytes, int B, int C, int D, int H, int W, int dtype, int variant,
                                   int phase, void *stream)
{
    if (phase < 0 || phase > 2) return DLKA_ERR_SHAPE;
    if (!p || !lka || !bn_stats || !grad_y || !saved || !grad_x || !gr || !glka || !workspace) return DLKA_ERR_NULL;
    if (!gr->norm_w || !gr->norm_b || !gr->gamma || !gr->conv51_conv1_w || !gr->conv51_conv2_w || !gr->conv51_norm1_w || !gr->conv51_norm1_b ||
        !gr->conv51_norm2_w || !gr->conv51_norm2_b || !gr->conv8_w || !gr->conv8_b)
        return DLKA_ERR_NULL;
    if ((p->pos_embed != nullptr) != (gr->pos_embed != nullptr)) return DLKA_ERR_NULL;
    if (!dlka_tblock3d_supported_v(B, C, D, H, W, dtype, variant)) return DLKA_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    TBlockGeoms G(B, C, D, H, W);
    Carver sv((void *)saved, saved_bytes), cv(workspace, workspace_bytes);
    TBlockSaved S;
    if (!carve_tblock_saved(sv, G, B, C, D, H, W, S, variant, dtype)) return DLKA_ERR_WORKSPACE;
    const size_t lka_ws_bytes = dlka_lka3d_tokens_workspace_bytes_v(B, C, D, H, W, dtype, variant);
    void *lka_ws = cv.take(lka_ws_bytes);
    (void)cv.take(G.wp_floats() * 4);   // (layout kept: the forward call carves the same region)
    float *part1 = (float *)cv.take(G.part_floats() * 4), *part2 = (float *)cv.take(G.part_floats() * 4);
    float *part8 = (float *)cv.take(cl_wgrad_part_floats(G.pw.M, 1, G.pw.Cout, G.pw.Cin) * 4);
    float *b0 = (float *)cv.take(G.E * 4), *b1 = (float *)cv.take(G.E * 4), *b2 = (float *)cv.take(G.E * 4), *b3 = (float *)cv.take(G.E * 4);
    float *b4 = (float *)cv.take(G.E * 4), *b5 = (float *)cv.take(G.E * 4);
    float *sums = (float *)cv.take(4096);
    const size_t lka_part_bytes = dlka_lka3d_tokens_partials_bytes_v(B, C, D, H, W, dtype, variant);
    void *lka_part = cv.take(lka_part_bytes);   // the attention's partial sums when the pass is split (phase 1 / 2): outside its own workspace
    if (!cv.ok()) return DLKA_ERR_WORKSPACE;
    const long M = (long)G.M, N = G.c3.N;
    const float *st1 = (const float *)bn_stats, *st2 = st1 + 3 * C;
    const float *gy = (const float *)grad_y, *mask = (const float *)drop_mask;
    const float slope = 0.01f;
    const int lo = dtype == DLKA_BF16 ? 1 : 0;   // g_e / g_xn are bf16 storage (the D-LKA attention ran DLKA_BF16)
    // (g_c1 lives in b0 — g_rd is dead by then —, not beside g_c2 in b1: phase 2 reads BOTH g_c2 and g_c1 after the data chain has finished)
    float *g_rd = b0, *g_c2 = b1, *g_skip = b2, *g_attn = b3, *g_a1 = b4, *g_c1 = b0, *g_e = b4, *g_xn = b5;
    if (phase == 2) {   // the weight gradients alone, from what phase 1 left in `workspace`
        FinalizeBatch fb2;
        memset(&fb2, 0, sizeof(fb2));
        DLKA_TRY(dense_backward_weight(G.pw, S.rd, gy, 0, (float *)gr->conv8_w, (float *)gr->conv8_b, part8, st, &fb2.j[fb2.njobs++]));
        DLKA_TRY(dense_backward_weight(G.c3, S.a1, g_c2, 0, (float *)gr->conv51_conv2_w, nullptr, part2, st, &fb2.j[fb2.njobs++]));
        DLKA_TRY(dense_backward_weight(G.c3, S.attn, g_c1, 0, (float *)gr->conv51_conv1_w, nullptr, part1, st, &fb2.j[fb2.njobs++]));
        FinalizeJob jobs[FIN_JOBS_PER_BLOCK];
        int nj = 0;
        DLKA_TRY(tokens_backward_impl(S.xn, lka, g_e, S.lka, S.lka_bytes, g_xn, glka, lka_ws, lka_ws_bytes, B, C, D, H, W, dtype, variant, stream, lka_part,
                                      lka_part_bytes, jobs, &nj, 2));
        if (fb2.njobs + nj > (int)(sizeof(fb2.j) / sizeof(fb2.j[0]))) return DLKA_ERR_UNSUPPORTED;
        for (int k = 0; k < nj; ++k) fb2.j[fb2.njobs++] = jobs[k];
        return launch_cl_wgrad_finalize(fb2, st);
    }
    // everything this direction accumulates into with atomics, zero-filled by ONE launch; the weight re-layouts were done by the forward call;
    // the three weight-gradient folds are ONE launch
    {
        ZeroBatch zb;
        memset(&zb, 0, sizeof(zb));
        zb.add(sums, 1024);
        zb.add((float *)gr->gamma, C);
        zb.add((float *)gr->norm_w, C);
        zb.add((float *)gr->norm_b, C);
        if (gr->pos_embed) zb.add((float *)gr->pos_embed, (size_t)N * C);
        if (dense_backward_data_splits(G.c3, 0) > 1) zb.add(g_a1, G.E);
        if (dense_backward_data_splits(G.c3, 3) > 1) zb.add(g_attn, G.E);
        DLKA_TRY(launch_zero_batch(zb, st));
    }
    FinalizeBatch fb;
    memset(&fb, 0, sizeof(fb));
    // conv8[1]:  y = W8 rd + b8 + attn
    if (phase == 0) DLKA_TRY(dense_backward_weight(G.pw, S.rd, gy, 0, (float *)gr->conv8_w, (float *)gr->conv8_b, part8, st, &fb.j[fb.njobs++]));
    DLKA_TRY(dense_backward_data(G.pw, gy, 0, nullptr, g_rd, S.w8_b, 0, nullptr, st, nullptr, nullptr, true));
    // Dropout3d + LeakyReLU + (BN2(c2) + attn):  g_c2, and everything that flows into attn so far:  g_skip = gy + g_pre
    DLKA_TRY(launch_cl_bn_bwd(g_rd, mask, S.c2, S.rd, (const float *)p->conv51_norm2_w, st2, sums, g_c2, g_skip, gy, (float *)gr->conv51_norm2_w,
                              (float *)gr->conv51_norm2_b, M, N, C, slope, training, st, true));
    // conv2
    if (phase == 0) DLKA_TRY(dense_backward_weight(G.c3, S.a1, g_c2, 0, (float *)gr->conv51_conv2_w, nullptr, part2, st, &fb.j[fb.njobs++]));
    DLKA_TRY(dense_backward_data(G.c3, g_c2, 0, nullptr, g_a1, S.w2_b, 0, nullptr, st, nullptr, nullptr, true));
    // LeakyReLU + BN1
    DLKA_TRY(launch_cl_bn_bwd(g_a1, nullptr, S.c1, S.a1, (const float *)p->conv51_norm1_w, st1, sums + 512, g_c1, nullptr, nullptr, (float *)gr->conv51_norm1_w,
                              (float *)gr->conv51_norm1_b, M, N, C, slope, training, st, true));
    // conv1:  g_attn = W1^T g_c1 + g_skip
    if (phase == 0) DLKA_TRY(dense_backward_weight(G.c3, S.attn, g_c1, 0, (float *)gr->conv51_conv1_w, nullptr, part1, st, &fb.j[fb.njobs++]));
    // (the fold of the three conv weight gradients above rides in the D-LKA block's finalize launch below: one dependent launch less per block)
    DLKA_TRY(dense_backward_data(G.c3, g_c1, 0, nullptr, g_attn, S.w1_b, 3, g_skip, st, nullptr, nullptr, true));
    // attn = xt + gamma * e
    DLKA_TRY(launch_cl_scale_residual_bwd(g_attn, S.e, (const float *)p->gamma, g_e, (float *)gr->gamma, M, C, st, true, lo));
    // epa_block
    if (phase == 1) {
        FinalizeJob jobs[FIN_JOBS_PER_BLOCK];
        int nj = 0;
        DLKA_TRY(tokens_backward_impl(S.xn, lka, g_e, S.lka, S.lka_bytes, g_xn, glka, lka_ws, lka_ws_bytes, B, C, D, H, W, dtype, variant, stream, lka_part, lka_part_bytes,
                                      jobs, &nj, 1));
    } else
        DLKA_TRY(tokens_backward_impl(S.xn, lka, g_e, S.lka, S.lka_bytes, g_xn, glka, lka_ws, lka_ws_bytes, B, C, D, H, W, dtype, variant, stream, nullptr, 0, nullptr,
                                      nullptr, 0, &fb));
    // LayerNorm (+ the residual branch g_attn), pos_embed
    DLKA_TRY(launch_cl_layernorm_bwd(g_xn, g_attn, S.xt, S.lnstats, (const float *)p->norm_w, (float *)grad_x, (float *)gr->norm_w, (float *)gr->norm_b,
                                     (float *)gr->pos_embed, B, (int)N, C, st, true, lo));
    return DLKA_OK;
}

}  // extern "C"
