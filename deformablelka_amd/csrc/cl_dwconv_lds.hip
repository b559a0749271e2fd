// Channels-last depthwise 3-D convolution through an LDS brick (round 4) — the large-volume stages of the D-LKA block
// (5^3 pad 2 and 7^3 dilation 3 pad 9: 3D/d_lka_former/network_architecture/synapse/transformerblock.py:637-638; cuDNN in the reference).
//
// Why: cl_dwconv_rowsN_kernel (cl_dwconv.hip) reads every input row segment through the vector L1 — one 256-byte wave load per segment element,
// 1.7 GB per launch at (32 channels, 32^3, 7^3 dilation 3) — and that launch's 42 us ARE those bytes at the L1's 64 bytes / clock / CU; its 720 M
// FMAs would take 9 us.  Register tiles cannot buy the missing reuse (more outputs per work-item = one wave per SIMD; measured slower, r01o).
//
// Here a workgroup stages the input brick of its outputs ONCE (global -> LDS, CG channels of every cell: 16- or 32-byte pieces) and every tap reads
// LDS (128 bytes / clock / CU, and only the brick travels through L1 / L2).  A DILATED conv decomposes into DIL^3 independent dense convs: outputs
// with coordinates = (rd, rh, rw) mod DIL only read inputs of the same residue class, so the brick lives in "residue space" (cell q <-> voxel
// r + DIL q), where the 7^3 dilation-3 conv is a dense 7^3 conv with halo 3 on an (D / 3)^3 sub-volume — 11^3 outputs + halo = 17^3 cells at 32^3:
// with 4 channels per workgroup the WHOLE class fits in 78.6 KB, two workgroups per CU.  Zero padding is stored (cells outside the volume = 0), the
// tap loops are branch-free.
//
// Work-item = (channel c of the group, W-run of TW outputs, TH consecutive h rows, one d) in residue space: per input row NR = KW + TH - 1 rows x
// SEG = TW + KW - 1 LDS reads feed TH * KW * TW FMAs (7.9 FMAs per 4-byte read at 7^3, TW = 11, TH = 2); lanes run c-fastest, rows are an ODD number
// of cells apart so that the 64 lanes of a read hit 64 different banks.  The KW x KW tap weights of a d-plane sit in registers, the next plane's are
// loaded at the top of the plane; row r + 1 is read from LDS while row r is multiplied.  Same kernel = forward and data gradient (flipped taps);
// epilogues as cl_dwconv_rowsN_kernel.  Compiled with -fno-slp-vectorize (Makefile): left alone, the SLP vectoriser pairs the FMAs of neighbouring
// outputs into v_pk_fma_f32, whose operand pairs need shifted copies of the row segment — 168 registers plus scratch instead of 113, and packed fp32
// is no faster than two plain FMAs on this chip (MI355X_MICROARCH.md: v_fma_f32 issues in 2 cycles per wave).
#include <stdlib.h>

#include <atomic>

#include "cl_args.h"
#include "dlka_kernels.h"

namespace dlka {

struct DwLdsGeom {
    int bd, bh, bw;      // outputs per brick along d, h, w (residue space)
    int nbd, nbh, nbw;   // bricks per residue class (sized for the largest class)
    int SD, SH, SW;      // cells of the LDS brick: bd + KW - 1, rows_h * TH + KW - 1, runs_w * TW + KW - 1 rounded up to odd
    int rows_h, runs_w;  // work-items per (c, d): ceil(bh / TH) row groups x ceil(bw / TW) runs
    int ncg;             // channel groups (C / CG)
    int nitems, xcd_nx;  // work items = B * DIL^3 * nbd * nbh * nbw * ncg; xcd_nx > 0: blockIdx.x is mapped through xcd_item()
};

template <typename T, int KW, int DIL, int CG, int TW, int TH>
__global__ __launch_bounds__(512, 3) void cl_dwconv_lds_kernel(DwArgs p, DwLdsGeom g)
{
    constexpr int R = (KW - 1) / 2, SEG = TW + KW - 1, NR = KW + TH - 1, KH = KW;
    static_assert(CG % 4 == 0, "channel groups are loaded in 4-channel pieces");
    DLKA_DYN_SMEM(float, Ws);   // [SD][SH][SW][CG]
    const T *inp = reinterpret_cast<const T *>(p.in), *gxp = reinterpret_cast<const T *>(p.gelu_x), *gap = reinterpret_cast<const T *>(p.gelu_add);
    T *outp = reinterpret_cast<T *>(p.out);
    const int tid = threadIdx.x;
    int item = g.xcd_nx > 0 ? xcd_item((int)blockIdx.x, g.xcd_nx) : (int)blockIdx.x;
    if (item < 0 || item >= g.nitems) return;   // (uniform per workgroup)
    const int cg = item % g.ncg; item /= g.ncg;
    const int bwi = item % g.nbw; item /= g.nbw;
    const int bhi = item % g.nbh; item /= g.nbh;
    const int bdi = item % g.nbd; item /= g.nbd;
    const int cls = item % (DIL * DIL * DIL), b = item / (DIL * DIL * DIL);
    const int rw = cls % DIL, rh = (cls / DIL) % DIL, rd = cls / (DIL * DIL);
    const int od = bdi * g.bd, oh = bhi * g.bh, ow = bwi * g.bw;   // first output of the brick (residue space)
    if (rd + DIL * od >= p.D || rh + DIL * oh >= p.H || rw + DIL * ow >= p.W) return;   // smaller residue classes have fewer bricks (uniform)

    // ---- stage the brick: cell (zd, zh, zw) <-> voxel r + DIL * (o + z - R); zeros outside the volume ----
    const int cells = g.SD * g.SH * g.SW, c0 = cg * CG;
    for (int e = tid; e < cells; e += blockDim.x) {
        const int zw = e % g.SW, zh = (e / g.SW) % g.SH, zd = e / (g.SW * g.SH);
        const int qd = od + zd - R, qh = oh + zh - R, qw = ow + zw - R;
        const int vd = rd + DIL * qd, vh = rh + DIL * qh, vw = rw + DIL * qw;
        const bool ok = qd >= 0 && qh >= 0 && qw >= 0 && vd < p.D && vh < p.H && vw < p.W;
        const long gi = ok ? ((((long)b * p.D + vd) * p.H + vh) * p.W + vw) * p.C + c0 : 0;
#pragma unroll
        for (int q = 0; q < CG / 4; ++q) {
            f32x4 v = act_load4(inp, gi + 4 * q);
            if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4 *>(Ws + (long)e * CG + 4 * q) = v;
        }
    }
    __syncthreads();

    // ---- this work-item's outputs ----
    const int c = tid % CG;
    int q = tid / CG;
    const int run = q % g.runs_w; q /= g.runs_w;
    const int hq = q % g.rows_h, dq = q / g.rows_h;
    if (dq >= g.bd) return;   // (no barrier below)
    const int cch = c0 + c;
    float acc[TH][TW];
    const float bv = p.bias ? p.bias[cch] : 0.f;
#pragma unroll
    for (int o = 0; o < TH; ++o)
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[o][t] = bv;

    const BufRsrc rwt = make_rsrc(p.wp, (size_t)KW * KH * KW * p.C * 4);
    const unsigned cv = (unsigned)cch * 4u, cbw = (unsigned)p.C * 4u;
    // tap weights of plane i in registers; the first two tap rows of the NEXT plane are requested a plane ahead (the rest is needed from input row 2 on:
    // a full second buffer does not fit the 168-register budget of three waves per SIMD)
    float wv[KH][KW], wp[2][KW];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < KW; ++k) wp[j][k] = buf_load_f32_s(rwt, cv, (unsigned)(j * KW + k) * cbw);
    const float *base = Ws + ((long)(dq * g.SH + hq * TH) * g.SW + run * TW) * CG + c;
    const int rowst = g.SW * CG, planest = g.SH * g.SW * CG;
#pragma unroll 1
    for (int i = 0; i < KW; ++i) {
#pragma unroll
        for (int j = 0; j < KH; ++j)
#pragma unroll
            for (int k = 0; k < KW; ++k) wv[j][k] = j < 2 ? wp[j][k] : buf_load_f32_s(rwt, cv, (unsigned)((i * KH + j) * KW + k) * cbw);
        if (i + 1 < KW) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < KW; ++k) wp[j][k] = buf_load_f32_s(rwt, cv, (unsigned)(((i + 1) * KH + j) * KW + k) * cbw);
        }
        const float *pl = base + (long)i * planest;
        float seg[2][SEG];   // row r + 1 is read from LDS while row r is multiplied (the scheduling fences keep the compiler from hoisting ALL rows' reads)
#pragma unroll
        for (int e = 0; e < SEG; ++e) seg[0][e] = pl[e * CG];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r + 1 < NR) {
#pragma unroll
                for (int e = 0; e < SEG; ++e) seg[(r + 1) & 1][e] = pl[(r + 1) * rowst + e * CG];
            }
#pragma unroll
            for (int o = 0; o < TH; ++o) {
                if (r - o < 0 || r - o >= KH) continue;   // compile time
#pragma unroll
                for (int k = 0; k < KW; ++k)
#pragma unroll
                    for (int t = 0; t < TW; ++t) acc[o][t] = fmaf(wv[r - o][k], seg[r & 1][t + k], acc[o][t]);
            }
            sched_fence();
        }
    }

    const int vd = rd + DIL * (od + dq);
    if (vd >= p.D) return;
#pragma unroll
    for (int o = 0; o < TH; ++o) {
        const int lh = hq * TH + o, vh = rh + DIL * (oh + lh);
        if (lh >= g.bh || vh >= p.H) continue;
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const int lw = run * TW + t, vw = rw + DIL * (ow + lw);
            if (lw >= g.bw || vw >= p.W) continue;
            const long o_ = ((((long)b * p.D + vd) * p.H + vh) * p.W + vw) * p.C + cch;
            if (p.gelu_x) act_store1(outp, o_, (acc[o][t] + act_load1(gap, o_)) * dgelu_f(act_load1(gxp, o_)));   // (uniform: see cl_dwconv_kernel)
            else {
                act_store1(outp, o_, acc[o][t]);
                if (sizeof(T) == 4 && p.out_lo) act_store1(reinterpret_cast<bf16_t *>(p.out_lo), o_, acc[o][t]);
            }
        }
    }
}

// Brick geometry for one launch; false = shape not worth / not able (the caller keeps cl_dwconv_rowsN_kernel).
template <int KW, int DIL, int CG, int TW, int TH>
static bool dw_lds_plan(const DwArgs &a, int bd_max, int bh_max, int bw_max, DwLdsGeom &g, size_t &lds, int &threads)
{
    if (a.C % CG != 0) return false;
    const int Dr = cdiv(a.D, DIL), Hr = cdiv(a.H, DIL), Wr = cdiv(a.W, DIL);
    g.bd = Dr < bd_max ? Dr : bd_max; g.bh = Hr < bh_max ? Hr : bh_max; g.bw = Wr < bw_max ? Wr : bw_max;
    g.nbd = cdiv(Dr, g.bd); g.nbh = cdiv(Hr, g.bh); g.nbw = cdiv(Wr, g.bw);
    g.rows_h = cdiv(g.bh, TH); g.runs_w = cdiv(g.bw, TW);
    g.SD = g.bd + KW - 1; g.SH = g.rows_h * TH + KW - 1; g.SW = (g.runs_w * TW + KW - 1) | 1;
    g.ncg = a.C / CG;
    const long items = (long)a.B * DIL * DIL * DIL * g.nbd * g.nbh * g.nbw * g.ncg;
    if (items > (1l << 30)) return false;
    g.nitems = (int)items; g.xcd_nx = 0;
    lds = (size_t)g.SD * g.SH * g.SW * CG * 4;
    threads = round_up(CG * g.runs_w * g.rows_h * g.bd, 64);
    return lds <= 156 * 1024 && threads <= 512 && threads >= 64;
}

static std::atomic<long> g_dw_lds_launches{0};   // dlka_dwconv_lds_launch_count (include/dlka.h): diagnostics

template <typename T, int KW, int DIL, int CG, int TW, int TH>
static int dw_lds_launch(const DwArgs &a, DwLdsGeom g, size_t lds, int threads, hipStream_t st)
{
    auto k = cl_dwconv_lds_kernel<T, KW, DIL, CG, TW, TH>;
#if !defined(HIPEMU)
    static std::atomic<uint64_t> attr_done{0};   // dynamic LDS above 64 KB: per function and per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return DLKA_ERR_LAUNCH;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DLKA_ERR_LAUNCH;
        attr_done.fetch_or(bit, std::memory_order_release);
    }
#endif
    DwArgs ax = a;
    dim3 grid((unsigned)g.nitems);
    if (xcd_swizzle_enabled() && g.nitems >= xcd_min_blocks()) { g.xcd_nx = g.nitems; grid.x = xcd_grid(g.nitems); }
    DLKA_LAUNCH(k, grid, dim3(threads), lds, st, ax, g);
    DLKA_CHECK_LAUNCH();
    g_dw_lds_launches.fetch_add(1, std::memory_order_relaxed);
    return DLKA_OK;
}

// DLKA_DW_LDS: unset / "1" = per-shape selection below, "0" = never (the A/B switch of profiles/r05_notes.md), "2" = wherever the geometry fits
// (emulator tests reach the kernel at small shapes with it).  Read per launch, like DLKA_DDW2D_GX.
static int dw_lds_mode()
{
    const char *e = getenv("DLKA_DW_LDS");
    return e ? atoi(e) : 1;
}

// Returns DLKA_ERR_UNSUPPORTED when this launch should stay on cl_dwconv_rowsN_kernel.
template <typename T>
static int launch_cl_dwconv_lds_t(const DwArgs &a, int kw, int dil_w, hipStream_t st)
{
    const int mode = dw_lds_mode();
    if (mode == 0) return DLKA_ERR_UNSUPPORTED;
    const bool cubic = a.kd == kw && a.kh == kw && a.dd == dil_w && a.dh == dil_w;
    const int pad = dil_w * (kw - 1) / 2;
    if (!cubic || a.pd != pad || a.ph != pad || a.pw != pad) return DLKA_ERR_UNSUPPORTED;
    DwLdsGeom g;
    size_t lds = 0;
    int threads = 0;
    const long outs = (long)a.B * a.D * a.H * a.W * a.C;
    if (kw == 7 && dil_w == 3) {
        // whole residue classes of up to 11^3 outputs, 4 channels: 17^3 cells = 78.6 KB, two workgroups per CU (stage 0: 432 workgroups of 5 waves)
        // Row pairs (TH = 2) halve the LDS reads per FMA but give 66 work-items per channel = 5 waves per workgroup, 10 per CU: three on two of the SIMDs;
        // single rows (TH = 1) give 121 = 8 waves, 16 per CU, four per SIMD.  DLKA_DW_LDS_TH=2 selects the pairs (A/B, profiles/r05_notes.md).
        const char *th_env = getenv("DLKA_DW_LDS_TH");
        const bool pairs = th_env && atoi(th_env) == 2;
        if (cdiv(a.W, 3) > 6 && !pairs && dw_lds_plan<7, 3, 4, 11, 1>(a, 11, 11, 11, g, lds, threads) && (mode == 2 || (outs >= (1l << 21) && g.nitems >= 256)))
            return dw_lds_launch<T, 7, 3, 4, 11, 1>(a, g, lds, threads, st);
        if (cdiv(a.W, 3) > 6 && dw_lds_plan<7, 3, 4, 11, 2>(a, 11, 12, 11, g, lds, threads) && (mode == 2 || (outs >= (1l << 21) && g.nitems >= 256)))
            return dw_lds_launch<T, 7, 3, 4, 11, 2>(a, g, lds, threads, st);
        // <= 6^3 outputs per class (16^3 volumes), 8 channels: 12 x 12 x 13 cells = 60 KB
        if (cdiv(a.W, 3) <= 6 && dw_lds_plan<7, 3, 8, 6, 2>(a, 6, 6, 6, g, lds, threads) && (mode == 2 || (outs >= (1l << 19) && g.nitems >= 256)))
            return dw_lds_launch<T, 7, 3, 8, 6, 2>(a, g, lds, threads, st);
        return DLKA_ERR_UNSUPPORTED;
    }
    if (kw == 5 && dil_w == 1) {
        // bricks of 8 x 8 x 16 outputs, 4 channels: 12 x 12 x 21 cells = 48 KB, three workgroups per CU
        if (dw_lds_plan<5, 1, 4, 8, 2>(a, a.D >= 32 ? 8 : 4, 8, 16, g, lds, threads) && (mode == 2 || (outs >= (1l << 19) && g.nitems >= 256)))
            return dw_lds_launch<T, 5, 1, 4, 8, 2>(a, g, lds, threads, st);
        return DLKA_ERR_UNSUPPORTED;
    }
    return DLKA_ERR_UNSUPPORTED;
}

int launch_cl_dwconv_lds(const DwArgs &a, int kw, int dil_w, hipStream_t st)
{
    return a.act_bf16 ? launch_cl_dwconv_lds_t<bf16_t>(a, kw, dil_w, st) : launch_cl_dwconv_lds_t<float>(a, kw, dil_w, st);
}

}  // namespace dlka

extern "C" long dlka_dwconv_lds_launch_count(void) { return dlka::g_dw_lds_launches.load(std::memory_order_relaxed); }
