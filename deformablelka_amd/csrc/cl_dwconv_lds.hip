// Channels-last depthwise 3-D convolution through an LDS brick (round 4) — an OPT-IN alternative (DLKA_DW_LDS=1) to cl_dwconv_rowsN_kernel for the
// large-volume stages of the D-LKA block (5^3 pad 2 and 7^3 dilation 3 pad 9: 3D/d_lka_former/network_architecture/synapse/transformerblock.py:637-638;
// cuDNN in the reference).  MEASURED NO FASTER than the register-row kernel (profiles/r05_notes.md: 40 - 45 us against 42 at (32 channels, 32^3, 7^3
// dilation 3)), hence not selected by default; kept, tested, as the record of what bounds this conv on the MI355X:
//
// cl_dwconv_rowsN_kernel reads every input row segment through the vector L1 — one 256-byte wave load per segment element, 1.7 GB per launch at
// that shape, 42 us at the L1's 64 bytes / clock / CU.  Here a workgroup stages the input brick of its outputs ONCE into LDS (4 channels of every
// cell) and every tap reads LDS.  A DILATED conv decomposes into DIL^3 independent dense convs: outputs with coordinates = (rd, rh, rw) mod DIL only
// read inputs of the same residue class, so the brick lives in "residue space" (cell q <-> voxel r + DIL q), where the 7^3 dilation-3 conv is a dense
// 7^3 conv with halo 3 on a (D / 3)^3 sub-volume — 11^3 outputs + halo = 17^3 cells at 32^3: with 4 channels per workgroup the WHOLE class fits in
// 78.6 KB, two workgroups per CU.  Zero padding is stored (cells outside the volume = 0), the tap loops are branch-free: per input row, 77 FMAs per
// lane against 9 LDS read instructions and one address add (ISA count).
//
// What the three versions measured (r5l, r5m, r5n; us per launch at the shape above, the register-row kernel 42):
//   1. lane = one channel, brick staged straight from the channels-last tensor (16 bytes of every 128-byte line; the other 112 fetched and dropped),
//      plain v_fma_f32: 40.0 with single rows per work-item (4 waves per SIMD), 68.8 with row pairs (66 work-items per channel = 5 waves per workgroup,
//      three on two of the SIMDs);
//   2. (this file) brick staged from a CLASS-BLOCKED fp32 copy of the input, blk[b][channel quad][residue class][qd][qh][qw][4], in which a brick row is
//      one contiguous run — written by cl_dw_block_kernel or, between two consecutive depthwise convs, by the producing conv's epilogue
//      (DwArgs::out_blk) — and lane = channel PAIR on v_pk_fma_f32: 42.7;  3. the same with two v_fma_f32 per pair (-DDLKA_NO_PK): 45.4.
// I.e. the time is the FMA issue itself: a wave64 v_fma_f32 occupies its 16-lane SIMD for 4 cycles and v_pk_fma_f32 for 8 (no gain from packing:
// the same A/B over the whole library, built with and without the SLP vectoriser's packed fp32, moved no kernel by more than 8 % either way), so the
// chip does 64 fp32 FMA lanes per clock and CU: 720 M FMAs (+ 16 % of lane and row padding here) = 21 - 25 us at the 2.1 - 2.4 GHz the clocks settle at,
// before staging, the per-plane weight loads and the tail of a 432-workgroup launch on 512 slots.  A depthwise conv has no contraction to put on the
// matrix cores (a banded Toeplitz product would spend 7x the useful FLOPs), so ~25 us is this conv's floor and both kernels sit at 1.7x of it.
//
// Work-item = (channel pair, W-run of TW outputs, TH consecutive h rows, one d) in residue space; rows are an ODD number of cells apart (bank spread).
// The KW x KW tap weights of a d-plane sit in registers, the first two tap rows of the next plane are requested a plane ahead; row r + 1 is read from
// LDS while row r is multiplied.  Same kernel = forward and data gradient (flipped taps); epilogues as cl_dwconv_rowsN_kernel.  Compiled with
// -fno-slp-vectorize (Makefile): the packing is explicit, and the SLP vectoriser's own pairing of neighbouring OUTPUTS needs shifted copies of the
// row segment (168 registers plus scratch).
#include <stdlib.h>

#include <atomic>

#include "cl_args.h"
#include "dlka_kernels.h"

namespace dlka {

typedef f32x2 dwv2;

constexpr int DWL_CG = 4;   // channels per workgroup = one 16-byte piece of the blocked layout

struct DwLdsGeom {
    int bd, bh, bw;      // outputs per brick along d, h, w (residue space)
    int nbd, nbh, nbw;   // bricks per residue class (sized for the largest class)
    int SD, SH, SW;      // cells of the LDS brick: bd + KW - 1, rows_h * TH + KW - 1, runs_w * TW + KW - 1 rounded up to odd
    int rows_h, runs_w;  // work-items per (channel pair, d): ceil(bh / TH) row groups x ceil(bw / TW) runs
    int ncg;             // channel quads (C / 4)
    int Dr, Hr, Wr;      // cells per residue class in the blocked source: ceil(D / DIL), ...
    int nitems, xcd_nx;  // work items = B * DIL^3 * nbd * nbh * nbw * ncg; xcd_nx > 0: blockIdx.x is mapped through xcd_item()
    int oDr, oHr, oWr;   // DwArgs::out_blk: cells per class of the blocked OUTPUT copy (dilation out_blk_dil)
};

__host__ __device__ __forceinline__ size_t dw_blk_floats(int B, int C, int D, int H, int W, int dil)
{
    return (size_t)B * C * dil * dil * dil * cdiv(D, dil) * cdiv(H, dil) * cdiv(W, dil);
}

// float index of (b, channel quad cg, voxel (d, h, w)) in the class-blocked layout of dilation dil (Dr, Hr, Wr = cells per class)
__device__ __forceinline__ long dw_blk_index(int b, int cg, int ncg, int d, int h, int w, int dil, int Dr, int Hr, int Wr)
{
    const int cls = ((d % dil) * dil + h % dil) * dil + w % dil;
    return ((((((long)b * ncg + cg) * (dil * dil * dil) + cls) * Dr + d / dil) * Hr + h / dil) * Wr + w / dil) * DWL_CG;
}

// in [B][D][H][W][C] (T) -> blk (fp32, class-blocked); one 16-byte piece per work-item, reads coalesced
template <typename T>
__global__ __launch_bounds__(256) void cl_dw_block_kernel(const float *__restrict__ in_, float *__restrict__ blk, int B, int D, int H, int W, int C, int dil)
{
    const T *in = reinterpret_cast<const T *>(in_);
    const int ncg = C / DWL_CG, Dr = cdiv(D, dil), Hr = cdiv(H, dil), Wr = cdiv(W, dil);
    const long n = (long)B * D * H * W * ncg;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int cg = (int)(e % ncg);
        long v = e / ncg;
        const int w = (int)(v % W); v /= W;
        const int h = (int)(v % H); v /= H;
        const int d = (int)(v % D), b = (int)(v / D);
        *reinterpret_cast<f32x4 *>(blk + dw_blk_index(b, cg, ncg, d, h, w, dil, Dr, Hr, Wr)) = act_load4(in, e * DWL_CG);
    }
}

template <typename T, int KW, int DIL, int TW, int TH>
__global__ __launch_bounds__(512, 2) void cl_dwconv_lds_kernel(DwArgs p, DwLdsGeom g)
{
    constexpr int R = (KW - 1) / 2, SEG = TW + KW - 1, NR = KW + TH - 1, KH = KW, CG = DWL_CG;
    DLKA_DYN_SMEM(float, Ws);   // [SD][SH][SW][CG]
    const T *gxp = reinterpret_cast<const T *>(p.gelu_x), *gap = reinterpret_cast<const T *>(p.gelu_add);
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

    // ---- stage the brick from the blocked copy: cell (zd, zh, zw) <-> class cell o + z - R; zeros outside the volume ----
    const int cells = g.SD * g.SH * g.SW;
    const int cd = cdiv(p.D - rd, DIL), ch = cdiv(p.H - rh, DIL), cw = cdiv(p.W - rw, DIL);   // cells of THIS class
    const float *src = p.blk + ((((long)b * g.ncg + cg) * (DIL * DIL * DIL) + cls) * g.Dr) * g.Hr * g.Wr * CG;
    for (int e = tid; e < cells; e += blockDim.x) {
        const int zw = e % g.SW, zh = (e / g.SW) % g.SH, zd = e / (g.SW * g.SH);
        const int qd = od + zd - R, qh = oh + zh - R, qw = ow + zw - R;
        const bool ok = (unsigned)qd < (unsigned)cd && (unsigned)qh < (unsigned)ch && (unsigned)qw < (unsigned)cw;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4 *>(src + (((long)qd * g.Hr + qh) * g.Wr + qw) * CG);
        *reinterpret_cast<f32x4 *>(Ws + (long)e * CG) = v;
    }
    __syncthreads();

    // ---- this work-item's outputs: channels cch, cch + 1 ----
    const int pr = tid % 2;
    int q = tid / 2;
    const int run = q % g.runs_w; q /= g.runs_w;
    const int hq = q % g.rows_h, dq = q / g.rows_h;
    if (dq >= g.bd) return;   // (no barrier below)
    const int cch = cg * CG + 2 * pr;
    dwv2 acc[TH][TW];
    dwv2 bv = {0.f, 0.f};
    if (p.bias) bv = dwv2{p.bias[cch], p.bias[cch + 1]};
#pragma unroll
    for (int o = 0; o < TH; ++o)
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[o][t] = bv;

    const BufRsrc rwt = make_rsrc(p.wp, (size_t)KW * KH * KW * p.C * 4);
    const unsigned cv = (unsigned)cch * 4u, cbw = (unsigned)p.C * 4u;
    // tap weights of plane i in registers; the first two tap rows of the NEXT plane are requested a plane ahead (a full second buffer costs 2 KW KH registers)
    dwv2 wv[KH][KW], wp[2][KW];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < KW; ++k) wp[j][k] = buf_load_f32x2_s(rwt, cv, (unsigned)(j * KW + k) * cbw);
    const float *base = Ws + ((long)(dq * g.SH + hq * TH) * g.SW + run * TW) * CG + 2 * pr;
    const int rowst = g.SW * CG, planest = g.SH * g.SW * CG;
#pragma unroll 1
    for (int i = 0; i < KW; ++i) {
#pragma unroll
        for (int j = 0; j < KH; ++j)
#pragma unroll
            for (int k = 0; k < KW; ++k) wv[j][k] = j < 2 ? wp[j][k] : buf_load_f32x2_s(rwt, cv, (unsigned)((i * KH + j) * KW + k) * cbw);
        if (i + 1 < KW) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < KW; ++k) wp[j][k] = buf_load_f32x2_s(rwt, cv, (unsigned)(((i + 1) * KH + j) * KW + k) * cbw);
        }
        const float *pl = base + (long)i * planest;
        dwv2 seg[2][SEG];   // (the scheduling fences keep the compiler from hoisting ALL rows' reads)
#pragma unroll
        for (int e = 0; e < SEG; ++e) seg[0][e] = *reinterpret_cast<const dwv2 *>(pl + e * CG);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r + 1 < NR) {
#pragma unroll
                for (int e = 0; e < SEG; ++e) seg[(r + 1) & 1][e] = *reinterpret_cast<const dwv2 *>(pl + (r + 1) * rowst + e * CG);
            }
#pragma unroll
            for (int o = 0; o < TH; ++o) {
                if (r - o < 0 || r - o >= KH) continue;   // compile time
#pragma unroll
                for (int k = 0; k < KW; ++k)
#pragma unroll
                    for (int t = 0; t < TW; ++t) acc[o][t] = pk_fma(wv[r - o][k], seg[r & 1][t + k], acc[o][t]);
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
            dwv2 r_ = acc[o][t];
            if (p.gelu_x) {   // (uniform: see cl_dwconv_kernel)
                r_[0] = (r_[0] + act_load1(gap, o_)) * dgelu_f(act_load1(gxp, o_));
                r_[1] = (r_[1] + act_load1(gap, o_ + 1)) * dgelu_f(act_load1(gxp, o_ + 1));
            }
            act_store2(outp, o_, r_[0], r_[1]);
            if (sizeof(T) == 4 && p.out_lo && !p.gelu_x) act_store2(reinterpret_cast<bf16_t *>(p.out_lo), o_, r_[0], r_[1]);
            if (p.out_blk)   // the following depthwise conv's blocked source (fp32 whatever T is)
                *reinterpret_cast<dwv2 *>(p.out_blk + dw_blk_index(b, cg, g.ncg, vd, vh, vw, p.out_blk_dil, g.oDr, g.oHr, g.oWr) + 2 * pr) = r_;
        }
    }
}

// Brick geometry for one launch; false = shape not able (the caller keeps cl_dwconv_rowsN_kernel).
template <int KW, int DIL, int TW, int TH>
static bool dw_lds_plan(const DwArgs &a, int bd_max, int bh_max, int bw_max, DwLdsGeom &g, size_t &lds, int &threads)
{
    if (a.C % DWL_CG != 0) return false;
    g.Dr = cdiv(a.D, DIL); g.Hr = cdiv(a.H, DIL); g.Wr = cdiv(a.W, DIL);
    g.bd = g.Dr < bd_max ? g.Dr : bd_max; g.bh = g.Hr < bh_max ? g.Hr : bh_max; g.bw = g.Wr < bw_max ? g.Wr : bw_max;
    g.nbd = cdiv(g.Dr, g.bd); g.nbh = cdiv(g.Hr, g.bh); g.nbw = cdiv(g.Wr, g.bw);
    g.rows_h = cdiv(g.bh, TH); g.runs_w = cdiv(g.bw, TW);
    g.SD = g.bd + KW - 1; g.SH = g.rows_h * TH + KW - 1; g.SW = (g.runs_w * TW + KW - 1) | 1;
    g.ncg = a.C / DWL_CG;
    const long items = (long)a.B * DIL * DIL * DIL * g.nbd * g.nbh * g.nbw * g.ncg;
    if (items > (1l << 30)) return false;
    g.nitems = (int)items; g.xcd_nx = 0;
    g.oDr = g.oHr = g.oWr = 0;
    if (a.out_blk) { g.oDr = cdiv(a.D, a.out_blk_dil); g.oHr = cdiv(a.H, a.out_blk_dil); g.oWr = cdiv(a.W, a.out_blk_dil); }
    lds = (size_t)g.SD * g.SH * g.SW * DWL_CG * 4;
    threads = round_up(2 * g.runs_w * g.rows_h * g.bd, 64);
    return lds <= 156 * 1024 && threads <= 512 && threads >= 64;
}

static std::atomic<long> g_dw_lds_launches{0};   // dlka_dwconv_lds_launch_count (include/dlka.h): diagnostics

template <typename T, int KW, int DIL, int TW, int TH>
static int dw_lds_launch(const DwArgs &a, DwLdsGeom g, size_t lds, int threads, hipStream_t st)
{
    auto k = cl_dwconv_lds_kernel<T, KW, DIL, TW, TH>;
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
    if (!a.in_blocked) {   // the class-blocked copy of the input (DwArgs::blk); a preceding LDS-brick conv may have written it already
        const long pieces = (long)a.B * a.D * a.H * a.W * (a.C / DWL_CG);
        long blocks = cdivl(pieces, 256);
        if (blocks > 8192) blocks = 8192;
        if (a.act_bf16) { auto kb = cl_dw_block_kernel<bf16_t>; DLKA_LAUNCH(kb, dim3((unsigned)blocks), dim3(256), 0, st, a.in, a.blk, a.B, a.D, a.H, a.W, a.C, DIL); }
        else { auto kb = cl_dw_block_kernel<float>; DLKA_LAUNCH(kb, dim3((unsigned)blocks), dim3(256), 0, st, a.in, a.blk, a.B, a.D, a.H, a.W, a.C, DIL); }
        DLKA_CHECK_LAUNCH();
    }
    DwArgs ax = a;
    dim3 grid((unsigned)g.nitems);
    if (xcd_swizzle_enabled() && g.nitems >= xcd_min_blocks()) { g.xcd_nx = g.nitems; grid.x = xcd_grid(g.nitems); }
    DLKA_LAUNCH(k, grid, dim3(threads), lds, st, ax, g);
    DLKA_CHECK_LAUNCH();
    g_dw_lds_launches.fetch_add(1, std::memory_order_relaxed);
    return DLKA_OK;
}

// DLKA_DW_LDS: unset / "0" = never (the default: see the header), "1" = per-shape selection below (the A/B runs of profiles/r05_notes.md), "2" = wherever
// the geometry fits (emulator tests reach the kernel at small shapes with it).  Read per launch, like DLKA_DDW2D_GX.
static int dw_lds_mode()
{
    const char *e = getenv("DLKA_DW_LDS");
    return e ? atoi(e) : 0;
}

// variant of the launch: 0 = not this kernel; 1 = 7^3 dil 3, classes of up to 11^3 (rows of 11); 2 = 7^3 dil 3, classes of up to 6^3; 3 = 5^3
template <typename T>
static int dw_lds_select(const DwArgs &a, int kw, int dil_w, DwLdsGeom &g, size_t &lds, int &threads)
{
    const int mode = dw_lds_mode();
    if (mode == 0 || !a.blk) return 0;
    const bool cubic = a.kd == kw && a.kh == kw && a.dd == dil_w && a.dh == dil_w;
    const int pad = dil_w * (kw - 1) / 2;
    if (!cubic || a.pd != pad || a.ph != pad || a.pw != pad) return 0;
    if (dw_blk_floats(a.B, a.C, a.D, a.H, a.W, dil_w) > a.blk_floats) return 0;
    const long outs = (long)a.B * a.D * a.H * a.W * a.C;
    if (kw == 7 && dil_w == 3) {
        // whole residue classes of up to 11^3 outputs: 17^3 cells = 78.6 KB, two workgroups of four waves per CU (stage 0: 432 workgroups)
        if (cdiv(a.W, 3) > 6 && dw_lds_plan<7, 3, 11, 1>(a, 11, 11, 11, g, lds, threads) && (mode == 2 || (outs >= (1l << 21) && g.nitems >= 256))) return 1;
        // <= 6^3 outputs per class (16^3 volumes): 12 x 12 x 13 cells = 30 KB
        if (cdiv(a.W, 3) <= 6 && dw_lds_plan<7, 3, 6, 1>(a, 6, 6, 6, g, lds, threads) && (mode == 2 || (outs >= (1l << 19) && g.nitems >= 256))) return 2;
        return 0;
    }
    if (kw == 5 && dil_w == 1) {
        // bricks of 8 x 8 x 16 outputs: 12 x 12 x 21 cells = 48 KB, three workgroups of four waves per CU
        if (dw_lds_plan<5, 1, 8, 1>(a, a.D >= 32 ? 8 : 4, 8, 16, g, lds, threads) && (mode == 2 || (outs >= (1l << 19) && g.nitems >= 256))) return 3;
        return 0;
    }
    return 0;
}

// Returns DLKA_ERR_UNSUPPORTED when this launch should stay on cl_dwconv_rowsN_kernel.
template <typename T>
static int launch_cl_dwconv_lds_t(const DwArgs &a, int kw, int dil_w, hipStream_t st)
{
    DwLdsGeom g;
    size_t lds = 0;
    int threads = 0;
    switch (dw_lds_select<T>(a, kw, dil_w, g, lds, threads)) {
    case 1: return dw_lds_launch<T, 7, 3, 11, 1>(a, g, lds, threads, st);
    case 2: return dw_lds_launch<T, 7, 3, 6, 1>(a, g, lds, threads, st);
    case 3: return dw_lds_launch<T, 5, 1, 8, 1>(a, g, lds, threads, st);
    default: return DLKA_ERR_UNSUPPORTED;
    }
}

int launch_cl_dwconv_lds(const DwArgs &a, int kw, int dil_w, hipStream_t st)
{
    return a.act_bf16 ? launch_cl_dwconv_lds_t<bf16_t>(a, kw, dil_w, st) : launch_cl_dwconv_lds_t<float>(a, kw, dil_w, st);
}

// would launch_cl_dwconv take the LDS-brick kernel for this conv?  (the token path asks before it lets the PREVIOUS conv write the blocked copy)
bool cl_dwconv_lds_selected(const DwArgs &a, int kw, int dil_w)
{
    DwLdsGeom g;
    size_t lds = 0;
    int threads = 0;
    return dw_lds_select<float>(a, kw, dil_w, g, lds, threads) != 0;
}

size_t cl_dwconv_blk_floats(int B, int C, int D, int H, int W, int dil) { return dw_blk_floats(B, C, D, H, W, dil); }
int cl_dwconv_lds_mode() { return dw_lds_mode(); }

}  // namespace dlka

extern "C" long dlka_dwconv_lds_launch_count(void) { return dlka::g_dw_lds_launches.load(std::memory_order_relaxed); }
