// Data gradient of the offset-predict conv (3^3, C -> 81: gt = Coff^T goff, 3D/dcn/modules/deform_conv.py:78-84 builds that conv) from an LDS BRICK.
//
//     out[m][n] = aux[m][n] + sum_tap sum_k in[b][k][voxel(m) + tap - 1] * Wp[tap][k][n]        in = grad_offset, PLANAR [B][81][N] fp32
//
// Why a kernel of its own.  cl_conv_wave_kernel<2, 0, 1, 2, 2> fetches its A operand — 16 dwords from 16 planes per lane and (tap, chunk) unit — from
// global memory 27 times over, once per tap: in-situ ablations of that kernel (profiles/r06_notes.md) put 45 of its 91 us at 32^3 on that fetch
// alone (split arithmetic 9, the matrix cores 6, the weight records 7), the planar layout being the better of two bad choices (a channels-last copy
// needs a quarter of the load instructions and four times the cache lines, profiles/r04_notes.md).  Here a workgroup owns a TD x TH x W tile of one
// volume (8 waves: 2 x 4 x 32 voxels at the 32^3 stage) and stages the tile's (TD + 2) x (TH + 2) x (W + 2) halo ONCE per 32-plane chunk in LDS — read
// coalesced along the planes' voxels, split into its two bf16 terms there and then (once per element instead of once per tap), zero padding written as
// zeros — as
//     brick[voxel][ hi: 32 x bf16 | lo: 32 x bf16 | 16 bytes of padding ]          144 bytes per voxel
// so that the A operand of v_mfma_f32_32x32x16_bf16 for ANY tap is one 16-byte LDS read per term at a tap-dependent constant offset: no address
// arithmetic, no bounds code, no conversion in the 27-tap loop (the 144-byte row pitch spreads eight lanes' 16-byte accesses over all banks).
// The weights come as the two-term records cl_conv_wave_kernel reads (prep mode 1 | 8), through a register ring three taps deep, straight from L2.
// Same products and the same fp32 accumulation as the kernel it replaces; the order of the sum over (tap, chunk) differs (chunks outer): rounding only.
// Measured (one box, profiles/r06_notes.md): 91 -> 53 us, step 11.04 -> 10.72 ms.  The file also holds the FORWARD conv's brick kernel (cl_conv_brick3_kernel, below).
#include <atomic>

#include "cl_args.h"
#include "dlka_kernels.h"

namespace dlka {

constexpr int BRICK_ROW = 144;   // bytes per brick voxel
static std::atomic<long> g_conv_brick_launches{0};   // dlka_conv_brick_launch_count (include/dlka.h): diagnostics

template <int NT, typename T, int WAVES, int MT = 1>   // NT = Cout / 32 column tiles per wave; T = storage of `out` (float | bf16_t; `aux` is fp32 when p.aux_f32 or T = float);
__global__ __launch_bounds__(64 * WAVES) void cl_conv_brick_kernel(IgemmArgs p, int TD, int TH)   // MT row tiles of 32 per wave (one weight-record fetch feeds all of them): WAVES x MT x 32 rows = a TD x TH x W tile
{
    DLKA_DYN_SMEM(unsigned char, brick);
    constexpr int NTHR = 64 * WAVES;
    constexpr int MAXI = MT == 2 ? 17 : 9;   // fill items (brick voxel, group of 8 planes) per thread: 4 * brick voxels <= MAXI * NTHR (checked by the launcher)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int W = p.W, BW = W + 2, BH = TH + 2, BD = TD + 2, nvox = BD * BH * BW;
    const int hblocks = p.H / TH, dblocks = p.D / TD;
    const int lgW = __builtin_ctz((unsigned)W), lgTH = __builtin_ctz((unsigned)TH);
    int bi = blockIdx.x;
    const int hb = bi % hblocks; bi /= hblocks;
    const int db = bi % dblocks;
    const int b = bi / dblocks;
    const int d0 = db * TD, h0 = hb * TH;
    const int nchunk = p.CinP / 32;
    const BufRsrc rin = make_rsrc(p.in, (size_t)p.B * p.CinReal * p.N * 4);
    const BufRsrc rw = make_rsrc(p.wp, (size_t)p.K * nchunk * 32 * p.NP * 4);
    const unsigned unit_bytes = (unsigned)(32 * p.NP) * 4u, seg_bytes = (unsigned)p.NP * 16u;
    const unsigned blane = (unsigned)(h * p.NP + i) * 16u;   // this lane's record inside segment (part, mf), column tile 0

    // this lane's A row: voxel r of the tile, at brick position (dl + 1, hl + 1, wl + 1) for the centre tap
    unsigned abase[MT];   // + tap offset + 16 mf (+ 64 for the low term)
#pragma unroll
    for (int u = 0; u < MT; ++u) {
        const int r = 32 * (MT * wave + u) + i, dl = r >> (lgW + lgTH), hl = (r >> lgW) & (TH - 1), wl = r & (W - 1);   // (W, TH: powers of two, launcher)
        abase[u] = (unsigned)(((dl * BH + hl) * BW + wl) * BRICK_ROW + 32 * h);
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int u = 0; u < MT; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[u][t][q] = 0.f;

#ifndef DLKA_BRICK_DEPTH
#define DLKA_BRICK_DEPTH 3
#endif
    constexpr int DEPTH = DLKA_BRICK_DEPTH;   // register ring of weight records: while tap t computes, taps t+1 .. t+DEPTH-1 are in flight (a tap is only
    f32x4 bring[DEPTH][4 * NT];               // 6 NT MFMAs = 192 NT cycles against an L2 round trip of several hundred)   [(part * 2 + mf) * NT + t]
    auto load_b = [&](int tap, int ck, f32x4 *bd) {
        const unsigned ub = (unsigned)(tap * nchunk + ck) * unit_bytes + blane;
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int t = 0; t < NT; ++t) bd[(part * 2 + mf) * NT + t] = buf_load_f32x4(rw, ub + (unsigned)((part * 2 + mf) * 2) * seg_bytes + (unsigned)t * 512u);
    };
    auto compute = [&](int tap, const f32x4 *bcur) {
        const int ti = tap / 9, tj = (tap - ti * 9) / 3, tk = tap - ti * 9 - tj * 3;
        const unsigned toff = (unsigned)(((ti * BH + tj) * BW + tk) * BRICK_ROW);
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
#pragma unroll
            for (int u = 0; u < MT; ++u) {
                const unsigned char *ap = brick + abase[u] + toff;
                const bf16x8 ahi = *reinterpret_cast<const bf16x8 *>(ap + 16 * mf), alo = *reinterpret_cast<const bf16x8 *>(ap + 64 + 16 * mf);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 bhi = __builtin_bit_cast(bf16x8, bcur[(0 * 2 + mf) * NT + t]), blo = __builtin_bit_cast(bf16x8, bcur[(1 * 2 + mf) * NT + t]);
                    acc[u][t] = mfma_32x32x16_bf16(alo, bhi, acc[u][t]);   // small terms first
                    acc[u][t] = mfma_32x32x16_bf16(ahi, blo, acc[u][t]);
                    acc[u][t] = mfma_32x32x16_bf16(ahi, bhi, acc[u][t]);
                }
            }
        }
    };

    // fill items of this thread, described once (the same for every plane chunk): consecutive threads = consecutive brick voxels of one group of 8
    // planes, i.e. coalesced along w.  src = byte offset of plane (8 pg) of the voxel inside batch b (DLKA_OOB: zero padding), dst = LDS byte offset.
    const float r_nvox = 1.0f / (float)nvox, r_plane = 1.0f / (float)(BH * BW), r_bw = 1.0f / (float)BW;
    unsigned fsrc[MAXI], fdst[MAXI];
    int fpg[MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int it = tid + j * NTHR;
        // (small non-negative integers: quotient by reciprocal multiply — (x + 0.5) / n never comes within 0.5 / n of an integer, far outside float error here)
        const int pg = (int)(((float)it + 0.5f) * r_nvox), vx = it - pg * nvox;
        const int dz = (int)(((float)vx + 0.5f) * r_plane), rem = vx - dz * (BH * BW);
        const int hy = (int)(((float)rem + 0.5f) * r_bw), wx = rem - hy * BW;
        const int zd = d0 + dz - 1, zh = h0 + hy - 1, zw = wx - 1;
        const bool ok = (pg < 4) & ((unsigned)zd < (unsigned)p.D) & ((unsigned)zh < (unsigned)p.H) & ((unsigned)zw < (unsigned)W);
        fsrc[j] = ok ? (unsigned)(((long)b * p.CinReal + 8 * pg) * p.N + ((long)zd * p.H + zh) * W + zw) * 4u : DLKA_OOB;
        fdst[j] = pg < 4 ? (unsigned)(vx * BRICK_ROW + pg * 16) : DLKA_OOB;
        fpg[j] = 8 * pg;
    }
    const unsigned plane_bytes = (unsigned)p.N * 4u;

#ifndef DLKA_BRICK_ABL   // TIMING-ONLY ablations (wrong results): 1 no fill, 2 no tap loop
#define DLKA_BRICK_ABL 0
#endif
    // gridDim.y > 1: the plane chunks are split over blockIdx.y (volumes too small to fill the chip with one workgroup per tile); the partial sums meet in
    // fp32 atomics on a ZEROED `out` (the caller's business, as for the tap-split kernels) and `aux` enters once
    const int cpw = (nchunk + (int)gridDim.y - 1) / (int)gridDim.y, ck_lo = (int)blockIdx.y * cpw, ck_hi = min(nchunk, ck_lo + cpw);
    for (int ck = ck_lo; ck < ck_hi; ++ck) {
#pragma unroll
        for (int s = 0; s < DEPTH - 1; ++s) load_b(s, ck, bring[s]);
        // ---- fill (split into the two bf16 terms on the way) ----
#pragma unroll
        for (int j = 0; j < ((DLKA_BRICK_ABL & 1) ? 0 : MAXI); ++j) {
            if (fdst[j] == DLKA_OOB) continue;
            float a[8];
            const unsigned s0 = fsrc[j] == DLKA_OOB ? DLKA_OOB : fsrc[j] + (unsigned)(ck * 32) * plane_bytes;
#pragma unroll
            for (int e = 0; e < 8; ++e)   // (planes beyond CinReal: padding of the contraction, read as zero)
                a[e] = buf_load_f32(rin, (s0 != DLKA_OOB && ck * 32 + fpg[j] + e < p.CinReal) ? s0 + (unsigned)e * plane_bytes : DLKA_OOB);
            bf16x8 hi, lo;
            split_bf16x8(a, hi, lo);
            *reinterpret_cast<bf16x8 *>(brick + fdst[j]) = hi;
            *reinterpret_cast<bf16x8 *>(brick + fdst[j] + 64) = lo;
        }
        __syncthreads();
        // ---- 27 taps from the brick ----
        // (K = 27 is a multiple of DEPTH; unconditional ring loads — see cl_conv_brick3_kernel)
#pragma unroll 1
        for (int tap = 0; tap < ((DLKA_BRICK_ABL & 2) ? 0 : p.K); tap += DEPTH) {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                load_b(min(tap + s + DEPTH - 1, p.K - 1), ck, bring[(s + DEPTH - 1) % DEPTH]);
#ifdef DLKA_BRICK_FENCE
                sched_fence();
#endif
                compute(tap + s, bring[s]);
            }
        }
        __syncthreads();   // every wave is done with this chunk's brick
    }

    // ---- epilogue (cl_conv_wave_kernel's): D layout col = lane & 31, row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5) ----
    T *outp = reinterpret_cast<T *>(p.out);
    const T *auxp = reinterpret_cast<const T *>(p.aux);
#pragma unroll
    for (int u = 0; u < MT; ++u)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = t * 32 + i;
        if (n >= p.Cout) continue;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int rr = 32 * (MT * wave + u) + (q & 3) + 8 * (q >> 2) + 4 * h;   // tile row -> voxel
            const int rd = rr >> (lgW + lgTH), rh = (rr >> lgW) & (TH - 1), rwv = rr & (W - 1);
            const long mr = (long)b * p.N + ((long)(d0 + rd) * p.H + h0 + rh) * W + rwv;
            const long o = mr * p.Cout + n;
            float val = acc[u][t][q];
            if (p.epi == 3 && blockIdx.y == 0) val += (sizeof(T) == 4 || p.aux_f32) ? p.aux[o] : act_load1(auxp, o);
            if (gridDim.y > 1) atomicAdd(p.out + o, val);   // (T = float by construction)
            else act_store1(outp, o, val);
        }
    }
}

// The tile a launch uses: 4-wave workgroups of 2 x TH x W = 128 voxels when their brick leaves room for TWO workgroups per CU (one stages its next chunk
// while the other runs its taps: measured against one 8-wave workgroup per CU in profiles/r06_notes.md), else 8 waves on 1 x TH x W = 256 voxels.
struct BrickTile { int waves, mt, TD, TH, csplit; size_t lds; };
#ifndef DLKA_BRICK_DEFAULT_CAND
#define DLKA_BRICK_DEFAULT_CAND 1   // 0: 4 waves (2 x TH x W tiles, two workgroups per CU), 1: 8 waves, 2: 4 waves of two row tiles
#endif
static bool brick_tile(const IgemmArgs &a, BrickTile &bt)
{
    if (a.W > 32 || a.W < 8 || 256 % a.W) return false;
    const char *force = getenv("DLKA_CONV_BRICK_WAVES");   // (A/B: 4 | 8 | 42 = 4 waves of two row tiles)
    const int fw = force ? atoi(force) : 0;
    for (int k = 0; k < 3; ++k) {
        const int cand = (DLKA_BRICK_DEFAULT_CAND + k) % 3;   // the default first, then whatever else fits the volume
        const int waves = cand == 1 ? 8 : 4, mt = cand == 2 ? 2 : 1;
        if (fw && fw != (mt == 2 ? 42 : waves)) continue;
        const char *etd = getenv("DLKA_CONV_BRICK_TD");   // (A/B: depth of the 8-wave tile)
        const int TD = (waves == 4 && mt == 1) ? 2 : (waves == 8 ? (etd ? atoi(etd) : ((a.D & 1) ? 1 : 2)) : 1), rows = 32 * waves * mt;   // (2 x 4 x 32: a fifth less halo than 1 x 8 x 32)
        if (TD < 1 || (TD & (TD - 1))) continue;
        if (rows % (TD * a.W)) continue;
        const int TH = rows / (TD * a.W);
        if (TH < 1 || (TH & (TH - 1)) || (a.W & (a.W - 1)) || a.H % TH || a.D % TD) continue;
        const size_t lds = (size_t)(TD + 2) * (TH + 2) * (a.W + 2) * BRICK_ROW;
        if (lds > ((waves == 4 && mt == 1) ? 80u : 160u) * 1024) continue;
        if (4 * (TD + 2) * (TH + 2) * (a.W + 2) > (mt == 2 ? 17 : 9) * 64 * waves) continue;   // MAXI fill items per thread
        // enough workgroups to fill the chip (DLKA_CONV_BRICK_MIN_WG lowers the bar for tests): one per tile, else — 4-wave tiles only — one per (tile, plane chunk)
        const char *mw = getenv("DLKA_CONV_BRICK_MIN_WG");
        const long need = mw ? atoi(mw) : 128, tiles = (long)a.M / rows, nchunk = a.CinP / 32;
        // The plane-chunk split (4-wave tiles, one workgroup per (tile, chunk)) exists for volumes below `need` tiles and is OFF by default: at (64, 16^3) it
        // measured 37.9 us against cl_igemm_kernel's 37.1 (profiles/r06_notes.md).  DLKA_CONV_BRICK_CSPLIT=2 switches it on (tests keep it alive).
        const char *fs = getenv("DLKA_CONV_BRICK_CSPLIT");
        const bool allow_split = fs && atoi(fs) > 1 && waves == 4 && mt == 1;
        int csplit = 1;
        if (tiles < need) {
            if (!allow_split || tiles * nchunk < need) continue;
            csplit = (int)nchunk;
        } else if (allow_split) csplit = (int)nchunk;
        bt.waves = waves; bt.mt = mt; bt.TD = TD; bt.TH = TH; bt.lds = lds; bt.csplit = csplit;
        return true;
    }
    return false;
}

// Planar input, channels-last output, 3^3 / stride 1 / padding 1 / dilation 1, two-term weights, epilogues 0 and 3, no tap split, volumes whose rows
// tile into 256-voxel workgroups (W <= 32, 256 % W == 0, H % (256 / W) == 0) and enough of them to fill the chip (DLKA_CONV_BRICK_MIN_WG lowers that
// bar so that small test shapes take this kernel; DLKA_CONV_BRICK=0 switches it off — both read per call: parity tests toggle them).
bool cl_conv_brick_supported(const IgemmArgs &a)
{
    const char *e = getenv("DLKA_CONV_BRICK");
    if (e && e[0] == '0') return false;
    if (a.K != 27 || a.kd != 3 || a.kh != 3 || a.kw != 3 || a.pd != 1 || a.ph != 1 || a.pw != 1 || a.dd != 1 || a.dh != 1 || a.dw != 1) return false;
    if (a.split_bf16 != 2 || a.a_packed || (a.epi != 0 && a.epi != 3) || a.bias || a.CinP % 32 || a.NP % 32 || a.Cout != a.NP) return false;
    if (a.NP != 32 && a.NP != 64) return false;
    BrickTile bt;
    if (!brick_tile(a, bt)) return false;
    if ((size_t)a.B * a.CinReal * a.N * 4 >= (1ull << 31) || (long)a.K * (a.CinP / 32) * 32 * a.NP * 4 >= (1l << 31)) return false;   // 32-bit buffer offsets
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The FORWARD offset-predict conv (C -> 81, channels-last fp32 in, planar out + bias, THREE-term split: its output decides floor() of the sampling
// positions) from the same kind of brick.  cl_igemm_kernel<0, 1, 3, 3> stages the weights of every (tap, chunk) unit through LDS behind a workgroup
// barrier and re-splits its A rows for every tap; its ablations (profiles/r06_notes.md) show the parts of a unit ADDING UP at two waves per SIMD: 26 us
// of MFMA + 13 A fetch + 9 weight fetch + 7 split arithmetic + a 26 us skeleton of staging, barriers and the transposing epilogue.  Here the three
// bf16 terms of the tile's halo are formed ONCE per element and live in LDS — a pass covers the 16 channels one MFMA k-step contracts (channels
// 8 mf .. 8 mf + 7 and 16 + 8 mf .. 16 + 8 mf + 7 of a 32-channel chunk, the k order of the prepared records):
//     brick[voxel][ hi: 16 x bf16 | mid | lo | 16 bytes of padding ]          112 bytes per voxel
// — and the tap loop is three 16-byte LDS reads, 3 NT record loads (register ring, two taps ahead) and 6 NT MFMAs, with no barrier inside a pass.
// Same six products per (term pair) and the same fp32 accumulation as the kernel it replaces; summation order: (chunk, k-half) outer, taps inner.
// ---------------------------------------------------------------------------------------------------------------------------------------------
constexpr int BRICK3_ROW = 112;

// Work split inside a workgroup (256 rows x NT column tiles): a wave owns ONE column tile and MT row tiles — wave = (row group, column tile) — so that a
// tap's weight records are fetched once per 32 MT rows: with every wave on all NT tiles the eight waves pulled 72 KB of records per tap round through the L1
// (1152 clocks at 64 B / clk against 1152 of MFMA: 70 us, profiles/r06_notes.md); here 8 / MT x NT waves pull 3 KB each.
template <int NT, int MT>
__global__ __launch_bounds__(64 * (8 / MT) * NT) void cl_conv_brick3_kernel(IgemmArgs p, int TD, int TH)
{
    DLKA_DYN_SMEM(unsigned char, brick);
    constexpr int WAVES = (8 / MT) * NT, NTHR = 64 * WAVES, MAXI = 5;   // fill items (brick voxel, 8-channel half) per thread: 2 * brick voxels <= MAXI * NTHR (launcher)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tcol = wave % NT, rg = wave / NT;   // this wave's column tile and row group
    const int i = lane & 31, h = lane >> 5;
    const int W = p.W, BW = W + 2, BH = TH + 2, BD = TD + 2, nvox = BD * BH * BW;
    const int hblocks = p.H / TH, dblocks = p.D / TD;
    const int lgW = __builtin_ctz((unsigned)W), lgTH = __builtin_ctz((unsigned)TH);
    int bi = blockIdx.x;
    const int hb = bi % hblocks; bi /= hblocks;
    const int db = bi % dblocks;
    const int b = bi / dblocks;
    const int d0 = db * TD, h0 = hb * TH;
    const int nchunk = p.CinP / 32;
    const BufRsrc rin = make_rsrc(p.in, (size_t)p.M * p.Cin * 4);
    const BufRsrc rw = make_rsrc(p.wp, (size_t)p.K * nchunk * 48 * p.NP * 4);
    const unsigned unit_bytes = (unsigned)(48 * p.NP) * 4u, seg_bytes = (unsigned)p.NP * 16u;
    const unsigned blane = (unsigned)(h * p.NP + tcol * 32 + i) * 16u;
    unsigned abase[MT];   // + tap offset + 32 * term
    long vox[MT];         // the row as a voxel of volume b (epilogue)
#pragma unroll
    for (int u = 0; u < MT; ++u) {
        const int r = 32 * (MT * rg + u) + i, dl = r >> (lgW + lgTH), hl = (r >> lgW) & (TH - 1), wl = r & (W - 1);
        abase[u] = (unsigned)(((dl * BH + hl) * BW + wl) * BRICK3_ROW + 16 * h);
        vox[u] = ((long)(d0 + dl) * p.H + h0 + hl) * W + wl;
    }

    f32x16 acc[MT];
#pragma unroll
    for (int u = 0; u < MT; ++u)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[u][q] = 0.f;

    constexpr int DEPTH = 3;
    f32x4 bring[DEPTH][3];   // [part]: the pass's k-half, this wave's column tile
    auto load_b = [&](int tap, int ck, int mf, f32x4 *bd) {
        const unsigned ub = (unsigned)(tap * nchunk + ck) * unit_bytes + blane;
#pragma unroll
        for (int part = 0; part < 3; ++part) bd[part] = buf_load_f32x4(rw, ub + (unsigned)((part * 2 + mf) * 2) * seg_bytes);
    };
    // A operands (three terms per row tile) are read from the brick ONE TAP AHEAD, behind the current tap's MFMAs in program order, so that the LDS latency is
    // covered by them instead of opening every tap (the compiler does not move LDS reads across the loop's back edge)
    bf16x8 a_nxt[MT][3];
    auto read_a = [&](int tap) {
        const int ti = tap / 9, tj = (tap - ti * 9) / 3, tk = tap - ti * 9 - tj * 3;
        const unsigned toff = (unsigned)(((ti * BH + tj) * BW + tk) * BRICK3_ROW);
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            const unsigned char *ap = brick + abase[u] + toff;
            a_nxt[u][0] = *reinterpret_cast<const bf16x8 *>(ap); a_nxt[u][1] = *reinterpret_cast<const bf16x8 *>(ap + 32); a_nxt[u][2] = *reinterpret_cast<const bf16x8 *>(ap + 64);
        }
    };
    auto compute = [&](int tap, const f32x4 *bcur) {
        const bf16x8 bhi = __builtin_bit_cast(bf16x8, bcur[0]), bmid = __builtin_bit_cast(bf16x8, bcur[1]), blo = __builtin_bit_cast(bf16x8, bcur[2]);
        bf16x8 ahi[MT], amid[MT], alo[MT];
#pragma unroll
        for (int u = 0; u < MT; ++u) { ahi[u] = a_nxt[u][0]; amid[u] = a_nxt[u][1]; alo[u] = a_nxt[u][2]; }
        read_a(min(tap + 1, p.K - 1));   // (behind the last tap: re-reads it)
        // product-major, row-tile-minor: consecutive MFMAs go to different accumulators; per accumulator the order is cl_igemm_kernel's (small terms first)
#pragma unroll
        for (int u = 0; u < MT; ++u) acc[u] = mfma_32x32x16_bf16(alo[u], bhi, acc[u]);
#pragma unroll
        for (int u = 0; u < MT; ++u) acc[u] = mfma_32x32x16_bf16(ahi[u], blo, acc[u]);
#pragma unroll
        for (int u = 0; u < MT; ++u) acc[u] = mfma_32x32x16_bf16(amid[u], bmid, acc[u]);
#pragma unroll
        for (int u = 0; u < MT; ++u) acc[u] = mfma_32x32x16_bf16(amid[u], bhi, acc[u]);
#pragma unroll
        for (int u = 0; u < MT; ++u) acc[u] = mfma_32x32x16_bf16(ahi[u], bmid, acc[u]);
#pragma unroll
        for (int u = 0; u < MT; ++u) acc[u] = mfma_32x32x16_bf16(ahi[u], bhi, acc[u]);
    };

    // fill items, described once: (brick voxel, half hh): 8 channels 16 hh + 8 mf .. of the voxel's row -> the voxel's three terms at k positions 8 hh ..
    const float r_nvox = 1.0f / (float)nvox, r_plane = 1.0f / (float)(BH * BW), r_bw = 1.0f / (float)BW;
    unsigned fsrc[MAXI], fdst[MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int it = tid + j * NTHR;
        const int hh = (int)(((float)it + 0.5f) * r_nvox), vx = it - hh * nvox;
        const int dz = (int)(((float)vx + 0.5f) * r_plane), rem = vx - dz * (BH * BW);
        const int hy = (int)(((float)rem + 0.5f) * r_bw), wx = rem - hy * BW;
        const int zd = d0 + dz - 1, zh = h0 + hy - 1, zw = wx - 1;
        const bool ok = (hh < 2) & ((unsigned)zd < (unsigned)p.D) & ((unsigned)zh < (unsigned)p.H) & ((unsigned)zw < (unsigned)W);
        fsrc[j] = ok ? (unsigned)((((long)b * p.N + ((long)zd * p.H + zh) * W + zw) * p.Cin + 16 * hh) * 4) : DLKA_OOB;
        fdst[j] = hh < 2 ? (unsigned)(vx * BRICK3_ROW + hh * 16) : DLKA_OOB;
    }

    for (int ck = 0; ck < nchunk; ++ck)
        for (int mf = 0; mf < 2; ++mf) {
#pragma unroll
            for (int s = 0; s < DEPTH - 1; ++s) load_b(s, ck, mf, bring[s]);
#ifndef DLKA_BRICK3_ABL   // TIMING-ONLY ablations (wrong results): 1 no fill, 2 no tap loop, 4 no output stores
#define DLKA_BRICK3_ABL 0
#endif
#pragma unroll
            for (int j = 0; j < ((DLKA_BRICK3_ABL & 1) ? 0 : MAXI); ++j) {
                if (fdst[j] == DLKA_OOB) continue;
                const unsigned s0 = fsrc[j] == DLKA_OOB ? DLKA_OOB : fsrc[j] + (unsigned)(ck * 32 + 8 * mf) * 4u;
                const f32x4 v0 = buf_load_f32x4(rin, s0), v1 = buf_load_f32x4(rin, s0 == DLKA_OOB ? DLKA_OOB : s0 + 16u);
                const float a[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                bf16x8 hi, mid, lo;
                split3_bf16x8(a, hi, mid, lo);
                *reinterpret_cast<bf16x8 *>(brick + fdst[j]) = hi;
                *reinterpret_cast<bf16x8 *>(brick + fdst[j] + 32) = mid;
                *reinterpret_cast<bf16x8 *>(brick + fdst[j] + 64) = lo;
            }
            __syncthreads();
            read_a(0);
            // (K = 27 is a multiple of DEPTH and the ring's loads are UNCONDITIONAL — behind the last tap they re-read it: a conditional load into the ring
            //  made the compiler merge the two paths with register copies behind s_waitcnt vmcnt(0), i.e. wait for the records it had just requested)
#pragma unroll 1
            for (int tap = 0; tap < ((DLKA_BRICK3_ABL & 2) ? 0 : p.K); tap += DEPTH) {
#pragma unroll
                for (int s = 0; s < DEPTH; ++s) {
                    load_b(min(tap + s + DEPTH - 1, p.K - 1), ck, mf, bring[(s + DEPTH - 1) % DEPTH]);
#ifdef DLKA_BRICK3_FENCE
                    sched_fence();   // (experiment: the record loads stay in front of the tap's MFMAs)
#endif
                    compute(tap + s, bring[s]);
                }
            }
            __syncthreads();   // every wave is done with this pass's brick
        }

    // ---- epilogue: planar output [B][Cout][N] (+ bias).  Each 32 x 32 tile goes through a wave-private LDS tile (the brick is free now) so that lanes run
    // over the tile's 32 rows — one W-run of voxels or several: 128 contiguous bytes of one plane per store at W = 32 ----
    float *Tt = reinterpret_cast<float *>(brick) + wave * (32 * 33);
#pragma unroll
    for (int u = 0; u < MT; ++u) {
        wave_sync();
#pragma unroll
        for (int q = 0; q < 16; ++q) Tt[((q & 3) + 8 * (q >> 2) + 4 * h) * 33 + i] = acc[u][q];
        wave_sync();
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const int col = 2 * cc + h, n = tcol * 32 + col;
            if (n >= p.Cout) continue;   // uniform per half-wave
            float val = Tt[i * 33 + col];
            if (p.bias) val += p.bias[n];
            if (!(DLKA_BRICK3_ABL & 4) || val == 123.456f) p.out[((long)b * p.Cout + n) * p.N + vox[u]] = val;
        }
    }
}

bool cl_conv_brick3_supported(const IgemmArgs &a)
{
    const char *e = getenv("DLKA_CONV_BRICK");
    if (e && (e[0] == '0' || e[0] == '2')) return false;   // 0: both brick kernels off, 2: the data gradient's only
    if (a.K != 27 || a.kd != 3 || a.kh != 3 || a.kw != 3 || a.pd != 1 || a.ph != 1 || a.pw != 1 || a.dd != 1 || a.dh != 1 || a.dw != 1) return false;
    if (a.split_bf16 != 3 || a.act_bf16 || a.epi != 0 || a.Cin % 32 || a.CinP != a.Cin || a.NP % 32 || a.NP > 96) return false;
    if (a.W > 32 || a.W < 8 || (a.W & (a.W - 1))) return false;
    const int TD = (a.D & 1) ? 1 : 2;
    if (256 % (TD * a.W)) return false;
    const int TH = 256 / (TD * a.W);
    if (TH < 1 || (TH & (TH - 1)) || a.H % TH) return false;
    const size_t nvox = (size_t)(TD + 2) * (TH + 2) * (a.W + 2);
    const int nt = a.NP / 32, mt = nt == 1 ? 1 : 2, waves = (8 / mt) * nt;
    if (nvox * BRICK3_ROW > 160 * 1024 || nvox * BRICK3_ROW < (size_t)waves * 32 * 33 * 4 || 2 * nvox > (size_t)5 * 64 * waves) return false;
    const char *mw = getenv("DLKA_CONV_BRICK_MIN_WG");
    if ((long)a.M / 256 < (mw ? atoi(mw) : 128)) return false;
    if ((size_t)a.M * a.Cin * 4 >= (1ull << 31) || (long)a.K * (a.CinP / 32) * 48 * a.NP * 4 >= (1l << 31)) return false;   // 32-bit buffer offsets
    return true;
}

int launch_cl_conv_brick3(const IgemmArgs &a, hipStream_t st)
{
    if (!cl_conv_brick3_supported(a)) return DLKA_ERR_UNSUPPORTED;
    const int TD = (a.D & 1) ? 1 : 2, TH = 256 / (TD * a.W), NT = a.NP / 32;
    const size_t lds = (size_t)(TD + 2) * (TH + 2) * (a.W + 2) * BRICK3_ROW;
#if !defined(HIPEMU)
    static std::atomic<uint64_t> attr_done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return DLKA_ERR_LAUNCH;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        const void *fns[3] = {reinterpret_cast<const void *>(cl_conv_brick3_kernel<1, 1>), reinterpret_cast<const void *>(cl_conv_brick3_kernel<2, 2>),
                              reinterpret_cast<const void *>(cl_conv_brick3_kernel<3, 2>)};
        for (int f = 0; f < 3; ++f)
            if (hipFuncSetAttribute(fns[f], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DLKA_ERR_LAUNCH;
        attr_done.fetch_or(bit, std::memory_order_release);
    }
#endif
    dim3 grid(a.B * (a.D / TD) * (a.H / TH));
    if (NT == 1) { auto k = cl_conv_brick3_kernel<1, 1>; DLKA_LAUNCH(k, grid, dim3(512), lds, st, a, TD, TH); }
    else if (NT == 2) { auto k = cl_conv_brick3_kernel<2, 2>; DLKA_LAUNCH(k, grid, dim3(512), lds, st, a, TD, TH); }
    else { auto k = cl_conv_brick3_kernel<3, 2>; DLKA_LAUNCH(k, grid, dim3(768), lds, st, a, TD, TH); }
    DLKA_CHECK_LAUNCH();
    g_conv_brick_launches.fetch_add(1, std::memory_order_relaxed);
    return DLKA_OK;
}

// plane-chunk split the brick kernel would use for this conv (1 = none; > 1: fp32 atomics into a zeroed `out`), 0 if it does not take it
int cl_conv_brick_split(const IgemmArgs &a)
{
    BrickTile bt;
    return (cl_conv_brick_supported(a) && brick_tile(a, bt)) ? bt.csplit : 0;
}

// DLKA_ERR_UNSUPPORTED: the caller takes cl_conv_wave_kernel / cl_igemm_kernel.
int launch_cl_conv_brick(const IgemmArgs &a, hipStream_t st)
{
    if (!cl_conv_brick_supported(a)) return DLKA_ERR_UNSUPPORTED;
    BrickTile bt;
    if (!brick_tile(a, bt)) return DLKA_ERR_UNSUPPORTED;
    const int NT = a.NP / 32;
    const int nwg = a.B * (a.D / bt.TD) * (a.H / bt.TH);
    const size_t lds = bt.lds;
#if !defined(HIPEMU)
    static std::atomic<uint64_t> attr_done{0};   // dynamic LDS above 64 KB: per function AND per device (cl_deform_bwd2.hip has the same pattern)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return DLKA_ERR_LAUNCH;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
#define DLKA_BRICK_FNS(WV, MTV) reinterpret_cast<const void *>(cl_conv_brick_kernel<1, float, WV, MTV>), reinterpret_cast<const void *>(cl_conv_brick_kernel<2, float, WV, MTV>), \
                           reinterpret_cast<const void *>(cl_conv_brick_kernel<1, bf16_t, WV, MTV>), reinterpret_cast<const void *>(cl_conv_brick_kernel<2, bf16_t, WV, MTV>)
        const void *fns[12] = {DLKA_BRICK_FNS(4, 1), DLKA_BRICK_FNS(8, 1), DLKA_BRICK_FNS(4, 2)};
#undef DLKA_BRICK_FNS
        for (int f = 0; f < 12; ++f)
            if (hipFuncSetAttribute(fns[f], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DLKA_ERR_LAUNCH;
        attr_done.fetch_or(bit, std::memory_order_release);
    }
#endif
    dim3 grid(nwg, bt.csplit), block(64 * bt.waves);
    if (bt.csplit > 1 && !a.out_zeroed && launch_zero(a.out, (size_t)a.M * a.Cout * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
#define DLKA_BRICK_GO(NTV, TT)                                                                                                   \
    {                                                                                                                            \
        if (bt.mt == 2) { auto k = cl_conv_brick_kernel<NTV, TT, 4, 2>; DLKA_LAUNCH(k, grid, block, lds, st, a, bt.TD, bt.TH); }      \
        else if (bt.waves == 4) { auto k = cl_conv_brick_kernel<NTV, TT, 4>; DLKA_LAUNCH(k, grid, block, lds, st, a, bt.TD, bt.TH); } \
        else { auto k = cl_conv_brick_kernel<NTV, TT, 8>; DLKA_LAUNCH(k, grid, block, lds, st, a, bt.TD, bt.TH); }               \
    }
    if (a.act_bf16 && bt.csplit == 1) { if (NT == 1) DLKA_BRICK_GO(1, bf16_t) else DLKA_BRICK_GO(2, bf16_t) }   // (split: `out` is the caller's fp32 accumulation buffer)
    else { if (NT == 1) DLKA_BRICK_GO(1, float) else DLKA_BRICK_GO(2, float) }
#undef DLKA_BRICK_GO
    DLKA_CHECK_LAUNCH();
    g_conv_brick_launches.fetch_add(1, std::memory_order_relaxed);
    return DLKA_OK;
}

}  // namespace dlka

extern "C" long dlka_conv_brick_launch_count(void) { return dlka::g_conv_brick_launches.load(std::memory_order_relaxed); }
