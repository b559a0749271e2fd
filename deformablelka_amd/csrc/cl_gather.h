// Trilinear gather of channels-last volumes with a line-friendly lane layout.
//
// Measured (profiles/archive/r01g_gather_layout_ubench.txt): the vector-memory front end charges per cache line touched per
// instruction.  A wave that loads "lane = (row i of 32, half h): 64 bytes of row i" touches 32 rows per instruction and
// gathers at 9.6 TB/s; "lane = (row r of 8, piece p of 8): 16 bytes" covers 8 WHOLE 128-byte rows per instruction and
// gathers at 33 TB/s.  The deformable kernels therefore gather in the second layout and transpose through LDS into
// whatever layout their MFMA operand needs.
//
// Per (32-row tile, tap): lanes 0..31 compute the sampling description of row `lane` (deform_im2col_cuda.cuh:244-259)
// and publish it in a wave-private LDS table; lane (r = lane >> 3, p = lane & 7) then serves rows 8g + r, g = 0..3:
// 8 corner loads of 16 bytes (channels 4p .. 4p+3 of the current 32-channel chunk) per row.
#pragma once
#include "dlka_common.h"
#include "deform_sample.h"

namespace dlka {

struct RowDesc {
    int base;        // b*N + linear index of corner 000 (coordinates may be -1: only corners flagged in okm are addressed)
    unsigned okm;    // bit q: corner q inside the volume and the sample inside the guard
    float ld, lh, lw;
    int zd, zh, zw;  // floor cell (read by the index-parity debug entry only; dead in the hot kernels)
};

constexpr int GATHER_DESC_WORDS = 8;   // LDS words per row in the description table

// Sampling rule for one (row, tap); q = base + offset.  Returns okm == 0 when the sample is outside the guard.
__device__ __forceinline__ RowDesc gather_describe3(float od, float oh, float ow, long N, int b, int bd, int bh, int bw, int D, int H, int W)
{
    RowDesc r;
    r.base = 0; r.okm = 0;
    int zd, zh, zw;
    const bool inside = sample_cell3(od, oh, ow, bd, bh, bw, D, H, W, zd, zh, zw, r.ld, r.lh, r.lw);   // the one sampling rule (deform_sample.h)
    r.zd = zd; r.zh = zh; r.zw = zw;
    if (inside) {  // floor in [-1, size-1]
        r.base = b * (int)N + (zd * H + zh) * W + zw;
        const unsigned vd0 = zd >= 0, vd1 = zd + 1 <= D - 1, vh0 = zh >= 0, vh1 = zh + 1 <= H - 1, vw0 = zw >= 0, vw1 = zw + 1 <= W - 1;
        unsigned okm = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned ok = (((q >> 2) & 1) ? vd1 : vd0) & (((q >> 1) & 1) ? vh1 : vh0) & ((q & 1) ? vw1 : vw0);
            okm |= ok << q;
        }
        r.okm = okm;
    }
    return r;
}

__device__ __forceinline__ RowDesc gather_describe(const float *__restrict__ off, long N, int b, int bd, int bh, int bw, int D, int H, int W)
{
    return gather_describe3(off[0], off[N], off[2 * N], N, b, bd, bh, bw, D, H, W);
}

// What a lane reads back for a row it serves.  The table does not hold RowDesc as it is: the publisher (one lane per row) does the work that
// would otherwise be repeated by the 8 lanes of every row for every corner — round 3 found the gather kernels bound by VALU issue, and ~55 % of
// the forward kernel's vector instructions were corner ADDRESS arithmetic (v_mul_lo, v_add, v_and, v_cmp, v_cndmask per corner; scripts/isa_loop_mix.py):
//   bbyte = base * rowbytes    (the 32-bit product wraps for base < 0 — floor cell -1 next to the volume's first rows; the sum with a VALID
//                               corner's delta wraps back to the true offset)
//   nokm  = ~okm               (bit q SET <=> corner q is dropped)
// so that a corner's load offset is add + shift + and_or (gather_offset below).
struct RowLook {
    unsigned bbyte, nokm;
    float ld, lh, lw;
};

__device__ __forceinline__ void gather_publish(float *tab, int row, const RowDesc &r, int rowbytes)
{
    f32x4 a, b;
    a[0] = __uint_as_float((unsigned)r.base * (unsigned)rowbytes); a[1] = __uint_as_float(~r.okm); a[2] = r.ld; a[3] = r.lh;
    b[0] = r.lw; b[1] = 0.f; b[2] = 0.f; b[3] = 0.f;
    reinterpret_cast<f32x4 *>(tab + row * GATHER_DESC_WORDS)[0] = a;
    reinterpret_cast<f32x4 *>(tab + row * GATHER_DESC_WORDS)[1] = b;
}

__device__ __forceinline__ RowLook gather_lookup(const float *tab, int row)
{
    const f32x4 a = reinterpret_cast<const f32x4 *>(tab + row * GATHER_DESC_WORDS)[0];
    RowLook r;
    r.bbyte = __float_as_uint(a[0]); r.nokm = __float_as_uint(a[1]); r.ld = a[2]; r.lh = a[3];
    r.lw = tab[row * GATHER_DESC_WORDS + 4];
    return r;
}

// byte offset of corner q of a looked-up row, channel byte offset cbyte; >= DLKA_OOB (-> loads 0) for dropped corners.  The corner's delta
// (q_d HW + q_h W + q_w) * rowbytes is wave-uniform (a scalar register); a dropped corner gets bit 31 set — valid offsets stay below 2 GB.
__device__ __forceinline__ unsigned gather_offset(const RowLook &r, int q, int HW, int W, int rowbytes, unsigned cbyte)
{
    const unsigned delta = (unsigned)(((q >> 2) & 1) * HW + ((q >> 1) & 1) * W + (q & 1)) * (unsigned)rowbytes;
    const unsigned off = (r.bbyte + cbyte) + delta;
    return ((r.nokm << (31 - q)) & DLKA_OOB) | off;
}

// ---- the same table with the eight trilinear weights of the row behind the description (16 words per row): kernels that only need the SAMPLE
// (forward, gathering weight gradient) read them back instead of forming them per lane (18 vector instructions per row group and lane) ----
constexpr int GATHER_DESCW_WORDS = 16;

__device__ __forceinline__ void gather_weights(const RowDesc &r, float w[8]);

__device__ __forceinline__ void gather_publish_w(float *tab, int row, const RowDesc &r, int rowbytes)
{
    float w[8];
    gather_weights(r, w);
    f32x4 a, b, c, d;
    a[0] = __uint_as_float((unsigned)r.base * (unsigned)rowbytes); a[1] = __uint_as_float(~r.okm); a[2] = r.ld; a[3] = r.lh;
    b[0] = r.lw; b[1] = 0.f; b[2] = 0.f; b[3] = 0.f;
    c[0] = w[0]; c[1] = w[1]; c[2] = w[2]; c[3] = w[3];
    d[0] = w[4]; d[1] = w[5]; d[2] = w[6]; d[3] = w[7];
    f32x4 *dst = reinterpret_cast<f32x4 *>(tab + row * GATHER_DESCW_WORDS);
    dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d;
}

// description only / weights only of a 16-word row: the kernels read each where it is needed instead of holding 13 registers per row group
// across their MFMA phase
__device__ __forceinline__ RowLook gather_lookup_d(const float *tab, int row)
{
    const f32x4 a = reinterpret_cast<const f32x4 *>(tab + row * GATHER_DESCW_WORDS)[0];
    RowLook r;
    r.bbyte = __float_as_uint(a[0]); r.nokm = __float_as_uint(a[1]); r.ld = a[2]; r.lh = a[3];
    r.lw = 0.f;   // (not read back: the weights are)
    return r;
}
__device__ __forceinline__ void gather_lookup_weights(const float *tab, int row, float w[8])
{
    const f32x4 *src = reinterpret_cast<const f32x4 *>(tab + row * GATHER_DESCW_WORDS);
    const f32x4 c = src[2], d = src[3];
    w[0] = c[0]; w[1] = c[1]; w[2] = c[2]; w[3] = c[3]; w[4] = d[0]; w[5] = d[1]; w[6] = d[2]; w[7] = d[3];
}

__device__ __forceinline__ RowLook gather_lookup_w(const float *tab, int row, float w[8])
{
    const f32x4 *src = reinterpret_cast<const f32x4 *>(tab + row * GATHER_DESCW_WORDS);
    const f32x4 a = src[0], c = src[2], d = src[3];
    RowLook r;
    r.bbyte = __float_as_uint(a[0]); r.nokm = __float_as_uint(a[1]); r.ld = a[2]; r.lh = a[3];
    r.lw = tab[row * GATHER_DESCW_WORDS + 4];
    w[0] = c[0]; w[1] = c[1]; w[2] = c[2]; w[3] = c[3]; w[4] = d[0]; w[5] = d[1]; w[6] = d[2]; w[7] = d[3];
    return r;
}

// Lane layout of the gather, by activation storage type.  fp32: a 32-channel chunk of a row is 128 bytes — lane = (row of 8, 16-byte piece
// of 8), four row groups cover a 32-row tile.  bf16: the chunk is 64 bytes — lane = (row of 16, 16-byte piece of 4), two row groups; the
// per-lane load is still 16 bytes (8 channels), so a 32-row tile takes HALF the load instructions (8-byte loads in the fp32 layout were
// measured slower than fp32 itself: 117 vs 102 us for the stage-0 forward, profiles/archive/r03c).
template <typename T> struct GatherGeom { static constexpr int NG = 4, RPI = 8, PE = 4, PSHIFT = 3; };
template <> struct GatherGeom<bf16_t> { static constexpr int NG = 2, RPI = 16, PE = 8, PSHIFT = 2; };
// What one lane holds of one corner row: fp32 = 4 channels; bf16 = 8 channels as the four RAW 32-bit words the load returned — converted
// by get() where they are consumed.  (A conversion next to its load makes the compiler wait for that load on the spot, which serialises
// the "next unit's corners fly under this unit's MFMAs" prefetch; raw words also halve the registers of the pieces in flight.)
template <typename T> struct GatherPiece {
    f32x4 w;
    __device__ __forceinline__ f32x4 get(int) const { return w; }
    __device__ __forceinline__ float elem(int k) const { return w[k]; }   // channel k of the piece
};
template <> struct GatherPiece<bf16_t> {
    f32x4 w;   // bit patterns: word k = channels 2k (low half) and 2k + 1 (high half)
    __device__ __forceinline__ f32x4 get(int v) const   // channels 4v .. 4v + 3
    {
        const unsigned a = __float_as_uint(w[2 * v]), b = __float_as_uint(w[2 * v + 1]);
        f32x4 r;
        r[0] = __uint_as_float(a << 16); r[1] = __uint_as_float(a & 0xffff0000u); r[2] = __uint_as_float(b << 16); r[3] = __uint_as_float(b & 0xffff0000u);
        return r;
    }
    __device__ __forceinline__ float elem(int k) const   // channel k (0 .. 7) of the piece
    {
        const unsigned a = __float_as_uint(w[k >> 1]);
        return __uint_as_float((k & 1) ? (a & 0xffff0000u) : (a << 16));
    }
};
template <typename T> __device__ __forceinline__ GatherPiece<T> gather_load(BufRsrc r, unsigned byteoff)
{
    GatherPiece<T> p;
    p.w = buf_load_f32x4(r, byteoff);   // 16 bytes either way
    return p;
}

template <typename R> __device__ __forceinline__ void gather_weights_of(const R &r, float w[8])
{
    const float fd[2] = {1.f - r.ld, r.ld}, fh[2] = {1.f - r.lh, r.lh}, fw[2] = {1.f - r.lw, r.lw};
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = fd[(q >> 2) & 1] * fh[(q >> 1) & 1] * fw[q & 1];
}
__device__ __forceinline__ void gather_weights(const RowLook &r, float w[8]) { gather_weights_of(r, w); }
__device__ __forceinline__ void gather_weights(const RowDesc &r, float w[8])
{
    const float fd[2] = {1.f - r.ld, r.ld}, fh[2] = {1.f - r.lh, r.lh}, fw[2] = {1.f - r.lw, r.lw};
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = fd[(q >> 2) & 1] * fh[(q >> 1) & 1] * fw[q & 1];
}

// ---- the grad_input window kernels' per-(voxel, tap) description (cl_deform_bwd2.hip) ----
struct LaneTap {
    int zd, zh, zw;        // floor corner (may be -1)
    float ld, lh, lw;      // fractions
    unsigned okm;          // bit q set <=> corner q is inside the volume and the sample passes the guard
};

// Sampling rule of deform_im2col_cuda.cuh:244-259 for one (voxel, tap); identical to setup_tap<3> (deform_sample.h).
__device__ __forceinline__ void lane_tap(LaneTap &s, float od, float oh, float ow, int bd, int bh, int bw, int D, int H, int W)
{
    s.okm = 0;
    const bool inside = sample_cell3(od, oh, ow, bd, bh, bw, D, H, W, s.zd, s.zh, s.zw, s.ld, s.lh, s.lw);   // the one sampling rule (deform_sample.h)
    if (inside) {  // floor in [-1, size-1]
        unsigned okm = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int cd = (q >> 2) & 1, ch = (q >> 1) & 1, cw = q & 1;
            const bool ok = (cd ? s.zd + 1 <= D - 1 : s.zd >= 0) && (ch ? s.zh + 1 <= H - 1 : s.zh >= 0) && (cw ? s.zw + 1 <= W - 1 : s.zw >= 0);
            okm |= (ok ? 1u : 0u) << q;
        }
        s.okm = okm;
    }
}

}  // namespace dlka
