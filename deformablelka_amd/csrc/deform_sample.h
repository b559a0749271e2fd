// Sampling rule shared by every deformable kernel (3-D trilinear / 2-D bilinear).
//
// Semantics follow the reference exactly (paths relative to the reference repo):
//   coordinate  q = float(int base) + offset                      3D/dcn/src/cuda/deform_im2col_cuda.cuh:244-246
//   guard       q > -1 && q < size on every axis, else sample = 0 cuh:247
//   corners     floor(q) + {0,1}; low corner valid iff >= 0, high corner valid iff <= size-1   cuh:43-66
//   weights     (1-l | l) per axis, multiplied d*h*w, corners summed in order 000,001,...,111   cuh:67-70
// NOFF = 3: D3D layout, three offset channels (d,h,w) per tap.
// NOFF = 2: torchvision deform_conv2d layout, two offset channels (y,x) per tap; the volume has D == 1 and the
//           depth axis is not sampled at all (weight 1 on plane 0).
#pragma once
#include "dlka_common.h"

namespace dlka {

// ---------------------------------------------------------------------------------------------------------------------
// THE sampling rule.  sample_cell3 / sample_cell2 are the ONLY places in the library that form the coordinate q, test the guard
// and take the floor; every deformable kernel — general NCDHW (setup_tap), channels-last gathers (gather_describe3, cl_gather.h),
// the grad_input scatters (lane_tap and the fixed-point kernel, cl_deform_bwd2.hip), the 2-D depthwise kernels (describe2 and the
// window scatter, cl_ddw2d.hip) — calls them, and so do the debug entries dlka_deform_conv{3,2}d_sample_index_path that the
// parity tests compare BIT FOR BIT with the oracle ("integer sampling indices bit-exact", north_star).
// Outside the guard the cell is (0, 0, 0) with zero fractions: callers key everything on the returned flag.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool sample_cell3(float od, float oh, float ow, int bd, int bh, int bw, int D, int H, int W,
                                             int &zd, int &zh, int &zw, float &ld, float &lh, float &lw)
{
    const float qd_ = (float)bd + od, qh_ = (float)bh + oh, qw_ = (float)bw + ow;                                     // cuh:244-246 (Q8: int, then float)
    const bool inside = (qd_ > -1.f) & (qh_ > -1.f) & (qw_ > -1.f) & (qd_ < (float)D) & (qh_ < (float)H) & (qw_ < (float)W);   // cuh:247
    const float qd = inside ? qd_ : 0.f, qh = inside ? qh_ : 0.f, qw = inside ? qw_ : 0.f;
    const float fd_ = floorf(qd), fh_ = floorf(qh), fw_ = floorf(qw);                                                 // in [-1, size - 1]   cuh:30-32
    zd = (int)fd_; zh = (int)fh_; zw = (int)fw_;
    ld = qd - fd_; lh = qh - fh_; lw = qw - fw_;
    return inside;
}

// torchvision 0.12 deform_conv2d (un-vendored; restated): the SAMPLE is guarded (bilinear_interpolate: 0 unless -1 < q < size), the
// coordinate weight is not (get_coordinate_weight: per-corner bounds only) — `reach` says whether any corner can lie inside the image
// (q >= -1 && q < size on both axes); the two flags differ exactly at q == -1.  The cell is formed when `reach`.
__device__ __forceinline__ bool sample_cell2(float oy, float ox, int by, int bx, int H, int W, int &y0, int &x0, float &ly, float &lx, bool &reach)
{
    const float qy_ = (float)by + oy, qx_ = (float)bx + ox;
    reach = (qy_ >= -1.f) & (qx_ >= -1.f) & (qy_ < (float)H) & (qx_ < (float)W);
    const bool inside = reach & (qy_ > -1.f) & (qx_ > -1.f);
    const float qy = reach ? qy_ : 0.f, qx = reach ? qx_ : 0.f;
    const float fy = floorf(qy), fx = floorf(qx);
    y0 = (int)fy; x0 = (int)fx;
    ly = qy - fy; lx = qx - fx;
    return inside;
}

template <int NOFF>
struct TapSample {
    static constexpr int NC = (NOFF == 3) ? 8 : 4;
    int idx[NC];      // linear index into one (b,c) plane; 0 for dropped corners (always a legal address)
    float w[NC];      // interpolation weight; 0 for dropped corners
    unsigned ok;      // bit q set <=> corner q contributes (inside guard && per-corner bounds)
    unsigned cok;     // bit q set <=> corner q is inside the volume (ignores the guard) — torchvision's coord weight
    float fd[2], fh[2], fw[2];  // per-axis weights of the low / high corner
    bool inside;
    bool reach;       // NOFF == 2: some corner can lie inside the image (sample_cell2); NOFF == 3: == inside
    int z0[3];        // floor cell (d,h,w) when inside (2-D: when reach), else 0; read by the index-parity debug entry only
};

// offp points at offset channel (NOFF*tap) of this (b, dg) at output voxel v; consecutive channels are No apart.
template <int NOFF, typename T>
__device__ __forceinline__ void setup_tap(TapSample<NOFF> &s, const T *__restrict__ offp, int No,
                                          int base_d, int base_h, int base_w, int D, int H, int W)
{
    int d0 = 0, h0, w0;
    float ld = 0.f, lh, lw;
    bool inside, reach;
    if (NOFF == 3) {
        inside = sample_cell3(ldf(offp), ldf(offp + No), ldf(offp + 2 * (long)No), base_d, base_h, base_w, D, H, W, d0, h0, w0, ld, lh, lw);
        reach = inside;
    } else {
        inside = sample_cell2(ldf(offp), ldf(offp + No), base_h, base_w, H, W, h0, w0, lh, lw, reach);
    }
    s.inside = inside;
    s.reach = reach;
    s.z0[0] = d0; s.z0[1] = h0; s.z0[2] = w0;
    s.fd[0] = 1.f - ld; s.fd[1] = ld;
    s.fh[0] = 1.f - lh; s.fh[1] = lh;
    s.fw[0] = 1.f - lw; s.fw[1] = lw;
    unsigned ok = 0, cok = 0;
#pragma unroll
    for (int q = 0; q < TapSample<NOFF>::NC; ++q) {
        const int cd = (NOFF == 3) ? (q >> 2) & 1 : 0, ch = (q >> 1) & 1, cw = q & 1;
        const int zd = d0 + cd, zh = h0 + ch, zw = w0 + cw;
        bool v = (ch ? zh <= H - 1 : zh >= 0) && (cw ? zw <= W - 1 : zw >= 0);
        if (NOFF == 3) v = v && (cd ? zd <= D - 1 : zd >= 0);
        // a high corner can be valid by the reference's test while negative (e.g. floor = -2 is excluded by the
        // guard, floor = -1 gives high = 0): require a legal address as well.
        v = v && zh >= 0 && zh <= H - 1 && zw >= 0 && zw <= W - 1 && zd >= 0 && zd <= D - 1;
        v = v && reach;          // (outside `reach` the cell is the dummy (0, 0, 0): no corner counts)
        const bool use = v && inside;
        cok |= (v ? 1u : 0u) << q;
        ok |= (use ? 1u : 0u) << q;
        s.idx[q] = v ? (zd * H + zh) * W + zw : 0;
        const float wt = (NOFF == 3) ? s.fd[cd] * s.fh[ch] * s.fw[cw] : s.fh[ch] * s.fw[cw];
        s.w[q] = use ? wt : 0.f;
    }
    s.ok = ok;
    s.cok = cok;
}

// value of the sample in plane `xp`
template <int NOFF, typename T>
__device__ __forceinline__ float sample_value(const TapSample<NOFF> &s, const T *__restrict__ xp)
{
    float val = 0.f;
#pragma unroll
    for (int q = 0; q < TapSample<NOFF>::NC; ++q) {
        const float v = ((s.ok >> q) & 1u) ? ldf(xp + s.idx[q]) : 0.f;
        val = fmaf(s.w[q], v, val);
    }
    return val;
}

}  // namespace dlka
