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

template <int NOFF>
struct TapSample {
    static constexpr int NC = (NOFF == 3) ? 8 : 4;
    int idx[NC];      // linear index into one (b,c) plane; 0 for dropped corners (always a legal address)
    float w[NC];      // interpolation weight; 0 for dropped corners
    unsigned ok;      // bit q set <=> corner q contributes (inside guard && per-corner bounds)
    unsigned cok;     // bit q set <=> corner q is inside the volume (ignores the guard) — torchvision's coord weight
    float fd[2], fh[2], fw[2];  // per-axis weights of the low / high corner
    bool inside;
    int z0[3];        // floor cell (d,h,w), clamped to [-2, size]; read by the index-parity debug entry only
};

// offp points at offset channel (NOFF*tap) of this (b, dg) at output voxel v; consecutive channels are No apart.
template <int NOFF, typename T>
__device__ __forceinline__ void setup_tap(TapSample<NOFF> &s, const T *__restrict__ offp, int No,
                                          int base_d, int base_h, int base_w, int D, int H, int W)
{
    float qd = 0.f, qh, qw;
    if (NOFF == 3) {
        qd = (float)base_d + ldf(offp);
        qh = (float)base_h + ldf(offp + No);
        qw = (float)base_w + ldf(offp + 2 * (long)No);
    } else {
        qh = (float)base_h + ldf(offp);
        qw = (float)base_w + ldf(offp + No);
    }
    bool inside = (qh > -1.f) && (qw > -1.f) && (qh < (float)H) && (qw < (float)W);
    if (NOFF == 3) inside = inside && (qd > -1.f) && (qd < (float)D);
    s.inside = inside;
    const float fl_d = floorf(qd), fl_h = floorf(qh), fl_w = floorf(qw);
    // keep the int conversion safe for absurd offsets: values outside the guard never index memory
    const int d0 = (NOFF == 3) ? (int)fminf(fmaxf(fl_d, -2.f), (float)D) : 0;
    const int h0 = (int)fminf(fmaxf(fl_h, -2.f), (float)H);
    const int w0 = (int)fminf(fmaxf(fl_w, -2.f), (float)W);
    const float ld = qd - fl_d, lh = qh - fl_h, lw = qw - fl_w;
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
