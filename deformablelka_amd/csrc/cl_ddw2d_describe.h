// Sampling description of one (pixel, tap) of the channels-last 2-D depthwise deformable kernels (cl_ddw2d.hip); in a header so that the
// index-parity debug entry (deform_conv.hip: sample_index2_kernel, path 2) runs THIS function, not a copy of it.
#pragma once
#include "deform_sample.h"

namespace dlka {

struct Tap2 {
    unsigned off[4];   // byte offsets of the 4 corner ROWS (pixel * C * 4); DLKA_OOB (-> loads 0) for corners outside the image
    float wt[4];       // bilinear weights; 0 for corners outside the image AND for samples outside the guard
    float ly, lx;
    unsigned okm;      // corners that contribute to the sample / receive grad_input (inside the image and the guard)
};

// torchvision bilinear_interpolate / get_coordinate_weight (deform_conv2d_kernel.cpp): corners (y0,x0) (y0,x1) (y1,x0) (y1,x1).
// The SAMPLE (forward, weight gradient, grad_input) is guarded: 0 unless -1 < q < size.  The coordinate weight (grad_offset) has no guard,
// only the per-corner bounds — the two differ exactly at q == -1, where the high corner is inside the image: its row offset stays valid
// here (the loads feed the derivative) while its bilinear weight and okm bit are cleared.
__device__ __forceinline__ void describe2(Tap2 &s, float oy, float ox, int b, int by, int bx, int H, int W, int N, int rowbytes)
{
    s.okm = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { s.off[q] = DLKA_OOB; s.wt[q] = 0.f; }
    int y0, x0;
    bool reach;   // some corner can lie inside the image
    const bool inside = sample_cell2(oy, ox, by, bx, H, W, y0, x0, s.ly, s.lx, reach);   // the one sampling rule (deform_sample.h)
    if (reach) {
        const float ly = s.ly, lx = s.lx, hy = 1.f - ly, hx = 1.f - lx;
        const bool vy0 = y0 >= 0, vy1 = y0 + 1 <= H - 1, vx0 = x0 >= 0, vx1 = x0 + 1 <= W - 1;
        const bool ok[4] = {vy0 && vx0, vy0 && vx1, vy1 && vx0, vy1 && vx1};
        const float w4[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
        const int base = b * N + y0 * W + x0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (ok[q]) {
                s.off[q] = (unsigned)(base + (q >> 1) * W + (q & 1)) * (unsigned)rowbytes;
                if (inside) { s.wt[q] = w4[q]; s.okm |= 1u << q; }
            }
        }
    }
}


}  // namespace dlka
