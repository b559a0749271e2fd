// A-operand fetch of the channels-last implicit-GEMM kernels (cl_igemm.hip, cl_conv_wave.hip).
#pragma once
#include "deform_sample.h"
#include "cl_args.h"

namespace dlka {

// Fetches this lane's 16 A values (channels ck*32 + 16*h + [0,16) of row m) for one (tap, chunk) unit.
// Every load is an unconditional buffer load: zero padding, rows beyond M and corners outside the volume read offset
// DLKA_OOB and come back as 0, so there is no branch around a load and the loads of unit u+1 stay in flight under
// the MFMAs of unit u.
template <int AMODE, typename T = float>   // T: storage of a channels-last `in` (AMODE 0); planar inputs (AMODE 2) are always fp32
struct ARow {
    int cur_tap;
    unsigned rowoff;     // AMODE 0: byte offset of the neighbour row; AMODE 2: of the neighbour voxel in plane 0; DLKA_OOB if padded
    unsigned coff[8];    // AMODE 1: byte offsets of the 8 corner rows (DLKA_OOB for dropped corners)
    float cw[8];         // AMODE 1: corner weights
    __device__ __forceinline__ ARow() : cur_tap(-1), rowoff(DLKA_OOB) {}

    __device__ __forceinline__ void fetch(const IgemmArgs &p, const BufRsrc &rin, int tap, int ck, int h, bool row_ok, int b, int v, int d0, int h0, int w0, float *a)
    {
        if (tap != cur_tap) {   // wave-uniform
            cur_tap = tap;
            int ti, tj, tk;
            tap_decode(tap, p.kw, p.kh, ti, tj, tk);
            if (AMODE == 0 || AMODE == 2) {
                const int zd = d0 + ti * p.dd - p.pd, zh = h0 + tj * p.dh - p.ph, zw = w0 + tk * p.dw - p.pw;
                const bool ok = row_ok & ((unsigned)zd < (unsigned)p.D) & ((unsigned)zh < (unsigned)p.H) & ((unsigned)zw < (unsigned)p.W);
                const int lin = (zd * p.H + zh) * p.W + zw;
                rowoff = !ok ? DLKA_OOB : (AMODE == 0 ? (unsigned)((b * p.N + lin) * p.Cin) * (unsigned)sizeof(T) : (unsigned)(b * p.CinReal * p.N + lin) * 4u);
            } else {
                TapSample<3> s;
                if (row_ok) {
                    const float *offp = p.off + ((long)b * 3 * p.K + 3 * tap) * p.N + v;
                    setup_tap<3>(s, offp, p.N, d0 + ti * p.dd - p.pd, h0 + tj * p.dh - p.ph, w0 + tk * p.dw - p.pw, p.D, p.H, p.W);
                } else {
                    s.ok = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) { s.idx[q] = 0; s.w[q] = 0.f; }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    coff[q] = ((s.ok >> q) & 1u) ? (unsigned)((b * p.N + s.idx[q]) * p.Cin) * 4u : DLKA_OOB;
                    cw[q] = s.w[q];
                }
            }
        }
        const int c0 = ck * 32 + 16 * h;
        if (AMODE == 0 && sizeof(T) == 2) {
            // bf16 rows feed the bf16 matrix cores as they are: a[0..7] carry the 16 channels as RAW packed words (two 16-byte loads, no
            // conversion; the consumer uses bf16x8_from_words(a + 4 * mf)); a[8..15] are unused
            const f32x4 t0 = buf_load_f32x4(rin, rowoff == DLKA_OOB ? DLKA_OOB : rowoff + (unsigned)c0 * 2u);
            const f32x4 t1 = buf_load_f32x4(rin, rowoff == DLKA_OOB ? DLKA_OOB : rowoff + (unsigned)c0 * 2u + 16u);
            a[0] = t0[0]; a[1] = t0[1]; a[2] = t0[2]; a[3] = t0[3]; a[4] = t1[0]; a[5] = t1[1]; a[6] = t1[2]; a[7] = t1[3];
        } else if (AMODE == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 t = act_buf_load4<T>(rin, rowoff + (unsigned)(c0 + 4 * e) * (unsigned)sizeof(T));
                a[4 * e] = t[0]; a[4 * e + 1] = t[1]; a[4 * e + 2] = t[2]; a[4 * e + 3] = t[3];
            }
        } else if (AMODE == 2) {
            if (p.a_packed) {   // uniform: all CinP planes exist (zero padded): one per-lane offset, the plane stride rides in the scalar offset
                const unsigned vo = rowoff == DLKA_OOB ? DLKA_OOB : rowoff + (unsigned)(16 * h * p.N) * 4u;
                const unsigned so = (unsigned)(ck * 32 * p.N) * 4u, ps = (unsigned)p.N * 4u;
#pragma unroll
                for (int e = 0; e < 16; ++e) a[e] = buf_load_f32_s(rin, vo, so + (unsigned)e * ps);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    a[e] = buf_load_f32(rin, (c0 + e < p.CinReal) ? rowoff + (unsigned)((c0 + e) * p.N) * 4u : DLKA_OOB);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) a[e] = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {   // dropped corners (outside the volume / the guard) read 0 with weight 0
                const float wq = cw[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 t = buf_load_f32x4(rin, coff[q] + (unsigned)(c0 + 4 * e) * 4u);
                    a[4 * e] = fmaf(wq, t[0], a[4 * e]); a[4 * e + 1] = fmaf(wq, t[1], a[4 * e + 1]);
                    a[4 * e + 2] = fmaf(wq, t[2], a[4 * e + 2]); a[4 * e + 3] = fmaf(wq, t[3], a[4 * e + 3]);
                }
            }
        }
    }
};

}  // namespace dlka
