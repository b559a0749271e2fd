// Channels-last backward of the 3-D deformable convolution w.r.t. input and offsets — LDS-window variant.
//
// Measured on MI355X (profiles/r01c_*): the first version of this kernel (cl_deform_bwd.hip) scatters every
// Col * w_corner product with a global fp32 atomic; 27 taps x 8 corners = 216 atomics per input element put the
// kernel on the L2 atomic-unit ceiling (~2.8e11 dword atomics/s: 1.6 ms for C=32, 32^3, B=2) no matter how the
// offsets are distributed.  Here a workgroup owns a 4x4x8 brick of OUTPUT voxels and a 16-channel slice, walks all
// its taps, and accumulates the scatter in an LDS window that covers the brick plus a 3-voxel halo
// (10 x 10 x 14 voxels x 16 ch fp32 = 89.6 KB of the 160 KB LDS; ds_add_f32 runs at LDS rate).  Only corners that
// fall outside the window (|tap + offset| > ~2 voxels beyond the brick: a few % for unit-variance offsets) still go
// to global atomics, and the window is flushed once (non-zero cells only).
//
// Col[v][c] = sum_co G[v][co] * W[co][c][tap] comes from v_mfma_f32_16x16x4_f32 (two 16-row tiles per wave, N = the
// 16-channel slice); D layout: lane & 15 = channel, rows (lane >> 4) * 4 + r.
#include "cl_args.h"
#include "dlka_kernels.h"

namespace dlka {

namespace {
constexpr int BD = 4, BH = 4, BW = 8;        // brick of output voxels (one wave per d-slice: 4 x 8 = 32 rows)
constexpr int HALO = 3;
constexpr int WD = BD + 2 * HALO, WH = BH + 2 * HALO, WW = BW + 2 * HALO;   // 10 x 10 x 14
constexpr int WVOX = WD * WH * WW;           // 1400
constexpr int CS = 16;                       // channel slice
}  // namespace

__global__ __launch_bounds__(256) void cl_deform_bwd_lds_kernel(DeformBwdArgs p, int nbw, int nbh, int nbd, int taps_per_group)
{
    __shared__ __attribute__((aligned(16))) float Win[WVOX * CS];       // 89,600 B
    __shared__ __attribute__((aligned(16))) float Bs[32 * CS];          // [co chunk 32][16 ch]
    __shared__ __attribute__((aligned(16))) float Sx[4][32][8];         // per wave/row: packed corner origin, mask, ld, lh, lw, batch
    __shared__ float Rd[4][96][CS + 1];                                 // per wave transpose-reduce buffer for grad_offset
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cj = lane & 15, kq = lane >> 4;                           // MFMA 16x16x4 roles
    // ---- which brick / channel slice / tap group ----
    int bid = blockIdx.x;
    const int bw_i = bid % nbw; bid /= nbw;
    const int bh_i = bid % nbh; bid /= nbh;
    const int bd_i = bid % nbd; const int b = bid / nbd;
    const int cs = blockIdx.y;                                          // channels cs*16 .. cs*16+15
    const int tap_lo = blockIdx.z * taps_per_group, tap_hi = min(p.K, tap_lo + taps_per_group);
    const int bd0 = bd_i * BD, bh0 = bh_i * BH, bw0 = bw_i * BW;
    const int HW = p.H * p.W;
    const int nkc = p.CoutP / 32;

    // this lane's "setup row": row i of the wave = voxel (bd0 + wave, bh0 + (i >> 3), bw0 + (i & 7))
    const int si = lane & 31;
    const int vd = bd0 + wave, vh = bh0 + (si >> 3), vw = bw0 + (si & 7);
    const bool srow_ok = vd < p.D && vh < p.H && vw < p.W;
    const int sv = (vd * p.H + vh) * p.W + vw;

    for (int e = tid; e < WVOX * CS; e += 256) Win[e] = 0.f;

    for (int tap = tap_lo; tap < tap_hi; ++tap) {
        const int tk = tap % p.kw, tj = (tap / p.kw) % p.kh, ti = tap / (p.kw * p.kh);
        // ---- sampling description of the 32 rows (lanes 0..31 of each wave publish) ----
        if (lane < 32) {
            int org = 0;
            unsigned okm = 0;
            float ld = 0.f, lh = 0.f, lw = 0.f;
            if (srow_ok) {
                const float *offp = p.off + ((long)b * 3 * p.K + 3 * tap) * p.N + sv;
                const float qd = (float)(vd + ti * p.dd - p.pd) + offp[0];
                const float qh = (float)(vh + tj * p.dh - p.ph) + offp[p.N];
                const float qw = (float)(vw + tk * p.dw - p.pw) + offp[2 * (long)p.N];
                const bool inside = qd > -1.f && qh > -1.f && qw > -1.f && qd < (float)p.D && qh < (float)p.H && qw < (float)p.W;
                if (inside) {  // floor in [-1, size-1]
                    const float fd_ = floorf(qd), fh_ = floorf(qh), fw_ = floorf(qw);
                    const int zd = (int)fd_, zh = (int)fh_, zw = (int)fw_;
                    ld = qd - fd_; lh = qh - fh_; lw = qw - fw_;
                    org = ((zd + 1) << 20) | ((zh + 1) << 10) | (zw + 1);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int cd = (q >> 2) & 1, ch = (q >> 1) & 1, cw = q & 1;
                        const bool ok = (cd ? zd + 1 <= p.D - 1 : zd >= 0) && (ch ? zh + 1 <= p.H - 1 : zh >= 0) &&
                                        (cw ? zw + 1 <= p.W - 1 : zw >= 0);
                        okm |= (ok ? 1u : 0u) << q;
                    }
                }
            }
            float *sx = &Sx[wave][si][0];
            sx[0] = __int_as_float(org);
            sx[1] = __int_as_float((int)okm);
            sx[2] = ld; sx[3] = lh; sx[4] = lw;
        }
        // ---- Col tile: two 16-row tiles x 16 channels, K = Cout ----
        f32x4 acc[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rt][r] = 0.f;
        for (int kc = 0; kc < nkc; ++kc) {
            __syncthreads();   // Bs consumed; Sx published; Win zeroed (first iteration)
            if (tid < 128) {   // 32 rows (co) x 16 floats: 128 float4
                const int rr = tid >> 2, c4 = tid & 3;
                reinterpret_cast<f32x4 *>(Bs)[tid] =
                    *reinterpret_cast<const f32x4 *>(p.wp + ((long)tap * p.CoutP + kc * 32 + rr) * p.C + cs * CS + c4 * 4);
            }
            // A values: G[row = rt*16 + cj][co = kc*32 + 8*kq + s], s = 0..7   (k-slot kq of step s)
            float ga[2][8];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int row = rt * 16 + cj;
                const int gd = bd0 + wave, gh = bh0 + (row >> 3), gw_ = bw0 + (row & 7);
                const bool ok = gd < p.D && gh < p.H && gw_ < p.W && (kc * 32 + 8 * kq) < p.Cout;
                if (ok) {
                    const long mrow = (long)b * p.N + (long)(gd * p.H + gh) * p.W + gw_;
                    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(p.g + mrow * p.Cout + kc * 32 + 8 * kq);
                    const f32x4 t0 = g4[0], t1 = g4[1];
                    ga[rt][0] = t0[0]; ga[rt][1] = t0[1]; ga[rt][2] = t0[2]; ga[rt][3] = t0[3];
                    ga[rt][4] = t1[0]; ga[rt][5] = t1[1]; ga[rt][6] = t1[2]; ga[rt][7] = t1[3];
                } else {
#pragma unroll
                    for (int s = 0; s < 8; ++s) ga[rt][s] = 0.f;
                }
            }
            __syncthreads();
            const float *brow = Bs + (8 * kq) * CS + cj;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float bvv = brow[s * CS];
                acc[0] = mfma_16x16x4(ga[0][s], bvv, acc[0]);
                acc[1] = mfma_16x16x4(ga[1][s], bvv, acc[1]);
            }
        }
        // ---- scatter + offset-gradient partials: lane = (channel cj, row group kq), 8 rows ----
        const int ch_g = cs * CS + cj;   // global channel
        float pd_[8], ph_[8], pw_[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int rt = rr >> 2, r = rr & 3;
            const int row = rt * 16 + kq * 4 + r;
            const f32x4 s0 = *reinterpret_cast<const f32x4 *>(&Sx[wave][row][0]);
            const float s1x = Sx[wave][row][4];
            const unsigned okm = (unsigned)__float_as_int(s0[1]);
            float dd_ = 0.f, dh_ = 0.f, dw_ = 0.f;
            const float col = acc[rt][r];
            if (okm != 0u) {   // uniform per 16-lane group
                const int org = __float_as_int(s0[0]);
                const int zd = (org >> 20) - 1, zh = ((org >> 10) & 1023) - 1, zw = (org & 1023) - 1;
                const float ld = s0[2], lh = s0[3], lw = s1x;
                const float fd[2] = {1.f - ld, ld}, fh[2] = {1.f - lh, lh}, fw[2] = {1.f - lw, lw};
                const long gbase = ((long)b * p.N + (long)(zd * p.H + zh) * p.W + zw) * p.C + ch_g;
                // window coordinates of corner 000
                const int wd0 = zd - (bd0 - HALO), wh0 = zh - (bh0 - HALO), ww0 = zw - (bw0 - HALO);
                float xv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int cd = (q >> 2) & 1, chh = (q >> 1) & 1, cw = q & 1;
                    xv[q] = ((okm >> q) & 1u) ? p.in[gbase + (long)(cd * HW + chh * p.W + cw) * p.C] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if ((okm >> q) & 1u) {
                        const int cd = (q >> 2) & 1, chh = (q >> 1) & 1, cw = q & 1;
                        dd_ = fmaf((cd ? 1.f : -1.f) * fh[chh] * fw[cw], xv[q], dd_);
                        dh_ = fmaf((chh ? 1.f : -1.f) * fd[cd] * fw[cw], xv[q], dh_);
                        dw_ = fmaf((cw ? 1.f : -1.f) * fd[cd] * fh[chh], xv[q], dw_);
                        if (p.gx) {
                            const float val = col * (fd[cd] * fh[chh] * fw[cw]);
                            const int wd_ = wd0 + cd, wh_ = wh0 + chh, ww_ = ww0 + cw;
                            if ((unsigned)wd_ < (unsigned)WD && (unsigned)wh_ < (unsigned)WH && (unsigned)ww_ < (unsigned)WW)
                                atomicAdd(&Win[((wd_ * WH + wh_) * WW + ww_) * CS + cj], val);
                            else
                                atomicAdd(p.gx + gbase + (long)(cd * HW + chh * p.W + cw) * p.C, val);
                        }
                    }
                }
            }
            pd_[rr] = col * dd_;
            ph_[rr] = col * dh_;
            pw_[rr] = col * dw_;
        }
        // ---- offset gradient: sum over the 16 channel lanes (LDS transpose), then over channel slices (atomics) ----
        if (p.goff) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int row = (rr >> 2) * 16 + kq * 4 + (rr & 3);
                Rd[wave][row * 3 + 0][cj] = pd_[rr];
                Rd[wave][row * 3 + 1][cj] = ph_[rr];
                Rd[wave][row * 3 + 2][cj] = pw_[rr];
            }
            __syncthreads();
            for (int q = lane; q < 96; q += 64) {
                float sum = 0.f;
#pragma unroll
                for (int e = 0; e < CS; ++e) sum += Rd[wave][q][e];
                const int row = q / 3, ax = q - row * 3;
                const int gd = bd0 + wave, gh = bh0 + (row >> 3), gw_ = bw0 + (row & 7);
                if (gd < p.D && gh < p.H && gw_ < p.W && sum != 0.f)
                    atomicAdd(p.goff + ((long)b * 3 * p.K + 3 * tap + ax) * p.N + (long)(gd * p.H + gh) * p.W + gw_, sum);
            }
        }
    }
    // ---- flush the window ----
    __syncthreads();
    if (p.gx) {
        for (int e = tid; e < WVOX * CS; e += 256) {
            const float val = Win[e];
            if (val != 0.f) {
                const int c = e % CS, vx = e / CS;
                const int ww_ = vx % WW, wh_ = (vx / WW) % WH, wd_ = vx / (WW * WH);
                const int zd = bd0 - HALO + wd_, zh = bh0 - HALO + wh_, zw = bw0 - HALO + ww_;
                // cells outside the volume are never written (corner validity), so val != 0 implies in-volume
                atomicAdd(p.gx + ((long)b * p.N + (long)(zd * p.H + zh) * p.W + zw) * p.C + cs * CS + c, val);
            }
        }
    }
}

int launch_cl_deform_bwd_lds(const DeformBwdArgs &a, hipStream_t st)
{
    if (a.C % CS) return DLKA_ERR_UNSUPPORTED;
    if (a.gx && launch_zero(a.gx, (size_t)a.B * a.N * a.C * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
    if (a.goff && launch_zero(a.goff, (size_t)a.B * 3 * a.K * a.N * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
    const int nbw = cdiv(a.W, BW), nbh = cdiv(a.H, BH), nbd = cdiv(a.D, BD);
    const int bricks = nbw * nbh * nbd * a.B, slices = a.C / CS;
    int groups = 1;
    if (bricks * slices < 256 && a.K % 3 == 0) groups = 3;
    if (bricks * slices * groups < 256 && a.K % 9 == 0) groups = 9;
    hipLaunchKernelGGL(cl_deform_bwd_lds_kernel, dim3(bricks, slices, groups), dim3(256), 0, st, a, nbw, nbh, nbd, cdiv(a.K, groups));
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

}  // namespace dlka
