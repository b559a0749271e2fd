// Channels-last forward of the 3-D deformable convolution (D3D semantics, groups = deformable_groups = 1) on the matrix
// cores, second generation:
//     out[m][n] = bias[n] + sum_tap sum_c  S(m, tap, c) * Wp[tap][c][n],     S = trilinear sample (deform_im2col_cuda.cuh:26-72)
// The reference writes S for all (tap, c) to a 27*C x B*N column buffer in HBM and multiplies it with at::addmm
// (deform_conv_cuda.cu:95-119).  Here S never leaves the chip: per (tap, 32-channel chunk) a wave gathers the 32 x 32
// sample tile in the line-friendly layout of cl_gather.h (lane = (row of 8, 16-byte piece of 8): every load instruction
// covers 8 whole 128-byte rows — 3.5x the gather rate of the first version's "lane = row" loads), transposes it through
// a wave-private LDS tile into the MFMA A layout, and feeds v_mfma_f32_32x32x2_f32.  The gather loads of unit u+1 are in
// flight under the MFMAs of unit u; the 4 waves of a workgroup share the 32 x NP weight chunk through LDS.
#include <stdlib.h>

#include "cl_args.h"
#include "cl_gather.h"
#include "dlka_kernels.h"

namespace dlka {

template <typename T, int NT>   // T: activation storage of `in` / `out` (float | bf16_t); split partial sums (gridDim.y > 1) always land in fp32
__global__ __launch_bounds__(256) void cl_deform_fwd_kernel(IgemmArgs p)
{
    constexpr unsigned SB = sizeof(T);
    constexpr int NPB = NT * 32;
    constexpr int BV = NT;
    constexpr int SROW = 36;   // padded sample-tile row (floats): 16-byte aligned, conflict-free b128 rows
    __shared__ __attribute__((aligned(16))) float Bs[2][32 * NPB];
    __shared__ __attribute__((aligned(16))) float Ssm[4][32 * SROW];
    __shared__ __attribute__((aligned(16))) float Dsm[4][32 * GATHER_DESC_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;      // MFMA roles
    using GG = GatherGeom<T>;
    const int gr = lane >> GG::PSHIFT, gp = lane & ((1 << GG::PSHIFT) - 1);   // gather roles: row gr of each group of RPI, 16-byte piece gp
    const int bx = DLKA_XCD_BX(p.xcd_nx);
    if (bx < 0) return;
    const int mbase = (bx * 4 + wave) * 32;
    const int m = mbase + i;
    const bool row_ok = m < p.M;
    const int b = row_ok ? m / p.N : 0;
    const int v = row_ok ? m - b * p.N : 0;
    const int w0 = v % p.W, h0 = (v / p.W) % p.H, d0 = v / (p.W * p.H);
    const int n0 = blockIdx.z * NPB;
    const int HW = p.H * p.W, rowbytes = p.Cin * SB;
    const BufRsrc rin = make_rsrc(p.in, (size_t)p.M * p.Cin * SB);
    float *S = Ssm[wave], *Dt = Dsm[wave];

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nchunk = p.CinP / 32;
    const int unit_lo = blockIdx.y * p.units_per_split;
    const int unit_hi = min(p.K * nchunk, unit_lo + p.units_per_split);

    f32x4 breg[BV];
    GatherPiece<T> xr[GG::NG][8];   // gathered corner pieces of the next unit, in flight
    RowDesc rd[GG::NG];             // descriptions of this lane's gather rows (current tap)
    int cur_tap = -1;

#define DLKA_LOAD_B(unit_)                                                                         \
    {                                                                                              \
        const int tap_ = (unit_) / nchunk, ck_ = (unit_) - tap_ * nchunk;                          \
        const float *src_ = p.wp + ((long)tap_ * p.CinP + ck_ * 32) * p.NP + n0;                   \
        _Pragma("unroll") for (int e = 0; e < BV; ++e) {                                           \
            const int idx_ = tid + e * 256;                                                        \
            const int rr_ = idx_ / (NPB / 4), c4_ = idx_ - rr_ * (NPB / 4);                        \
            breg[e] = reinterpret_cast<const f32x4 *>(src_ + (long)rr_ * p.NP)[c4_];              \
        }                                                                                          \
    }
    // The offsets of a tap are requested one tap AHEAD of their description (onx): loaded inside the description, every tap began with two
    // dependent L2 round trips — offsets, then the corner rows they address — of which the one-unit prefetch distance hides one at best.
    float onx[3] = {0.f, 0.f, 0.f};
    const int tap_first = unit_lo / nchunk, tap_last = (unit_hi - 1) / nchunk;
    auto load_offsets = [&](int tap) {
        if (h == 0 && row_ok && tap <= tap_last) {
            const float *op = p.off + ((long)b * 3 * p.K + 3 * tap) * p.N + v;
            onx[0] = op[0]; onx[1] = op[p.N]; onx[2] = op[2 * (long)p.N];
        }
    };
    if (unit_lo < unit_hi) load_offsets(tap_first);
    // describe (when the tap changes) and issue the 32 corner loads of one unit
    auto issue = [&](int unit) {
        const int tap = unit / nchunk, ck = unit - tap * nchunk;
        if (tap != cur_tap) {   // uniform
            cur_tap = tap;
            const int tk = tap % p.kw, tj = (tap / p.kw) % p.kh, ti = tap / (p.kw * p.kh);
            wave_sync();        // every lane has consumed the previous table
            if (h == 0) {
                RowDesc r;
                r.base = 0; r.okm = 0; r.ld = r.lh = r.lw = 0.f;
                if (row_ok)
                    r = gather_describe3(onx[0], onx[1], onx[2], p.N, b, d0 + ti * p.dd - p.pd, h0 + tj * p.dh - p.ph, w0 + tk * p.dw - p.pw, p.D, p.H, p.W);
                gather_publish(Dt, i, r);
            }
            load_offsets(tap + 1);   // in flight until the next tap is described
            wave_sync();
#pragma unroll
            for (int g = 0; g < GG::NG; ++g) rd[g] = gather_lookup(Dt, GG::RPI * g + gr);
        }
        const unsigned cbyte = (unsigned)(ck * 32 + GG::PE * gp) * SB;
#pragma unroll
        for (int g = 0; g < GG::NG; ++g)
#pragma unroll
            for (int q = 0; q < 8; ++q) xr[g][q] = gather_load<T>(rin, gather_offset(rd[g], q, HW, p.W, rowbytes, cbyte));
    };
    // interpolate, transpose through the wave-private tile, return the MFMA A values of this lane
    auto finish = [&](float *a) {
        wave_sync();   // previous tile consumed
#pragma unroll
        for (int g = 0; g < GG::NG; ++g) {
            float wq[8];
            gather_weights(rd[g], wq);
#pragma unroll
            for (int v = 0; v < GG::PE / 4; ++v) {
                f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x4 x4 = xr[g][q].get(v);
                    s4[0] = fmaf(wq[q], x4[0], s4[0]); s4[1] = fmaf(wq[q], x4[1], s4[1]);
                    s4[2] = fmaf(wq[q], x4[2], s4[2]); s4[3] = fmaf(wq[q], x4[3], s4[3]);
                }
                *reinterpret_cast<f32x4 *>(S + (GG::RPI * g + gr) * SROW + GG::PE * gp + 4 * v) = s4;
            }
        }
        wave_sync();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(S + i * SROW + 16 * h + 4 * e);
            a[4 * e] = t[0]; a[4 * e + 1] = t[1]; a[4 * e + 2] = t[2]; a[4 * e + 3] = t[3];
        }
    };

    if (unit_lo < unit_hi) {
        DLKA_LOAD_B(unit_lo)
        issue(unit_lo);
    }
    int buf = 0;
    for (int unit = unit_lo; unit < unit_hi; ++unit, buf ^= 1) {
        float a_cur[16];
#pragma unroll
        for (int e = 0; e < BV; ++e) reinterpret_cast<f32x4 *>(Bs[buf])[tid + e * 256] = breg[e];
        finish(a_cur);      // consumes xr (the loads issued one iteration ago)
        __syncthreads();    // Bs[buf] staged; Bs[buf^1] (read two iterations ago) is free again
        if (unit + 1 < unit_hi) {
            DLKA_LOAD_B(unit + 1)
            issue(unit + 1);
        }
        const float *brow = Bs[buf] + (16 * h) * NPB + i;
#pragma unroll
        for (int st = 0; st < 16; ++st) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma_32x32x2(a_cur[st], brow[st * NPB + t * 32], acc[t]);
        }
    }
#undef DLKA_LOAD_B

    // ---- epilogue: D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
    const bool split = gridDim.y > 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + t * 32 + i;
        if (n >= p.Cout) continue;
        const float bv = (p.bias && blockIdx.y == 0) ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = mbase + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mr >= p.M) continue;
            const float val = acc[t][r] + bv;
            if (split) atomicAdd(p.out + (long)mr * p.Cout + n, val);   // fp32 accumulation buffer (the launcher's caller casts it for bf16)
            else act_store1(reinterpret_cast<T *>(p.out), (long)mr * p.Cout + n, val);
        }
    }
}

int launch_cl_deform_fwd(IgemmArgs a, int splits, hipStream_t st)
{
    if ((long)a.M * a.Cin * 4 >= (1l << 31)) return DLKA_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    // act_bf16 with splits > 1: a.out must be an fp32 [M][Cout] accumulation buffer (the caller converts it afterwards)
    a.units_per_split = cdiv(a.K * (a.CinP / 32), splits);
    splits = cdiv(a.K * (a.CinP / 32), a.units_per_split);
    if (splits > 1 && !a.out_zeroed) {
        if (launch_zero(a.out, (size_t)a.M * a.Cout * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
    }
    const int NT_total = a.NP / 32;
    const int mblocks = cdiv(a.M, 128);
    // all column tiles in one workgroup (the gather is not repeated) up to 4; wider outputs split over gridDim.z
    int NT = NT_total;
    if (NT_total == 8) NT = 4;
    else if (NT_total == 3 || NT_total > 4) return DLKA_ERR_UNSUPPORTED;
    // small volumes: fewer column tiles per workgroup (-> more workgroups) beats not repeating the gather
    // (C=256 / 4^3: 57.8 -> 36.0 us with NT = 1; C=128 / 8^3: 47.6 -> 41.0 us with NT = 2; profiles/r01s_dfwd_tuning.txt)
    if (a.M <= 256) NT = 1;
    else if (a.M <= 2048 && NT_total == 4) NT = 2;
    constexpr int nt_env = 0;
    if (nt_env == 1 || nt_env == 2 || nt_env == 4) { if (NT_total % nt_env == 0 && nt_env <= NT_total) NT = nt_env; }
    dim3 grid(mblocks, splits, NT_total / NT), block(256);
    a.xcd_nx = 0;
    if (xcd_swizzle_enabled() && mblocks >= (unsigned)xcd_min_blocks()) { a.xcd_nx = mblocks; grid.x = xcd_grid(mblocks); }
    if (a.act_bf16) {
        switch (NT) {
            case 1: { auto k = cl_deform_fwd_kernel<bf16_t, 1>; DLKA_LAUNCH(k, grid, block, 0, st, a); } break;
            case 2: { auto k = cl_deform_fwd_kernel<bf16_t, 2>; DLKA_LAUNCH(k, grid, block, 0, st, a); } break;
            case 4: { auto k = cl_deform_fwd_kernel<bf16_t, 4>; DLKA_LAUNCH(k, grid, block, 0, st, a); } break;
            default: return DLKA_ERR_UNSUPPORTED;
        }
    } else {
        switch (NT) {
            case 1: { auto k = cl_deform_fwd_kernel<float, 1>; DLKA_LAUNCH(k, grid, block, 0, st, a); } break;
            case 2: { auto k = cl_deform_fwd_kernel<float, 2>; DLKA_LAUNCH(k, grid, block, 0, st, a); } break;
            case 4: { auto k = cl_deform_fwd_kernel<float, 4>; DLKA_LAUNCH(k, grid, block, 0, st, a); } break;
            default: return DLKA_ERR_UNSUPPORTED;
        }
    }
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

}  // namespace dlka
