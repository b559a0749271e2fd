// Channels-last backward of the 3-D deformable convolution w.r.t. input and offsets, fused.
//
// Replaces three reference passes (3D/dcn/src/cuda/deform_conv_cuda.cu:226-251):
//   columns = W^T * grad_out  (at::mm into a 27*C x B*N buffer)          -> here a 32-voxel x 32-channel tile per MFMA chain,
//   deformable_col2im_coord_gpu_kernel (grad_offset, deform_im2col_cuda.cuh:336-405)   kept in registers,
//   deformable_col2im_gpu_kernel       (grad_input,  cuh:267-334, fp32 atomics)         never written to HBM.
//
// Per wave: 32 voxels (MFMA rows).  Col[v][c] = sum_co G[v][co] * W[co][c][tap] lands in the MFMA D layout with
// lane = channel, registers = 16 voxels.  For each of its 16 voxels a lane then
//   * reads the 8 corner values x[corner][c]           (a half-wave reads one contiguous 128-byte row piece),
//   * scatters Col*w_corner with fp32 atomics          (a half-wave hits one contiguous 128-byte piece -> few L2 ops),
//   * accumulates Col * d(sample)/d(q_axis) for the offset gradient.
// The per-voxel sampling description (corner base, validity mask, fractions) is computed once per (voxel, tap) by
// lane = voxel and broadcast through LDS; the channel reduction of the offset gradient is an LDS transpose-reduce.
#include "deform_sample.h"
#include "cl_args.h"
#include "dlka_kernels.h"

namespace dlka {

__global__ __launch_bounds__(256) void cl_deform_bwd_kernel(DeformBwdArgs p)
{
    __shared__ __attribute__((aligned(16))) float Bs[32 * 32];          // [co chunk][ci chunk]
    __shared__ __attribute__((aligned(16))) float Sx[4][32][8];         // per wave, per row: base, mask, ld, lh, lw, batch
    __shared__ float Rd[4][96][33];                                     // per wave transpose-reduce buffer
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int mbase = (blockIdx.x * 4 + wave) * 32;
    const int m = mbase + i;
    const bool row_ok = m < p.M;
    const int b = row_ok ? m / p.N : 0;
    const int v = row_ok ? m - b * p.N : 0;
    const int w0 = v % p.W, h0 = (v / p.W) % p.H, d0 = v / (p.W * p.H);
    const int HW = p.H * p.W;
    const int ncc = p.C / 32, nkc = p.CoutP / 32;

    const int cc_lo = blockIdx.z * p.cc_per_block, cc_hi = min(ncc, cc_lo + p.cc_per_block);
    {
        const int tap = blockIdx.y;
        const int tk = tap % p.kw, tj = (tap / p.kw) % p.kh, ti = tap / (p.kw * p.kh);
        // ---- sampling description of row i (lanes of half 0 publish it) ----
        {
            int base = 0;
            unsigned okm = 0;
            float ld = 0.f, lh = 0.f, lw = 0.f;
            if (row_ok) {
                const float *offp = p.off + ((long)b * 3 * p.K + 3 * tap) * p.N + v;
                const float qd = (float)(d0 + ti * p.dd - p.pd) + offp[0];
                const float qh = (float)(h0 + tj * p.dh - p.ph) + offp[p.N];
                const float qw = (float)(w0 + tk * p.dw - p.pw) + offp[2 * (long)p.N];
                const bool inside = qd > -1.f && qh > -1.f && qw > -1.f && qd < (float)p.D && qh < (float)p.H && qw < (float)p.W;
                if (inside) {  // floor in [-1, size-1]
                    const float fd_ = floorf(qd), fh_ = floorf(qh), fw_ = floorf(qw);
                    const int zd = (int)fd_, zh = (int)fh_, zw = (int)fw_;
                    ld = qd - fd_; lh = qh - fh_; lw = qw - fw_;
                    base = (zd * p.H + zh) * p.W + zw;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int cd = (q >> 2) & 1, ch = (q >> 1) & 1, cw = q & 1;
                        const bool ok = (cd ? zd + 1 <= p.D - 1 : zd >= 0) && (ch ? zh + 1 <= p.H - 1 : zh >= 0) &&
                                        (cw ? zw + 1 <= p.W - 1 : zw >= 0);
                        okm |= (ok ? 1u : 0u) << q;
                    }
                }
            }
            if (h == 0) {
                float *sx = &Sx[wave][i][0];
                sx[0] = __int_as_float(base);
                sx[1] = __int_as_float((int)okm);
                sx[2] = ld; sx[3] = lh; sx[4] = lw;
                sx[5] = __int_as_float(b);
            }
        }
        float pd_[16], ph_[16], pw_[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { pd_[r] = 0.f; ph_[r] = 0.f; pw_[r] = 0.f; }

        for (int cc = cc_lo; cc < cc_hi; ++cc) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            for (int kc = 0; kc < nkc; ++kc) {
                __syncthreads();  // Bs consumed (also orders the Sx publication before its first read)
                {
                    // 32 rows (co) x 32 floats (ci): 256 float4, one per thread
                    const int rr = tid >> 3, c4 = tid & 7;
                    const float4 *src = reinterpret_cast<const float4 *>(p.wp + ((long)tap * p.CoutP + kc * 32 + rr) * p.C + cc * 32) + c4;
                    reinterpret_cast<float4 *>(Bs)[rr * 8 + c4] = *src;
                }
                float a[16];
                if (row_ok && kc * 32 + 16 * h < p.Cout) {
                    const float4 *g4 = reinterpret_cast<const float4 *>(p.g + (long)m * p.Cout + kc * 32 + 16 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float4 t = g4[e];
                        a[4 * e] = t.x; a[4 * e + 1] = t.y; a[4 * e + 2] = t.z; a[4 * e + 3] = t.w;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) a[e] = 0.f;
                }
                __syncthreads();
                const float *brow = Bs + (16 * h) * 32 + i;
#pragma unroll
                for (int st = 0; st < 16; ++st) acc = mfma_32x32x2(a[st], brow[st * 32], acc);
            }
            // ---- acc[r] = Col[row(r,h)][ci = cc*32 + i] ----
            const int ci = cc * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (mbase + row >= p.M) continue;
                const float4 s0 = *reinterpret_cast<const float4 *>(&Sx[wave][row][0]);
                const float4 s1 = *reinterpret_cast<const float4 *>(&Sx[wave][row][4]);
                const unsigned okm = (unsigned)__float_as_int(s0.y);
                if (okm == 0u) continue;  // uniform per half-wave: nothing sampled for this (voxel, tap)
                const int base = __float_as_int(s0.x);
                const int rb = __float_as_int(s1.y);
                const float ld = s0.z, lh = s0.w, lw = s1.x;
                const float fd[2] = {1.f - ld, ld}, fh[2] = {1.f - lh, lh}, fw[2] = {1.f - lw, lw};
                const float col = acc[r];
                const long rowoff = ((long)rb * p.N) * p.C + ci;
                float dd_ = 0.f, dh_ = 0.f, dw_ = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if ((okm >> q) & 1u) {
                        const int cd = (q >> 2) & 1, ch = (q >> 1) & 1, cw = q & 1;
                        const long o = rowoff + (long)(base + cd * HW + ch * p.W + cw) * p.C;
                        const float xv = p.in[o];
                        dd_ = fmaf((cd ? 1.f : -1.f) * fh[ch] * fw[cw], xv, dd_);
                        dh_ = fmaf((ch ? 1.f : -1.f) * fd[cd] * fw[cw], xv, dh_);
                        dw_ = fmaf((cw ? 1.f : -1.f) * fd[cd] * fh[ch], xv, dw_);
                        if (p.gx) atomicAdd(p.gx + o, col * (fd[cd] * fh[ch] * fw[cw]));
                    }
                }
                pd_[r] = fmaf(col, dd_, pd_[r]);
                ph_[r] = fmaf(col, dh_, ph_[r]);
                pw_[r] = fmaf(col, dw_, pw_[r]);
            }
        }
        // ---- offset gradient: sum the partials over the 32 channel-lanes of each half ----
        if (p.goff) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                Rd[wave][row * 3 + 0][i] = pd_[r];
                Rd[wave][row * 3 + 1][i] = ph_[r];
                Rd[wave][row * 3 + 2][i] = pw_[r];
            }
            __syncthreads();
            for (int q = lane; q < 96; q += 64) {
                float sum = 0.f;
#pragma unroll
                for (int e = 0; e < 32; ++e) sum += Rd[wave][q][e];
                const int row = q / 3, ax = q - row * 3;
                const int mr = mbase + row;
                if (mr < p.M) {
                    const int bb = mr / p.N, vv = mr - bb * p.N;
                    float *dst = p.goff + ((long)bb * 3 * p.K + 3 * tap + ax) * p.N + vv;
                    if (gridDim.z > 1) atomicAdd(dst, sum); else *dst = sum;
                }
            }
        }
    }
}

int launch_cl_deform_bwd(const DeformBwdArgs &a, hipStream_t st)
{
    if (a.gx) {
        if (launch_zero(a.gx, (size_t)a.B * a.N * a.C * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
    }
    DeformBwdArgs b = a;
    const int ncc = a.C / 32, mblocks = cdiv(a.M, 128);
    int zsplit = 1;
    while (mblocks * a.K * zsplit < 512 && zsplit < ncc) ++zsplit;
    b.cc_per_block = cdiv(ncc, zsplit);
    zsplit = cdiv(ncc, b.cc_per_block);
    if (zsplit > 1 && a.goff) {
        if (launch_zero(a.goff, (size_t)a.B * 3 * a.K * a.N * 4, st) != DLKA_OK) return DLKA_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(cl_deform_bwd_kernel, dim3(mblocks, a.K, zsplit), dim3(256), 0, st, b);
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

}  // namespace dlka
