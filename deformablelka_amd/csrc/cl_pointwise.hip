// Pointwise (1x1x1) convolutions on the token tensor, wave-granular:  out[m][n] = sum_k x[m][k] * Wp[k][n]  (+ epilogue).
// (proj_1 / conv1 / proj_2 of the D-LKA block, transformerblock.py:641,659,662, and their data gradients.)
//
// The general implicit-GEMM kernel (cl_igemm.hip) tiles 128 rows x all columns per workgroup and walks the C/32 channel chunks
// one after the other with a one-deep prefetch: right for 27-tap convs, wrong for K = 1, where the whole contraction is 1..8 chunks.
// At the small stages that left a handful of workgroups on the chip, each paying one global-load latency per chunk in sequence
// (C = 256 / 4^3: 21.7 us for a 128 x 256 x 256 GEMM; C = 64 / 16^3: 64 workgroups on 256 CUs; profiles/r01n).  Here one wave
// owns a 32 x 32 output tile, issues the loads of up to four chunks back to back (A rows straight from the token tensor, B from the
// L2-resident prepared weights, both already in MFMA operand order), and runs its MFMAs when they land: grid = (M/32) x (C/32) waves.
#include "cl_args.h"
#include "dlka_kernels.h"

namespace dlka {

template <typename T>   // activation storage: float, or bf16_t (fp32 arithmetic either way; weights / bias are fp32)
__global__ __launch_bounds__(64, 2) void cl_pointwise_kernel(IgemmArgs p)
{
    constexpr unsigned SB = sizeof(T);
    const T *auxp = reinterpret_cast<const T *>(p.aux), *aux2p = reinterpret_cast<const T *>(p.aux2);
    T *outp = reinterpret_cast<T *>(p.out), *out2p = reinterpret_cast<T *>(p.out2);
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int mbase = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int m = mbase + i, n = n0 + i;
    const BufRsrc rin = make_rsrc(p.in, (size_t)p.M * p.Cin * SB);
    const BufRsrc rw = make_rsrc(p.wp, (size_t)p.CinP * p.NP * 4);
    // MFMA k order within a 32-channel chunk: step s contracts channels s (lanes 0-31) and 16 + s (lanes 32-63) — the same on both operands
    const unsigned abase = m < p.M ? (unsigned)m * (unsigned)p.Cin * SB + 16u * SB * h : DLKA_OOB;
    const unsigned bbase = ((unsigned)(16 * h) * (unsigned)p.NP + (unsigned)n) * 4u;
    const unsigned bstep = (unsigned)p.NP * 4u;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // epilogue operands (residual / gate tensors) are requested FIRST: their addresses are known now, and a wave's timeline is otherwise
    // load latency -> MFMAs -> a second load latency for them
    float auxv[16], aux2v[16];
    const bool n_ok = n < p.Cout;
    if (p.epi >= 2) {   // uniform
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = mbase + (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool ok = n_ok && mr < p.M;
            auxv[r] = ok ? act_load1(auxp, (long)mr * p.Cout + n) : 0.f;
            aux2v[r] = (ok && p.epi == 4) ? act_load1(aux2p, (long)mr * p.Cout + n) : 0.f;
        }
    }
    const int nchunk = p.CinP / 32;
    for (int c0 = 0; c0 < nchunk; c0 += 4) {
        f32x4 a[4][4];
        float b[4][16];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c0 + u >= nchunk) break;   // uniform
            const unsigned ao = abase + (unsigned)(c0 + u) * 32u * SB;
            const unsigned bo = bbase + (unsigned)(c0 + u) * 32u * bstep;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[u][e] = act_buf_load4<T>(rin, ao + 4u * SB * e);
#pragma unroll
            for (int s = 0; s < 16; ++s) b[u][s] = buf_load_f32(rw, bo + (unsigned)s * bstep);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c0 + u >= nchunk) break;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = mfma_32x32x2(a[u][s >> 2][s & 3], b[u][s], acc);
        }
    }
    // ---- epilogue: D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); same menu as cl_igemm_kernel ----
    if (n >= p.Cout) return;
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mr = mbase + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (mr >= p.M) continue;
        const float val = acc[r] + bv;
        const long o = (long)mr * p.Cout + n;
        if (p.epi == 0) {
            act_store1(outp, o, val);
        } else if (p.epi == 1) {
            act_store1(outp, o, val);
            act_store1(out2p, o, gelu_f(val));
        } else if (p.epi == 2) {
            act_store1(outp, o, val);
            act_store1(out2p, o, auxv[r] * val);
        } else if (p.epi == 3) {
            act_store1(outp, o, val + auxv[r]);
        } else {
            act_store1(outp, o, val * auxv[r]);
            act_store1(out2p, o, val * aux2v[r]);
        }
    }
}

// K = 1, channels-last in and out, unsplit, exact fp32.  Returns DLKA_ERR_UNSUPPORTED for anything else.
int launch_cl_pointwise(const IgemmArgs &a, hipStream_t st)
{
    if (a.K != 1 || a.split_bf16 || a.Cin % 32 || a.CinP != a.Cin || a.NP % 32) return DLKA_ERR_UNSUPPORTED;
    if ((long)a.M * a.Cin * 4 >= (1l << 31) || (long)a.CinP * a.NP * 4 >= (1l << 31)) return DLKA_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    dim3 grid(cdiv(a.M, 32), a.NP / 32), block(64);
    if (a.act_bf16) { auto k = cl_pointwise_kernel<bf16_t>; hipLaunchKernelGGL(k, grid, block, 0, st, a); }
    else { auto k = cl_pointwise_kernel<float>; hipLaunchKernelGGL(k, grid, block, 0, st, a); }
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

}  // namespace dlka
