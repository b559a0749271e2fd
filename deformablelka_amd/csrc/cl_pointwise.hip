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
    if (p.zero_xblocks && (int)blockIdx.x >= (int)gridDim.x - p.zero_xblocks) {   // riding zero fills (IgemmArgs::zero): the workgroups behind the row tiles
        zero_batch_block(p.zero, (blockIdx.x - (gridDim.x - p.zero_xblocks)) * gridDim.y + blockIdx.y, lane, 64);
        return;
    }
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
    // bf16 storage: 2-byte accesses were the slow part of this kernel (10.5 vs 7.7 us against fp32 at C=64 / 16^3).  Rows come in pairs
    // (r, r + 1) = (R, R + 1); the even lane of a column pair handles row R, the odd lane row R + 1, each with ONE dword holding both columns
    // of the pair, and the halves that belong to the neighbour cross over with one DPP move.
    constexpr bool B16 = sizeof(T) == 2;
    const bool odd = i & 1;
    // (the epilogue operands are only REQUESTED here — raw words; exchanged and converted in the epilogue, so that nothing waits on them now)
    unsigned auxw[8], aux2w[8];
    auto pair_request = [&](const T *src, int r) -> unsigned {   // the dword (src[Rl][n & ~1], src[Rl][n | 1]) of this lane's row Rl = R + odd
        const int mrow = mbase + (r & 3) + 8 * (r >> 2) + 4 * h + (odd ? 1 : 0);
        return (n_ok && mrow < p.M) ? *reinterpret_cast<const unsigned *>(src + ((long)mrow * p.Cout + (n & ~1))) : 0u;
    };
    auto pair_finish = [&](unsigned own, float &x0, float &x1) {   // -> src[R][n], src[R + 1][n]
        const unsigned oth = lane_xor1(own);
        x0 = __uint_as_float(odd ? (oth & 0xffff0000u) : (own << 16));
        x1 = __uint_as_float(odd ? (own & 0xffff0000u) : (oth << 16));
    };
    if (p.epi >= 2) {   // uniform
        if (B16) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                auxw[r >> 1] = pair_request(auxp, r);
                aux2w[r >> 1] = p.epi == 4 ? pair_request(aux2p, r) : 0u;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = mbase + (r & 3) + 8 * (r >> 2) + 4 * h;
                const bool ok = n_ok && mr < p.M;
                auxv[r] = ok ? act_load1(auxp, (long)mr * p.Cout + n) : 0.f;
                aux2v[r] = (ok && p.epi == 4) ? act_load1(aux2p, (long)mr * p.Cout + n) : 0.f;
            }
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
            if (B16) {   // 16 channels = 32 bytes: two 16-byte loads
                buf_load_bf16x8(rin, ao, a[u][0], a[u][1]);
                buf_load_bf16x8(rin, ao == DLKA_OOB ? DLKA_OOB : ao + 16u, a[u][2], a[u][3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) a[u][e] = act_buf_load4<T>(rin, ao + 4u * SB * e);
            }
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
    if (n >= p.Cout) return;   // (whole waves: Cout % 32 == 0 on the bf16 path, so the lane exchanges below stay complete)
    const float bv = p.bias ? p.bias[n] : 0.f;
    if (B16) {
        auto pair_store = [&](T *dst, int r, float x0, float x1) {   // dst[R][n] = x0, dst[R + 1][n] = x1 (bf16, round to nearest even)
            const unsigned give = bf16_bits(odd ? x0 : x1), mine = bf16_bits(odd ? x1 : x0);
            const unsigned got = lane_xor1(give);
            const unsigned word = odd ? (got | (mine << 16)) : (mine | (got << 16));
            const int mrow = mbase + (r & 3) + 8 * (r >> 2) + 4 * h + (odd ? 1 : 0);
            if (mrow < p.M) *reinterpret_cast<unsigned *>(dst + ((long)mrow * p.Cout + (n & ~1))) = word;
        };
        if (p.epi >= 2) {   // uniform
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                pair_finish(auxw[r >> 1], auxv[r], auxv[r + 1]);
                pair_finish(aux2w[r >> 1], aux2v[r], aux2v[r + 1]);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float v0 = acc[r] + bv, v1 = acc[r + 1] + bv;
            if (p.epi == 0) {
                pair_store(outp, r, v0, v1);
            } else if (p.epi == 1) {
                pair_store(outp, r, v0, v1);
                pair_store(out2p, r, gelu_f(v0), gelu_f(v1));
            } else if (p.epi == 2) {
                pair_store(outp, r, v0, v1);
                pair_store(out2p, r, auxv[r] * v0, auxv[r + 1] * v1);
            } else if (p.epi == 3) {
                pair_store(outp, r, v0 + auxv[r], v1 + auxv[r + 1]);
            } else {
                pair_store(outp, r, v0 * auxv[r], v1 * auxv[r + 1]);
                pair_store(out2p, r, v0 * aux2v[r], v1 * aux2v[r + 1]);
            }
        }
        return;
    }
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
    IgemmArgs ax = a;
    ax.zero_xblocks = 0;
    if (a.zero.n > 0) {   // zero fills riding in this launch: 4096 floats per extra workgroup
        if (a.zero.overflow) return DLKA_ERR_WORKSPACE;
        unsigned blk = 0;
        for (int r = 0; r < a.zero.n; ++r) {
            if ((uintptr_t)a.zero.p[r] & 15) return DLKA_ERR_UNSUPPORTED;
            ax.zero.block0[r] = blk;
            blk += (unsigned)cdivl(a.zero.cnt[r], 4096);
        }
        ax.zero.block0[a.zero.n] = blk;
        ax.zero_xblocks = (int)cdiv((int)blk, (int)grid.y);
        grid.x += ax.zero_xblocks;
    }
    if (a.act_bf16) { auto k = cl_pointwise_kernel<bf16_t>; hipLaunchKernelGGL(k, grid, block, 0, st, ax); }
    else { auto k = cl_pointwise_kernel<float>; hipLaunchKernelGGL(k, grid, block, 0, st, ax); }
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

}  // namespace dlka
