// Pointwise (1x1x1) convolutions on the token tensor, wave-granular:  out[m][n] = sum_k x[m][k] * Wp[k][n]  (+ epilogue).
// (proj_1 / conv1 / proj_2 of the D-LKA block, transformerblock.py:641,659,662, and their data gradients.)
//
// The general implicit-GEMM kernel (cl_igemm.hip) tiles 128 rows x all columns per workgroup and walks the C/32 channel chunks
// one after the other with a one-deep prefetch: right for 27-tap convs, wrong for K = 1, where the whole contraction is 1..8 chunks.
// At the small stages that left a handful of workgroups on the chip, each paying one global-load latency per chunk in sequence
// (C = 256 / 4^3: 21.7 us for a 128 x 256 x 256 GEMM; C = 64 / 16^3: 64 workgroups on 256 CUs; profiles/archive/r01n).  Here one wave
// owns a 32 x 32 output tile, issues the loads of up to four chunks back to back (A rows straight from the token tensor, B from the
// L2-resident prepared weights, both already in MFMA operand order), and runs its MFMAs when they land: grid = (M/32) x (C/32) waves.
#include "cl_args.h"
#include "dlka_kernels.h"

namespace dlka {

// KW > 1 (small M, wide C — the C = 128 / 256 stages): KW waves share an output tile, wave w contracts the channel chunks w, w + KW, ..., the partial tiles
// meet in LDS and wave 0 runs the epilogue.  With one wave per tile a 128 x 256 x 256 product is 32 waves of two load round trips and 128 dependent
// MFMAs (11.4 us, of which 3.4 us is the MFMA chain); four waves: one round trip and 32 MFMAs each.  No atomics, no zero fill, same epilogues; the sum
// of the chunk products is formed in a different order (rounding only).
template <typename T, int KW = 1>   // T: activation storage: float, or bf16_t (fp32 arithmetic either way; weights / bias are fp32)
__global__ __launch_bounds__(64 * KW, 2) void cl_pointwise_kernel(IgemmArgs p)
{
    constexpr unsigned SB = sizeof(T);
    const T *auxp = reinterpret_cast<const T *>(p.aux), *aux2p = reinterpret_cast<const T *>(p.aux2);
    T *outp = reinterpret_cast<T *>(p.out), *out2p = reinterpret_cast<T *>(p.out2);
    const int lane = threadIdx.x & 63, kwave = KW > 1 ? (int)(threadIdx.x >> 6) : 0, i = lane & 31, h = lane >> 5;
    if (p.zero_xblocks && (int)blockIdx.x >= (int)gridDim.x - p.zero_xblocks) {   // riding zero fills (IgemmArgs::zero): the workgroups behind the row tiles
        if (kwave == 0) zero_batch_block(p.zero, (blockIdx.x - (gridDim.x - p.zero_xblocks)) * gridDim.y + blockIdx.y, lane, 64);
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
    if (p.epi >= 2 && kwave == 0) {   // uniform
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
    for (int c0 = 0; c0 * KW + kwave < nchunk; c0 += 4) {   // this wave's chunks: kwave, kwave + KW, ...
        f32x4 a[4][4];
        float b[4][16];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ck = (c0 + u) * KW + kwave;
            if (ck >= nchunk) break;   // uniform per wave
            const unsigned ao = abase + (unsigned)ck * 32u * SB;
            const unsigned bo = bbase + (unsigned)ck * 32u * bstep;
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
            if ((c0 + u) * KW + kwave >= nchunk) break;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = mfma_32x32x2(a[u][s >> 2][s & 3], b[u][s], acc);
        }
    }
    if (KW > 1) {   // fold the KW partial tiles: waves 1 .. KW-1 hand theirs over through LDS, wave 0 goes on to the epilogue
        __shared__ __attribute__((aligned(16))) float red[(KW > 1 ? KW - 1 : 1) * 16 * 64];
        if (kwave > 0) {
#pragma unroll
            for (int r = 0; r < 16; r += 4)
                *reinterpret_cast<f32x4 *>(&red[((kwave - 1) * 16 + r) * 64 + 4 * lane]) = f32x4{acc[r], acc[r + 1], acc[r + 2], acc[r + 3]};
        }
        __syncthreads();
        if (kwave > 0) return;
#pragma unroll
        for (int w = 0; w < KW - 1; ++w)
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const f32x4 o = *reinterpret_cast<const f32x4 *>(&red[(w * 16 + r) * 64 + 4 * lane]);
                acc[r] += o[0]; acc[r + 1] += o[1]; acc[r + 2] += o[2]; acc[r + 3] += o[3];
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
                const float g0 = gelu_f(v0), g1 = gelu_f(v1);
                pair_store(outp, r, v0, v1);
                pair_store(out2p, r, g0, g1);
                if (p.out2_f32) {   // uniform: the unrounded a = GELU(h) for the fp32 offset-determining chain (lanes 0..31 of a half cover one 128-byte row piece)
                    const int mr0 = mbase + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (mr0 < p.M) p.out2_f32[(long)mr0 * p.Cout + n] = g0;
                    if (mr0 + 1 < p.M) p.out2_f32[(long)(mr0 + 1) * p.Cout + n] = g1;
                }
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

// ---------------------------------------------------------------------------------------------------------------------
// Two dependent pointwise convs in ONE launch (PwPairArgs): conv1 + gate -> proj_2 + shortcut in the forward pass, proj_2^T + gate backward ->
// conv1^T in the backward pass.  Each of these kernels is ~8-10 us of load latency -> 16 MFMAs -> store latency, plus ~4.5 us of graph
// dispatch; a wave that owns ALL C <= 64 columns of its 32 rows can feed the second contraction from its own first result: the intermediate
// tile goes through a wave-private LDS tile (D layout in, A-operand rows out), nothing else changes — same MFMA order, same epilogue
// arithmetic, same rounding of the stored intermediate (bf16 storage: the second conv consumes the ROUNDED values, as it does unfused).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int NTC>   // NTC = C / 32 column tiles = 32-channel chunks
__global__ __launch_bounds__(64, 2) void cl_pointwise_pair_kernel(PwPairArgs p)
{
    constexpr unsigned SB = sizeof(T);
    constexpr bool B16 = SB == 2;
    constexpr int C = NTC * 32, LD = C + 4;
    __shared__ __attribute__((aligned(16))) float Tm[32 * LD];
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int mtiles = (int)gridDim.x - p.zero_blocks;
    if ((int)blockIdx.x >= mtiles) {   // riding zero fills
        zero_batch_block(p.zero, blockIdx.x - mtiles, lane, 64);
        return;
    }
    const T *ap = reinterpret_cast<const T *>(p.a), *bp = reinterpret_cast<const T *>(p.b);
    T *o1 = reinterpret_cast<T *>(p.out1), *o1b = reinterpret_cast<T *>(p.out1b), *o2 = reinterpret_cast<T *>(p.out2);
    const int mbase = blockIdx.x * 32, m = mbase + i;
    const bool odd = i & 1;
    const BufRsrc rin = make_rsrc(p.in, (size_t)p.M * C * SB);
    const BufRsrc rw1 = make_rsrc(p.wp1, (size_t)C * C * 4), rw2 = make_rsrc(p.wp2, (size_t)C * C * 4);
    const unsigned abase = m < p.M ? (unsigned)m * (unsigned)C * SB + 16u * SB * h : DLKA_OOB;
    // element (row R(r), column n) of a channels-last tensor, D layout: R(r) = mbase + (r & 3) + 8 * (r >> 2) + 4 * h
    auto row_of = [&](int r) { return mbase + (r & 3) + 8 * (r >> 2) + 4 * h; };
    auto load_tile = [&](const T *src, int nt, float *v) {   // v[r] = src[R(r)][nt * 32 + i]
        const int n = nt * 32 + i;
        if (B16) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int mrow = row_of(r) + (odd ? 1 : 0);
                const unsigned own = mrow < p.M ? *reinterpret_cast<const unsigned *>(src + ((long)mrow * C + (n & ~1))) : 0u;
                const unsigned oth = lane_xor1(own);
                v[r] = __uint_as_float(odd ? (oth & 0xffff0000u) : (own << 16));
                v[r + 1] = __uint_as_float(odd ? (own & 0xffff0000u) : (oth << 16));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = row_of(r) < p.M ? act_load1(src, (long)row_of(r) * C + n) : 0.f;
        }
    };
    auto store_tile = [&](T *dst, int nt, const float *v) {   // dst[R(r)][nt * 32 + i] = v[r]
        const int n = nt * 32 + i;
        if (B16) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const unsigned give = bf16_bits(odd ? v[r] : v[r + 1]), mine = bf16_bits(odd ? v[r + 1] : v[r]);
                const unsigned got = lane_xor1(give);
                const unsigned word = odd ? (got | (mine << 16)) : (mine | (got << 16));
                const int mrow = row_of(r) + (odd ? 1 : 0);
                if (mrow < p.M) *reinterpret_cast<unsigned *>(dst + ((long)mrow * C + (n & ~1))) = word;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (row_of(r) < p.M) act_store1(dst, (long)row_of(r) * C + n, v[r]);
        }
    };
    auto stored = [&](float x) { return B16 ? bf16_value(bf16_bits(x)) : x; };   // the value a tensor of type T holds after x was stored

    // ---- first conv: A rows from global memory ----
    f32x4 a[NTC][4];
#pragma unroll
    for (int u = 0; u < NTC; ++u) {
        const unsigned ao = abase == DLKA_OOB ? DLKA_OOB : abase + (unsigned)u * 32u * SB;
        if (B16) {
            buf_load_bf16x8(rin, ao, a[u][0], a[u][1]);
            buf_load_bf16x8(rin, ao == DLKA_OOB ? DLKA_OOB : ao + 16u, a[u][2], a[u][3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[u][e] = act_buf_load4<T>(rin, ao == DLKA_OOB ? DLKA_OOB : ao + 16u * e);
        }
    }
#pragma unroll 1
    for (int nt = 0; nt < NTC; ++nt) {   // (not unrolled: two column tiles in flight at once spill)
        const int n = nt * 32 + i;
        float av[16], bv2[16];
        load_tile(ap, nt, av);                       // gate operand a
        if (p.bwd) load_tile(bp, nt, bv2);           // backward: the second gate operand (g1)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int u = 0; u < NTC; ++u) {
            float b[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) b[s] = buf_load_f32(rw1, (unsigned)((u * 32 + 16 * h + s) * C + n) * 4u);
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = mfma_32x32x2(a[u][s >> 2][s & 3], b[s], acc);
        }
        const float bias = (!p.bwd && p.bias1) ? p.bias1[n] : 0.f;
        float t1[16], t2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float t = acc[r] + bias;
            if (p.bwd) { t1[r] = t * av[r]; t2[r] = t * bv2[r]; }   // gg1 = gm * a, ga1 = gm * g1
            else { t1[r] = t; t2[r] = av[r] * t; }                     // g1, m = a * g1
        }
        store_tile(o1, nt, t1);
        store_tile(o1b, nt, t2);
        // operand of the second conv (backward: out1, forward: out1b) -> LDS tile, rows = voxels
#pragma unroll
        for (int r = 0; r < 16; ++r) Tm[((r & 3) + 8 * (r >> 2) + 4 * h) * LD + n] = stored(p.bwd ? t1[r] : t2[r]);
    }
    wave_sync();
    // ---- second conv: A rows from the LDS tile ----
    f32x4 a2[NTC][4];
#pragma unroll
    for (int u = 0; u < NTC; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) a2[u][e] = *reinterpret_cast<const f32x4 *>(Tm + i * LD + u * 32 + 16 * h + 4 * e);
#pragma unroll 1
    for (int nt = 0; nt < NTC; ++nt) {
        const int n = nt * 32 + i;
        float res[16];
        if (!p.bwd) load_tile(bp, nt, res);   // forward: the shortcut operand
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int u = 0; u < NTC; ++u) {
            float b[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) b[s] = buf_load_f32(rw2, (unsigned)((u * 32 + 16 * h + s) * C + n) * 4u);
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = mfma_32x32x2(a2[u][s >> 2][s & 3], b[s], acc);
        }
        const float bias = (!p.bwd && p.bias2) ? p.bias2[n] : 0.f;
        float y[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = p.bwd ? acc[r] : acc[r] + bias + res[r];
        store_tile(o2, nt, y);
    }
}

int launch_cl_pointwise_pair(const PwPairArgs &a, hipStream_t st)
{
    if ((a.C != 32 && a.C != 64) || (long)a.M * a.C * 4 >= (1l << 31)) return DLKA_ERR_UNSUPPORTED;
    PwPairArgs ax = a;
    ax.zero_blocks = 0;
    unsigned blocks = (unsigned)cdiv(a.M, 32);
    if (a.zero.n > 0) {
        if (a.zero.overflow) return DLKA_ERR_WORKSPACE;
        unsigned blk = 0;
        for (int r = 0; r < a.zero.n; ++r) {
            if ((uintptr_t)a.zero.p[r] & 15) return DLKA_ERR_UNSUPPORTED;
            ax.zero.block0[r] = blk;
            blk += (unsigned)cdivl(a.zero.cnt[r], 4096);
        }
        ax.zero.block0[a.zero.n] = blk;
        ax.zero_blocks = (int)blk;
        blocks += blk;
    }
    dim3 grid(blocks), block(64);
    if (a.act_bf16) {
        if (a.C == 32) { auto k = cl_pointwise_pair_kernel<bf16_t, 1>; DLKA_LAUNCH(k, grid, block, 0, st, ax); }
        else { auto k = cl_pointwise_pair_kernel<bf16_t, 2>; DLKA_LAUNCH(k, grid, block, 0, st, ax); }
    } else {
        if (a.C == 32) { auto k = cl_pointwise_pair_kernel<float, 1>; DLKA_LAUNCH(k, grid, block, 0, st, ax); }
        else { auto k = cl_pointwise_pair_kernel<float, 2>; DLKA_LAUNCH(k, grid, block, 0, st, ax); }
    }
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
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
    // few row tiles and a long contraction (C = 128 / 256 at 8^3 / 4^3): four waves per tile split the channel chunks (see the kernel)
    const char *ekw = getenv("DLKA_PW_KW");   // (not cached: a parity test compares both forms)
    const bool kw4 = ekw ? atoi(ekw) == 4 : ((long)cdiv(a.M, 32) * (a.NP / 32) <= 256 && a.CinP >= 128);
    if (kw4) {
        block = dim3(256);
        if (a.act_bf16) { auto k = cl_pointwise_kernel<bf16_t, 4>; DLKA_LAUNCH(k, grid, block, 0, st, ax); }
        else { auto k = cl_pointwise_kernel<float, 4>; DLKA_LAUNCH(k, grid, block, 0, st, ax); }
    } else if (a.act_bf16) { auto k = cl_pointwise_kernel<bf16_t>; DLKA_LAUNCH(k, grid, block, 0, st, ax); }
    else { auto k = cl_pointwise_kernel<float>; DLKA_LAUNCH(k, grid, block, 0, st, ax); }
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

}  // namespace dlka
