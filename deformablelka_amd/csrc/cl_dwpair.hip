// Two chained channels-last depthwise 3-D convolutions of a SMALL volume in ONE launch (round 5; VERDICT r4 "small-stage path"): the 8^3 and 4^3 stages of the D-LKA
// block (C = 128 / 256: model_components.py:33-39,127-131), where dw 5^3 -> dw 7^3 dilation 3 (LKA3d_deform, transformerblock.py:646-647) and, in the backward pass, their
// data gradients dw 7^3^T -> dw 5^3^T (+ GELU') are two launches of 9 - 13 us each for a few MFLOP: latency, not work.  A whole volume of a channel group fits in LDS there:
//
//   workgroup = (batch b, 4 channels): in[b][:, c0..c0+3] (N <= 512 voxels) and both convs' tap weights are staged in LDS as per-channel planes; conv A reads them and
//   leaves its result in LDS (conv B's input; copied out to global too: it is a saved activation / a gradient the weight gradients read); conv B reads conv A's planes.
//   One launch, one boundary, no round trip through L2 between the two.  (DESIGN 4.2's "a fused dw5 -> dw7 is impossible" holds at 32^3 — the dilated conv needs 19
//   planes of the intermediate — not here.)
//
//   thread = (channel, w-row (d, h)): the W outputs of a row in registers; per (kd, kh) tap the input row is ONE or TWO 16-byte LDS reads and the kw taps two, and — W
//   being a template parameter — the (kw, w) pairs that fall outside the row are dropped at compile time: 34 FMAs (5 / dil 1) and 22 (7 / dil 3) per row at W = 8, of 40
//   and 56.  The (kd, kh) loops run over the valid range only (of the dilated conv's 49 (kd, kh) taps ~7 touch an 8^3 volume).
//   Plane stride N + 4 (W = 8) / N + 32 (W = 4) with lanes ordered (8 rows, channel): the 16-byte reads of a 16-lane group cover the 64 banks once.
//
// Same arithmetic per output as cl_dwconv.hip (fp32 FMAs, bias added last; another summation order: rounding only); same epilogues: the plain one (+ optional bf16 copy of
// an fp32 result) and the GELU-backward one, out = (acc + gelu_add) * gelu'(gelu_x).  bf16 storage: conv B reads conv A's result ROUNDED to bf16, as stored.
// Taken for W in {4, 8}, N = D H W <= 512, C % 4 = 0 and the two cubic "same" shapes of the Synapse block (5 / dil 1 and 7 / dil 3) in either order; anything else
// stays on the per-conv kernels.
// Measured (profiles/r08_notes.md, r09a / r09d): 11.5 / 12.7 us at 8^3 against 12.6 + 8.8, 8.8 / 8.3 us at 4^3 against 11.0 + 6.1; step 10.40 -> 10.21 ms fp32.  The kernel sits on the
// ~5 us floor of a launch in a replayed graph plus three barriers (batched staging loads and a prefetched tap loop changed nothing); a W = 16 / two-channel variant for the 16^3
// stage was SLOWER than the two row kernels (33 + 41 us against 13 + 19: its LDS row reads are 2-way conflicted and a 4096-voxel volume is work, not latency) and is not here.
#include <atomic>

#include "cl_args.h"
#include "dlka_kernels.h"

namespace dlka {

namespace {
constexpr int PG = 4;       // channels per workgroup
constexpr int PNT = 256;    // most threads per workgroup

__host__ __device__ constexpr int pair_stride(int N, int WT) { return N + (WT == 4 ? 32 : 4); }

// one output row (d, hh) of a depthwise conv over the LDS plane `src` ([D][H][WT] floats) with tap weights `wl` ([KW * KW][8] floats, kw fastest)
template <int KW, int DIL, int WT>
__device__ __forceinline__ void dwpair_row(const float *__restrict__ src, const float *__restrict__ wl, int D, int H, int d, int hh, float (&acc)[WT])
{
    constexpr int P = (KW - 1) * DIL / 2;   // "same" padding
#pragma unroll
    for (int w = 0; w < WT; ++w) acc[w] = 0.f;
    const int i_lo = d >= P ? 0 : (P - d + DIL - 1) / DIL, i_hi = min(KW - 1, (D - 1 + P - d) / DIL);
    const int j_lo = hh >= P ? 0 : (P - hh + DIL - 1) / DIL, j_hi = min(KW - 1, (H - 1 + P - hh) / DIL);
    for (int i = i_lo; i <= i_hi; ++i) {
        const int zd = d + i * DIL - P;
        for (int j = j_lo; j <= j_hi; ++j) {
            const int zh = hh + j * DIL - P;
            const f32x4 *row = reinterpret_cast<const f32x4 *>(src + (zd * H + zh) * WT);
            float seg[WT];
#pragma unroll
            for (int q = 0; q < WT / 4; ++q) {
                const f32x4 v = row[q];
                seg[4 * q] = v[0]; seg[4 * q + 1] = v[1]; seg[4 * q + 2] = v[2]; seg[4 * q + 3] = v[3];
            }
            const f32x4 *wq = reinterpret_cast<const f32x4 *>(wl + (i * KW + j) * 8);
            const f32x4 w0 = wq[0], w1 = wq[1];
            const float wv[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
            for (int k = 0; k < KW; ++k)
#pragma unroll
                for (int w = 0; w < WT; ++w) {
                    const int zw = w + k * DIL - P;   // compile-time after unrolling
                    if (zw >= 0 && zw < WT) acc[w] = fmaf(wv[k], seg[zw], acc[w]);
                }
        }
    }
}

template <int KW>
__device__ __forceinline__ void dwpair_stage_weights(float *wl, const float *__restrict__ wp, int C, int c0, int tid, int nt)
{
    constexpr int K = KW * KW * KW;
    for (int idx = tid; idx < K * PG; idx += nt) {
        const int t = idx / PG, cc = idx % PG;
        wl[(cc * KW * KW + t / KW) * 8 + t % KW] = wp[(long)t * C + c0 + cc];
    }
}
}  // namespace

template <typename T, int KWA, int DILA, int KWB, int DILB, int WT>
__global__ __launch_bounds__(PNT) void cl_dwpair_small_kernel(DwPairArgs p)
{
    DLKA_DYN_SMEM(float, lds);
    const int D = p.D, H = p.H, N = D * H * WT, S = pair_stride(N, WT);
    float *A = lds, *T1 = A + PG * S, *WA = T1 + PG * S, *WB = WA + PG * KWA * KWA * 8;
    const int b = blockIdx.x, c0 = blockIdx.y * PG;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int c = (tid >> 3) & (PG - 1), rs = ((tid >> 5) << 3) | (tid & 7), nrs = nt / PG;
    const T *inp = reinterpret_cast<const T *>(p.in);
    const long base = (long)b * N * p.C + c0;
    constexpr bool LO = sizeof(T) == 4;
    for (int v = tid; v < N; v += nt) {
        const f32x4 x = act_load4(inp, base + (long)v * p.C);
#pragma unroll
        for (int cc = 0; cc < PG; ++cc) A[cc * S + v] = x[cc];
    }
    dwpair_stage_weights<KWA>(WA, p.wpA, p.C, c0, tid, nt);
    dwpair_stage_weights<KWB>(WB, p.wpB, p.C, c0, tid, nt);
    __syncthreads();
    {
        const float bias = p.biasA ? p.biasA[c0 + c] : 0.f;
        for (int r = rs; r < D * H; r += nrs) {
            float acc[WT];
            dwpair_row<KWA, DILA, WT>(A + c * S, WA + c * KWA * KWA * 8, D, H, r / H, r % H, acc);
            f32x4 *dst = reinterpret_cast<f32x4 *>(T1 + c * S + r * WT);
#pragma unroll
            for (int q = 0; q < WT / 4; ++q) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float val = acc[4 * q + e] + bias;
                    if (!LO) val = __uint_as_float((unsigned)bf16_bits(val) << 16);   // what the stored copy holds
                    o[e] = val;
                }
                dst[q] = o;
            }
        }
    }
    __syncthreads();
    {   // conv A's result -> global
        T *oa = reinterpret_cast<T *>(p.outA);
        bf16_t *oa_lo = reinterpret_cast<bf16_t *>(p.outA_lo);
        for (int v = tid; v < N; v += nt) {
            f32x4 o;
#pragma unroll
            for (int cc = 0; cc < PG; ++cc) o[cc] = T1[cc * S + v];
            act_store4(oa, base + (long)v * p.C, o);
            if (LO && oa_lo) act_store4(oa_lo, base + (long)v * p.C, o);
        }
    }
    {   // conv B, into the input's planes (dead since the barrier)
        const float bias = p.biasB ? p.biasB[c0 + c] : 0.f;
        for (int r = rs; r < D * H; r += nrs) {
            float acc[WT];
            dwpair_row<KWB, DILB, WT>(T1 + c * S, WB + c * KWB * KWB * 8, D, H, r / H, r % H, acc);
            f32x4 *dst = reinterpret_cast<f32x4 *>(A + c * S + r * WT);
#pragma unroll
            for (int q = 0; q < WT / 4; ++q) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[4 * q + e] + bias;
                dst[q] = o;
            }
        }
    }
    __syncthreads();
    {
        T *ob = reinterpret_cast<T *>(p.outB);
        bf16_t *ob_lo = reinterpret_cast<bf16_t *>(p.outB_lo);
        const T *gxp = reinterpret_cast<const T *>(p.gelu_x), *gap = reinterpret_cast<const T *>(p.gelu_add);
        for (int v = tid; v < N; v += nt) {
            const long o_ = base + (long)v * p.C;
            f32x4 o;
#pragma unroll
            for (int cc = 0; cc < PG; ++cc) o[cc] = A[cc * S + v];
            if (gxp) {
                const f32x4 gx = act_load4(gxp, o_), ga = act_load4(gap, o_);
#pragma unroll
                for (int cc = 0; cc < PG; ++cc) o[cc] = (o[cc] + ga[cc]) * dgelu_f(gx[cc]);
            }
            act_store4(ob, o_, o);
            if (LO && ob_lo) act_store4(ob_lo, o_, o);
        }
    }
}

bool cl_dwpair_small_supported(const DwPairArgs &a)
{
    const long N = (long)a.D * a.H * a.W;
    auto shape_ok = [](int k, int dil, int pad) { return (k == 5 && dil == 1 && pad == 2) || (k == 7 && dil == 3 && pad == 9); };
    if (N > 512 || (a.W != 4 && a.W != 8) || a.C % PG || a.B <= 0 || a.D <= 0 || a.H <= 0) return false;
    if (!shape_ok(a.kA, a.dA, a.pA) || !shape_ok(a.kB, a.dB, a.pB) || a.kA == a.kB) return false;
    if (a.act_bf16 && (a.outA_lo || a.outB_lo)) return false;   // (the bf16 copies ride in the fp32 kernels only)
    return true;
}

template <typename T, int KWA, int DILA, int KWB, int DILB>
static int launch_pair_t(const DwPairArgs &a, hipStream_t st)
{
    const int N = a.D * a.H * a.W, rows = a.D * a.H;
    const size_t lds = (size_t)(2 * PG * pair_stride(N, a.W) + PG * 8 * (KWA * KWA + KWB * KWB)) * 4;
    int nt = (rows * PG + 63) / 64 * 64;
    if (nt > PNT) nt = PNT;
    dim3 grid(a.B, a.C / PG), block(nt);
    if (a.W == 8) { auto k = cl_dwpair_small_kernel<T, KWA, DILA, KWB, DILB, 8>; DLKA_LAUNCH(k, grid, block, lds, st, a); }
    else { auto k = cl_dwpair_small_kernel<T, KWA, DILA, KWB, DILB, 4>; DLKA_LAUNCH(k, grid, block, lds, st, a); }
    DLKA_CHECK_LAUNCH();
    return DLKA_OK;
}

static std::atomic<long> g_dwpair_launches{0};   // dlka_dwpair_launch_count (include/dlka.h): diagnostics

// DLKA_ERR_UNSUPPORTED: the caller runs the two convs through launch_cl_dwconv
int launch_cl_dwpair_small(const DwPairArgs &a, hipStream_t st)
{
    if (!cl_dwpair_small_supported(a)) return DLKA_ERR_UNSUPPORTED;
    g_dwpair_launches.fetch_add(1, std::memory_order_relaxed);
    if (a.act_bf16) return a.kA == 5 ? launch_pair_t<bf16_t, 5, 1, 7, 3>(a, st) : launch_pair_t<bf16_t, 7, 3, 5, 1>(a, st);
    return a.kA == 5 ? launch_pair_t<float, 5, 1, 7, 3>(a, st) : launch_pair_t<float, 7, 3, 5, 1>(a, st);
}

}  // namespace dlka

extern "C" long dlka_dwpair_launch_count(void) { return dlka::g_dwpair_launches.load(std::memory_order_relaxed); }
