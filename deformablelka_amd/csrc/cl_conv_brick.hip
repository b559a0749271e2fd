// Data gradient of the offset-predict conv (3^3, C -> 81: gt = Coff^T goff, 3D/dcn/modules/deform_conv.py:78-84 builds that conv) from an LDS BRICK.
//
//     out[m][n] = aux[m][n] + sum_tap sum_k in[b][k][voxel(m) + tap - 1] * Wp[tap][k][n]        in = grad_offset, PLANAR [B][81][N] fp32
//
// Why a kernel of its own.  cl_conv_wave_kernel<2, 0, 1, 2, 2> fetches its A operand — 16 dwords from 16 planes per lane and (tap, chunk) unit — from
// global memory 27 times over, once per tap: in-situ ablations of that kernel (profiles/r06_notes.md) put 45 of its 91 us at 32^3 on that fetch
// alone (split arithmetic 9, the matrix cores 6, the weight records 7), the planar layout being the better of two bad choices (a channels-last copy
// needs a quarter of the load instructions and four times the cache lines, profiles/r04_notes.md).  Here a workgroup owns 256 consecutive voxels
// (TH rows of W at one depth) and stages their 3 x (TH + 2) x (W + 2) halo ONCE per 32-plane chunk in LDS — read coalesced along the planes' voxels,
// split into its two bf16 terms there and then (once per element instead of once per tap), zero padding written as zeros — as
//     brick[voxel][ hi: 32 x bf16 | lo: 32 x bf16 | 16 bytes of padding ]          144 bytes per voxel
// so that the A operand of v_mfma_f32_32x32x16_bf16 for ANY tap is one 16-byte LDS read per term at a tap-dependent constant offset: no address
// arithmetic, no bounds code, no conversion in the 27-tap loop (the 144-byte row pitch spreads eight lanes' 16-byte accesses over all banks).
// The weights come as the two-term records cl_conv_wave_kernel reads (prep mode 1 | 8), one tap ahead, straight from L2.
// Same products and the same fp32 accumulation as the kernel it replaces; the order of the sum over (tap, chunk) differs (chunks outer): rounding only.
#include <atomic>

#include "cl_args.h"
#include "dlka_kernels.h"

namespace dlka {

constexpr int BRICK_ROW = 144;   // bytes per brick voxel
static std::atomic<long> g_conv_brick_launches{0};   // dlka_conv_brick_launch_count (include/dlka.h): diagnostics

template <int NT, typename T>   // NT = Cout / 32 column tiles per wave; T = storage of `out` (float | bf16_t); `aux` is fp32 when p.aux_f32 or T = float
__global__ __launch_bounds__(512) void cl_conv_brick_kernel(IgemmArgs p, int TH)
{
    DLKA_DYN_SMEM(unsigned char, brick);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int W = p.W, BW = W + 2, BH = TH + 2, nvox = 3 * BH * BW;
    const int hblocks = p.H / TH;
    int bi = blockIdx.x;
    const int hb = bi % hblocks; bi /= hblocks;
    const int d0 = bi % p.D;
    const int b = bi / p.D;
    const int h0 = hb * TH;
    const long mbase = (long)b * p.N + ((long)d0 * p.H + h0) * W;   // first of this workgroup's 256 rows
    const int nchunk = p.CinP / 32;
    const BufRsrc rin = make_rsrc(p.in, (size_t)p.B * p.CinReal * p.N * 4);
    const BufRsrc rw = make_rsrc(p.wp, (size_t)p.K * nchunk * 32 * p.NP * 4);
    const unsigned unit_bytes = (unsigned)(32 * p.NP) * 4u, seg_bytes = (unsigned)p.NP * 16u;
    const unsigned blane = (unsigned)(h * p.NP + i) * 16u;   // this lane's record inside segment (part, mf), column tile 0

    // this lane's A row: voxel r of the workgroup's 256, at brick position (1, hl + 1, wl + 1) for the centre tap
    const int r = 32 * wave + i, hl = r / W, wl = r - hl * W;
    const unsigned abase = (unsigned)((hl * BW + wl) * BRICK_ROW + 32 * h);   // + tap offset + 16 mf (+ 64 for the low term)

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    f32x4 bcur[4 * NT], bnxt[4 * NT];   // [(part * 2 + mf) * NT + t]
    auto load_b = [&](int tap, int ck, f32x4 *bd) {
        const unsigned ub = (unsigned)(tap * nchunk + ck) * unit_bytes + blane;
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int t = 0; t < NT; ++t) bd[(part * 2 + mf) * NT + t] = buf_load_f32x4(rw, ub + (unsigned)((part * 2 + mf) * 2) * seg_bytes + (unsigned)t * 512u);
    };

    const int nitems = nvox * 4;   // (brick voxel, group of 8 planes)
    for (int ck = 0; ck < nchunk; ++ck) {
        load_b(0, ck, bcur);
        // ---- fill: brick voxels x 4 plane groups; consecutive threads = consecutive brick voxels of one plane group (coalesced along w) ----
        for (int it = tid; it < nitems; it += 512) {
            const int pg = it / nvox, vx = it - pg * nvox;
            const int dz = vx / (BH * BW), rem = vx - dz * (BH * BW);
            const int hy = rem / BW, wx = rem - hy * BW;
            const int zd = d0 + dz - 1, zh = h0 + hy - 1, zw = wx - 1;
            const bool ok = ((unsigned)zd < (unsigned)p.D) & ((unsigned)zh < (unsigned)p.H) & ((unsigned)zw < (unsigned)W);
            const int plane0 = ck * 32 + pg * 8;
            const unsigned voff = ok ? (unsigned)(((long)b * p.CinReal + plane0) * p.N + ((long)zd * p.H + zh) * W + zw) * 4u : DLKA_OOB;
            float a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)   // (planes beyond CinReal: padding of the contraction, read as zero)
                a[e] = buf_load_f32(rin, (voff != DLKA_OOB && plane0 + e < p.CinReal) ? voff + (unsigned)e * (unsigned)p.N * 4u : DLKA_OOB);
            bf16x8 hi, lo;
            split_bf16x8(a, hi, lo);
            unsigned char *dst = brick + (size_t)vx * BRICK_ROW + pg * 16;
            *reinterpret_cast<bf16x8 *>(dst) = hi;
            *reinterpret_cast<bf16x8 *>(dst + 64) = lo;
        }
        __syncthreads();
        // ---- 27 taps from the brick ----
#pragma unroll 1
        for (int tap = 0; tap < p.K; ++tap) {
            if (tap + 1 < p.K) load_b(tap + 1, ck, bnxt);
            const int ti = tap / 9, tj = (tap - ti * 9) / 3, tk = tap - ti * 9 - tj * 3;
            const unsigned char *ap = brick + abase + (unsigned)(((ti * BH + tj) * BW + tk) * BRICK_ROW);
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                const bf16x8 ahi = *reinterpret_cast<const bf16x8 *>(ap + 16 * mf), alo = *reinterpret_cast<const bf16x8 *>(ap + 64 + 16 * mf);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 bhi = __builtin_bit_cast(bf16x8, bcur[(0 * 2 + mf) * NT + t]), blo = __builtin_bit_cast(bf16x8, bcur[(1 * 2 + mf) * NT + t]);
                    acc[t] = mfma_32x32x16_bf16(alo, bhi, acc[t]);   // small terms first
                    acc[t] = mfma_32x32x16_bf16(ahi, blo, acc[t]);
                    acc[t] = mfma_32x32x16_bf16(ahi, bhi, acc[t]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4 * NT; ++q) bcur[q] = bnxt[q];
        }
        __syncthreads();   // every wave is done with this chunk's brick
    }

    // ---- epilogue (cl_conv_wave_kernel's): D layout col = lane & 31, row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5) ----
    T *outp = reinterpret_cast<T *>(p.out);
    const T *auxp = reinterpret_cast<const T *>(p.aux);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = t * 32 + i;
        if (n >= p.Cout) continue;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const long mr = mbase + 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * h;
            const long o = mr * p.Cout + n;
            float val = acc[t][q];
            if (p.epi == 3) val += (sizeof(T) == 4 || p.aux_f32) ? p.aux[o] : act_load1(auxp, o);
            act_store1(outp, o, val);
        }
    }
}

// Planar input, channels-last output, 3^3 / stride 1 / padding 1 / dilation 1, two-term weights, epilogues 0 and 3, no tap split, volumes whose rows
// tile into 256-voxel workgroups (W <= 32, 256 % W == 0, H % (256 / W) == 0) and enough of them to fill the chip (DLKA_CONV_BRICK_MIN_WG lowers that
// bar so that small test shapes take this kernel; DLKA_CONV_BRICK=0 switches it off — both read per call: parity tests toggle them).
bool cl_conv_brick_supported(const IgemmArgs &a)
{
    const char *e = getenv("DLKA_CONV_BRICK");
    if (e && e[0] == '0') return false;
    if (a.K != 27 || a.kd != 3 || a.kh != 3 || a.kw != 3 || a.pd != 1 || a.ph != 1 || a.pw != 1 || a.dd != 1 || a.dh != 1 || a.dw != 1) return false;
    if (a.split_bf16 != 2 || a.a_packed || (a.epi != 0 && a.epi != 3) || a.bias || a.CinP % 32 || a.NP % 32 || a.Cout != a.NP) return false;
    if (a.NP != 32 && a.NP != 64) return false;
    if (a.W > 32 || a.W < 8 || 256 % a.W) return false;
    const int TH = 256 / a.W;
    if (a.H % TH) return false;
    const char *mw = getenv("DLKA_CONV_BRICK_MIN_WG");
    if (a.B * a.D * (a.H / TH) < (mw ? atoi(mw) : 128)) return false;   // (a workgroup per CU is what the kernel is built around)
    if ((size_t)3 * (TH + 2) * (a.W + 2) * BRICK_ROW > 160 * 1024) return false;
    if ((size_t)a.B * a.CinReal * a.N * 4 >= (1ull << 31) || (long)a.K * (a.CinP / 32) * 32 * a.NP * 4 >= (1l << 31)) return false;   // 32-bit buffer offsets
    return true;
}

// DLKA_ERR_UNSUPPORTED: the caller takes cl_conv_wave_kernel / cl_igemm_kernel.
int launch_cl_conv_brick(const IgemmArgs &a, hipStream_t st)
{
    if (!cl_conv_brick_supported(a)) return DLKA_ERR_UNSUPPORTED;
    const int NT = a.NP / 32, TH = 256 / a.W;
    const int nwg = a.B * a.D * (a.H / TH);
    const size_t lds = (size_t)3 * (TH + 2) * (a.W + 2) * BRICK_ROW;
#if !defined(HIPEMU)
    static std::atomic<uint64_t> attr_done{0};   // dynamic LDS above 64 KB: per function AND per device (cl_deform_bwd2.hip has the same pattern)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return DLKA_ERR_LAUNCH;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        const void *fns[4] = {reinterpret_cast<const void *>(cl_conv_brick_kernel<1, float>), reinterpret_cast<const void *>(cl_conv_brick_kernel<2, float>),
                              reinterpret_cast<const void *>(cl_conv_brick_kernel<1, bf16_t>), reinterpret_cast<const void *>(cl_conv_brick_kernel<2, bf16_t>)};
        for (int f = 0; f < 4; ++f)
            if (hipFuncSetAttribute(fns[f], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DLKA_ERR_LAUNCH;
        attr_done.fetch_or(bit, std::memory_order_release);
    }
#endif
    dim3 grid(nwg), block(512);
    if (a.act_bf16) {
        if (NT == 1) { auto k = cl_conv_brick_kernel<1, bf16_t>; DLKA_LAUNCH(k, grid, block, lds, st, a, TH); }
        else { auto k = cl_conv_brick_kernel<2, bf16_t>; DLKA_LAUNCH(k, grid, block, lds, st, a, TH); }
    } else {
        if (NT == 1) { auto k = cl_conv_brick_kernel<1, float>; DLKA_LAUNCH(k, grid, block, lds, st, a, TH); }
        else { auto k = cl_conv_brick_kernel<2, float>; DLKA_LAUNCH(k, grid, block, lds, st, a, TH); }
    }
    DLKA_CHECK_LAUNCH();
    g_conv_brick_launches.fetch_add(1, std::memory_order_relaxed);
    return DLKA_OK;
}

}  // namespace dlka

extern "C" long dlka_conv_brick_launch_count(void) { return dlka::g_conv_brick_launches.load(std::memory_order_relaxed); }
