// h16_halo.hip — the 3x3 / stride 1 / pad 1 convolutions of the 16-bit throughput mode (bf16 or fp16 activations and
// weights, fp32 accumulation) on the design of the split-fp32 halo kernel (conv_f32x.hip), re-balanced for ONE MFMA per
// product instead of three:
//   * a workgroup owns an 8 x 16 (8 x 8) patch of output pixels and stages the 10 x 18 (10 x 10) halo of the patch ONCE
//     per 32-channel slab (16-bit: 64 bytes per halo pixel, one LDS plane); the nine taps read their MFMA A fragments from
//     it at a wavefront-uniform offset, through explicit ds_read_b128 statements one step ahead;
//   * GroupNorm(+swish) of the input is applied while the halo is staged (fp32 arithmetic, one rounding to 16 bits), from
//     {mean, rstd} per (image, group): the stand-alone normalise pass (statistics + finalize + apply: three launches and two
//     trips over the activation in this mode) disappears; the statistics of the OUTPUT leave the epilogue as per-chunk
//     partial sums exactly like the split-fp32 kernel's, so ResnetBlock = conv -> fold -> conv -> fold here too;
//   * weights are stored in MFMA-fragment order (one plane): per (32-row tile, 32-element K slab) 128 pieces of 16 bytes,
//     piece = (k-step * 2 + k-half) * 32 + row — the B operand of one v_mfma_f32_32x32x16 is one contiguous kilobyte that
//     goes straight from L2 / L1 to registers, two taps ahead;
//   * wavefront layout (SGAM_HWGM): the four wavefronts sit side by side along N — each owns all BM rows x 32 channels, so
//     a weight fragment fetched (1 KB, L2 / L1 -> registers) feeds BM / 32 MFMAs and nobody else in the workgroup fetches
//     it, and every A fragment is read from LDS by all four.  At the matrix pipe's rate that is 32 B/clk/CU through the
//     vector L1 (of ~64) and 128 B/clk/CU of ds_read_b128 (of 256).  Rounds 1 - 2 ran a 2 x 2 grid (BM / 2 rows x 64
//     channels per wavefront: half the LDS reads, but every weight fragment fetched by two wavefronts = 64 B/clk/CU, the
//     whole L1 port: the loop took 1.3 x its MFMA time with or without the MFMAs); the 1 x 4 layout is -5 % on the
//     256^2 x 128 layer at B = 8, -8 ... -16 % on the 16^2 ... 128^2 maps, +2.7 % frames/s for the bf16 loop (A / B on one box,
//     scripts/r03x.sh);
//   * epilogue (SGAM_HDIRECT = 1, the default): the product is computed TRANSPOSED (weights = MFMA rows, pixels = columns)
//     and the weight rows of a 32-channel tile are packed so that a lane's sixteen accumulator slots are sixteen
//     CONSECUTIVE channels of one pixel: they leave as 16-byte stores with bias, residual and the output statistics applied
//     in registers — no LDS transpose (70 -> 31 KB of LDS; the generic 16-bit kernel stores 2 bytes per instruction);
//   * the maps too small to fill the chip with whole-K workgroups (16^2 ... 64^2) split the K slabs over grid.y: fp32
//     partial tiles in the same lane-owned layout, then h16_splitk_reduce_kernel adds them in a fixed order, applies bias /
//     residual, rounds and leaves the per-chunk statistics (so those layers, too, normalise while staging).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "sgam_common.h"

#ifndef SGAM_HDIRECT
#define SGAM_HDIRECT 1     // 1: the product is computed TRANSPOSED (weights = MFMA rows, pixels = columns) and leaves the
#endif                     //    accumulators straight for memory; 0: pixels = rows, LDS transpose in the epilogue
#ifndef SGAM_HWGM
#define SGAM_HWGM 1        // wavefront layout of the 64- / 128-row tiles: 1 = four side by side along N (all BM rows x 32 channels
#endif                     //    each), 2 = a 2 x 2 grid (BM / 2 rows x 64 channels each: twice the weight-fragment bytes through L1)
#ifndef SGAM_HGN_MAXC
#define SGAM_HGN_MAXC 1024  // most input channels the fused GroupNorm takes (its per-channel scale / shift table lives in LDS)
#endif
#ifndef SGAM_HSB
#define SGAM_HSB 2         // scheduling barriers in the slab body: 0 none, 1 in front of the staging arithmetic of tap 1, 2 at every tap
#endif
#ifndef SGAM_HFD2
#define SGAM_HFD2 1        // A-fragment read-ahead of the one-role kernel, in steps: tiles of two row tiles per wavefront (64-row) ...
#endif
#ifndef SGAM_HFD4
#define SGAM_HFD4 1        // ... and of four (128-row; 256-row)
#endif
#ifndef SGAM_HNBR
#define SGAM_HNBR 3        // weight-fragment ring of the 128-row tile, in taps: 3 (two taps ahead) or 6 (five ahead: every weight
#endif                     //    load a slab still needs is in the in-order vector-memory queue BEFORE the next halo load; slab loop unrolled by 2)
#ifndef SGAM_HNBR64
#define SGAM_HNBR64 6      // ... of the 64-row tile's whole-K launches (grids of <= 256 workgroups: ONE wavefront per SIMD, two taps of its
#endif                     //    own MFMAs = 256 cycles do not cover an L2 round trip: 13.2 -> 12.3 us per launch in the bf16 frame, +2.3 % frames/s, A / B x 3);
                           //    launches with an odd slab count per workgroup (split-K plans) keep the ring of 3
#ifndef SGAM_HNBRF
#define SGAM_HNBRF 6       // ... and of its folding form (split-K workgroups of the 16^2 / 32^2 maps: one or two slabs)
#endif
#ifndef SGAM_HLT
#define SGAM_HLT 0         // tap at which the staged halo is stored and the next halo load issued (0: NH + 1, right behind the last piece)
#endif
#ifndef SGAM_HPEEL
#define SGAM_HPEEL 1       // 1: the last two slabs of a workgroup are peeled: no staging of a slab that does not exist, and (SGAM_HRPF) the
#endif                     //    residual tile is requested where the (dead) halo load of the second-to-last slab stood, not in the epilogue
#ifndef SGAM_HRPF
#define SGAM_HRPF 1        // residual prefetch of the peeled form: 0 off, 1 the 128-row tile only (the 64-row tile would pay its fourth
#endif                     //    wavefront per SIMD for the 16 registers), 2 every tile
#ifndef SGAM_HSWISH
#define SGAM_HSWISH 0      // fused GroupNorm + swish arithmetic: 0 fp32 (v_exp_f32 / v_rcp_f32, one rounding to 16 bits), 1 packed fp16
#endif                     //    after the affine step (v_pk_* + v_exp_f16 / v_rcp_f16) — an agreement-rate experiment, DESIGN.md 5.5d
#ifndef SGAM_HABLATE
#define SGAM_HABLATE 0     // timing experiments only (results are wrong when != 0): 1 no MFMAs, 2 no epilogue, 4 no main loop,
                           // 8 no weight-fragment loads in the loop, 16 no halo staging in the loop, 32 stores dropped
#endif

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int HT> struct HH;
template <> struct HH<0> {
    __device__ static __forceinline__ float to_f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
    __device__ static __forceinline__ unsigned short from_f(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
    __device__ static __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct HH<1> {
    __device__ static __forceinline__ float to_f(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
    __device__ static __forceinline__ unsigned short from_f(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
    __device__ static __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

struct HHParams {
    const unsigned short *x, *w, *res;
    const float *bias;
    void *out;                    // 16-bit or fp32 (out_f32)
    int B, Hi, Wi, Cin, Ho, Wo, N, ups;
    int lda, ldb, ldc, ldr, n_valid, out_f32;
    int M, slabs;
    int ksplit, slabs_per_split;  // split-K: grid.y = ksplit ranges of the K slabs, fp32 partial tiles to `ws`
    float *ws;                    // [ksplit][M][N] (ksplit > 1)
    int gx, gy, xcd_swizzle;
    unsigned x_bytes, w_bytes;
    const float *gn_stats, *gn_gamma, *gn_beta;   // GroupNorm(+swish) of the input: {mean, rstd} [B][32][2] + affine [Cin]
    int gn_swish;
    double *gn_partial;           // optional: per-(tile row half, group) {sum, sumsq} of the output
    int gn_cpg;
    // GNF: the statistics of the INPUT as its producer's chunk partials [B][gn_chunks_in <= 16][32][2] (the group-major split-K
    // combine of a small map), folded by the kernel itself instead of a fold launch
    const double *gn_partial_in;
    int gn_chunks_in;
    float gn_inv_n, gn_eps;
};

__device__ __forceinline__ unsigned hsel(bool c, unsigned a, unsigned b) {
    const unsigned m = 0u - (unsigned)c;
    return (a & m) | (b & ~m);
}

// a lane's contribution of four stored values to the output statistics {sum, sum of squares}: ONE spelled sequence of roundings
// (explicit fused multiply-adds, no further contraction) shared by the one-role and the producer / consumer kernel — left to the
// compiler's fp-contract choice the two kernels rounded the sum of squares differently on the fp32-output path (the experimental
// bit-identity test of round 5 found it: tensors equal, chunk statistics one ulp apart)
__device__ __forceinline__ void hstat_add(float &s, float &ss, const f32x4 v) {
#pragma clang fp contract(off)
    s += (v[0] + v[1]) + (v[2] + v[3]);
    ss += __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], v[0] * v[0])));
}

__device__ __forceinline__ void hxcd_block(const HHParams &p, int &bx, int &by) {
    const unsigned L = blockIdx.x, T = gridDim.x;
    unsigned Lp = L;
    if (p.xcd_swizzle) {
        const unsigned q = T >> 3, r = T & 7u, xcd = L & 7u, idx = L >> 3;
        Lp = xcd * q + (xcd < r ? xcd : r) + idx;
    }
    bx = (int)(Lp % (unsigned)p.gx);
    by = (int)(Lp / (unsigned)p.gx);
}

// `SW`: the fused GroupNorm is followed by swish (a template parameter, not a flag: a run-time test per staged piece cuts the
// main loop into a dozen basic blocks and the MFMA / VALU interleaving stops at each of their borders)
template <int BM, int BN, int HT, bool GN, bool UPS, bool SW = true, bool GNF = false, int NB = 3>
__global__ __launch_bounds__(256, BM == 256 ? 1 : 2) void conv3x3_h16_halo_kernel(const HHParams p) {
    // BM = 256 (a 16 x 16 patch, every wavefront 128 rows x 64 channels, TM = 4: a weight fragment fetched feeds four MFMAs,
    // one workgroup per CU with the 128 accumulators in AccVGPRs) compiles and passes the tests but is not dispatched:
    // measured 27-29 us against 25 for BM = 128 on the 256 x 256 x 128 layer (one wavefront per SIMD hides less than the
    // halved fragment stream saves).
    constexpr int TH = BM == 256 ? 16 : 8, TW = BM / TH, TWS = (TW == 16) ? 4 : 3;
    constexpr int HROWS = UPS ? TH / 2 + 2 : TH + 2, HWID = UPS ? TW / 2 + 2 : TW + 2, HR = HROWS * HWID;
    static_assert(!(UPS && GN), "no GroupNorm precedes an upsampling conv");
    static_assert(!GNF || (GN && BM == 64), "GNF = GroupNorm statistics folded from the producer's chunk partials: the 64-row GN kernel");
    static_assert(BM == 256 || BM == 128 || BM == 64, "16 x 16, 8 x 16 or 8 x 8 output patches");
    static_assert(BN == 128, "2 x 2 wavefronts of 64 channels, or 1 x 4 of 32");
    constexpr int WGM_ = BM == 256 ? 2 : SGAM_HWGM, WGN_ = 4 / WGM_;
    static_assert(WGM_ == 2 || (WGM_ == 1 && SGAM_HDIRECT), "the 1 x 4 layout has the direct epilogue only");
    constexpr int XBK = 32, XLD = XBK + 8;
    constexpr int TM = BM / (32 * WGM_), TN = BN / (32 * WGN_);
    constexpr int RH = WGM_ == 1 ? 2 : 1;               // row halves of the tile (= statistics chunks) a wavefront covers
    constexpr int LP = UPS ? ((TW == 16) ? 408 : 240) : ((TW == 16) ? 768 : 448);   // line pitch (halfs), see conv_f32x.hip
    constexpr int HPL = HROWS * LP;                     // halfs per halo buffer (one plane)
    constexpr int NH = (HR * 4 + 255) / 256;            // 16-byte halo pieces (8 channels) per thread
    constexpr int OP_BYTES = 2 * HPL * 2;
    constexpr int WM = 32 * TM, WN = 32 * TN, LDR = WN + 4;
    constexpr int EPI_BYTES = SGAM_HDIRECT ? 4 * RH * 2 * TN * 8 * 4 : 4 * WM * LDR * 4;
    constexpr int SM_BYTES = OP_BYTES > EPI_BYTES ? OP_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned short smem[SM_BYTES / 2];
    // GroupNorm of the input as per-channel {scale = rstd * gamma, shift = beta - mean * scale}, formed ONCE per workgroup in the
    // prologue (same expressions and order as before) and read back from LDS per slab.  Round 3 formed them per slab from
    // global loads issued right behind the halo loads of slab s + 2: the vector-memory queue returns in order, so the
    // `s_waitcnt vmcnt(1)` in front of `rstd * gamma` waited out the halo loads' trip to L2 / HBM as well — every wavefront,
    // every slab.
    constexpr int GN_TAB = (GN && !GNF) ? SGAM_HGN_MAXC : 4;
    __shared__ __attribute__((aligned(16))) float gn_tab[2][GN_TAB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = WGM_ == 2 ? wave >> 1 : 0, wn = WGM_ == 2 ? (wave & 1) : wave;
    int bx, by;
    hxcd_block(p, bx, by);
    const int n0 = by * BN;
    const int tiles_x = p.Wo / TW, tiles_img = tiles_x * (p.Ho / TH);
    const int b = bx / tiles_img;
    const int t_img = bx - b * tiles_img;
    const int ty0 = (t_img / tiles_x) * TH, tx0 = (t_img % tiles_x) * TW;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)p.w_bytes, 0x00020000);

    unsigned h_off[NH];
    int h_lds[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        // a thread past the end of the halo stages an earlier piece a second time (same bytes to the same place):
        // every store is unconditional and the loop keeps one basic block
        int idx = tid + 256 * j;
        if (idx >= HR * 4) idx -= (NH > 1 ? 256 : HR * 4);        // (NH == 1: another thread's piece, same bytes again)
        const int row = idx >> 2, c8 = idx & 3;
        const int hy = row / HWID, hx = row - hy * HWID;
        const int iy = (UPS ? ty0 / 2 : ty0) + hy - 1, ix = (UPS ? tx0 / 2 : tx0) + hx - 1;
        const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        h_off[j] = ok ? (unsigned)(((b * p.Hi + iy) * p.Wi + ix) * p.lda + c8 * 8) * 2u : 0xFFFFFFFFu;
        h_lds[j] = hy * LP + hx * XLD + c8 * 8;
    }
    unsigned bf_off[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nt = (n0 + wn * (BN / WGN_) + j * 32) >> 5;
        bf_off[j] = ((unsigned)nt * ((unsigned)p.ldb / 32u) * 128u + (unsigned)lane) * 16u;
    }

    u32x4 hreg[NH];
    float gsc[8], gsh[8];                  // GroupNorm scale / shift of this thread's 8 channels of the slab in flight
    auto hload_issue = [&](int ch, bool live) {
        const unsigned coff = (unsigned)ch * (XBK * 2u);       // wave-uniform: the load's scalar offset (no VALU per load)
#pragma unroll
        for (int j = 0; j < NH; ++j)
            hreg[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(live ? h_off[j] : 0xFFFFFFFFu), (int)coff, 0);
    };
    auto hparams = [&](int ch, bool live) {
        if constexpr (GN) {
            const int c = (live ? ch : 0) * XBK + (tid & 3) * 8;
            if constexpr (!GNF) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(&gn_tab[0][c + 4 * h]);
                    const f32x4 d = *reinterpret_cast<const f32x4 *>(&gn_tab[1][c + 4 * h]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        gsc[4 * h + e] = a[e];
                        gsh[4 * h + e] = d[e];
                    }
                }
                return;
            }
            const int cpg = p.Cin / 32;
            float fmean = 0.f, frstd = 0.f;
            if constexpr (GNF) {
                // (host: cpg >= 8, so the thread's eight channels are ONE group.)  The staging map gives the sixteen lanes l,
                // l ^ 4, l ^ 8, l ^ 16, l ^ 32 of a wavefront the same eight channels: lane sub = l >> 2 fetches chunk `sub`
                // (all loads in flight at once), a four-step xor butterfly adds them in a fixed order, in fp64, and the result is
                // finished like gn_finalize_stats_kernel
                const int sub = (tid & 63) >> 2;
                typedef double f64x2 __attribute__((ext_vector_type(2)));
                const double *q = p.gn_partial_in + ((int64_t)b * p.gn_chunks_in * 32 + c / cpg) * 2;
                const f64x2 a = sub < p.gn_chunks_in ? *reinterpret_cast<const f64x2 *>(q + (int64_t)sub * 64) : f64x2{0.0, 0.0};
                double ps = a[0], pss = a[1];
#pragma unroll
                for (int o = 4; o < 64; o <<= 1) {
                    ps += __shfl_xor(ps, o, 64);
                    pss += __shfl_xor(pss, o, 64);
                }
                const double m = ps * (double)p.gn_inv_n;
                double var = pss * (double)p.gn_inv_n - m * m;
                if (var < 0.0) var = 0.0;
                fmean = (float)m;
                frstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {            // two float4 halves: a 4-channel group never straddles one
                const int g = (c + 4 * h) / cpg;
                const float mean = GNF ? fmean : p.gn_stats[(b * 32 + g) * 2], rstd = GNF ? frstd : p.gn_stats[(b * 32 + g) * 2 + 1];
                const f32x4 ga = *reinterpret_cast<const f32x4 *>(p.gn_gamma + c + 4 * h);
                const f32x4 be = *reinterpret_cast<const f32x4 *>(p.gn_beta + c + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gsc[4 * h + e] = rstd * ga[e];
                    gsh[4 * h + e] = be[e] - mean * gsc[4 * h + e];
                }
            }
        }
    };
    auto hload = [&](int ch, bool live) {
        hload_issue(ch, live);
        hparams(ch, live);
    };
    auto hprep_piece = [&](const int j) {
        if constexpr (GN) {
            u32x4 q = hreg[j];
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                float v0 = HH<HT>::to_f((unsigned short)(q[w2] & 0xFFFFu)), v1 = HH<HT>::to_f((unsigned short)(q[w2] >> 16));
                v0 = v0 * gsc[2 * w2] + gsh[2 * w2];
                v1 = v1 * gsc[2 * w2 + 1] + gsh[2 * w2 + 1];
                if constexpr (SW && SGAM_HSWISH == 1) {
                    // the affine step in fp32, everything behind it on PAIRS in packed fp16: y (v_cvt_pk_f16_f32), -y log2 e
                    // (v_pk_mul_f16), 2^. per half (v_exp_f16: no packed transcendental exists), 1 + e (v_pk_add_f16), 1 / . per half
                    // (v_rcp_f16), y r (v_pk_mul_f16; bf16: two fp32 products and one v_cvt_pk_bf16_f32)
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    h2 y;
                    y[0] = (_Float16)v0;
                    y[1] = (_Float16)v1;
                    const h2 z = y * (h2){(_Float16)-1.4426950408889634f, (_Float16)-1.4426950408889634f};
                    h2 e, r;
                    e[0] = __builtin_exp2f16(z[0]);
                    e[1] = __builtin_exp2f16(z[1]);
                    const h2 d = e + (h2){(_Float16)1.0f, (_Float16)1.0f};
                    r[0] = __builtin_amdgcn_rcph(d[0]);
                    r[1] = __builtin_amdgcn_rcph(d[1]);
                    if constexpr (HT == 1) {
                        q[w2] = __builtin_bit_cast(unsigned, y * r);
                        continue;
                    } else {
                        v0 = (float)y[0] * (float)r[0];
                        v1 = (float)y[1] * (float)r[1];
                    }
                } else if constexpr (SW) {
                    v0 = sgam_swish(v0);
                    v1 = sgam_swish(v1);
                }
                q[w2] = (unsigned)HH<HT>::from_f(v0) | ((unsigned)HH<HT>::from_f(v1) << 16);
            }
            if (h_off[j] == 0xFFFFFFFFu) q = u32x4{0u, 0u, 0u, 0u};       // zero padding applies to the normalised tensor
            hreg[j] = q;
        }
    };
    auto hstore = [&](int hb) {
        unsigned short *halo = smem + hb * HPL;
#pragma unroll
        for (int j = 0; j < NH; ++j)
            *reinterpret_cast<u32x4 *>(halo + h_lds[j]) = hreg[j];
    };

    // weight-fragment ring: SGAM_HNBR = 3 sets, loaded two taps ahead (two workgroups per CU share the registers), or 6 sets, five
    // taps ahead (round 5: the vector-memory queue returns IN ORDER, so a weight fragment requested behind the halo load of slab
    // s + 2 — HBM / MALL latency — cannot be consumed before that load has landed; with three sets the first such fragment is
    // needed three taps after the halo request, with six it is needed when the halo itself is, one slab later) — or, for the
    // one-workgroup-per-CU 256-row tile, one set per tap, each refilled for the NEXT slab as soon as its tap is done (nine taps
    // = 4 600 MFMA cycles ahead: an L2 round trip is ~0.7 us, two taps of this kernel are 0.2)
    constexpr int NBR = BM == 256 ? 9 : NB;
    static_assert(NBR == 3 || NBR == 6 || NBR == 9, "ring of 3 (two taps ahead), 6 (five ahead) or 9 (one set per tap)");
    static_assert(NBR != 6 || (SGAM_HPEEL && SGAM_HDIRECT), "the ring of six is written for the peeled slab loop");
    constexpr int SUN = NBR == 6 ? 2 : 1;  // slabs per trip of the slab loop: the set of (slab, tap) must be a compile-time index, 9 taps mod 6 repeat every second slab
    u32x4 bq[NBR][TN][2];                  // [(slab phase + tap) % NBR][n tile][k-step]
    auto bload = [&](const int set, int tap, int ch, bool live) {
        const unsigned koff = (unsigned)(tap * p.Cin + ch * XBK) * 64u;   // 2048 bytes per (row tile, slab); scalar offset
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const unsigned vo = live ? bf_off[j] : 0xFFFFFFF0u;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                bq[set][j][kk] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)vo, (int)(koff + (unsigned)(kk * 1024)), 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 8;
    int a_base[TM], a_py[TM], a_px[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * (BM / WGM_) + i * 32 + frag_row;
        a_py[i] = r >> TWS;
        a_px[i] = r & (TW - 1);
        a_base[i] = a_py[i] * LP + a_px[i] * XLD + frag_k;
    }

    const int s0 = (int)blockIdx.y * p.slabs_per_split;            // this workgroup's range of K slabs (split-K: grid.y)
    const int s1 = min(p.slabs, s0 + p.slabs_per_split);
    int hcur = 0;
    hload_issue(s0, true);
    if constexpr (NBR == 9) {
#pragma unroll
        for (int t = 0; t < 9; ++t) bload(t, t, s0, true);
    } else {
#pragma unroll
        for (int t = 0; t < NBR - 1; ++t) bload(t, t, s0, true);
    }
    if constexpr (GN && !GNF) {
        // (behind the first halo and weight loads, so that its own round trip overlaps theirs)
        const int cpg = p.Cin / 32;
        constexpr int CPT = SGAM_HGN_MAXC / 256;                   // channels per thread, at most
        float mr[CPT][2];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int c = tid + 256 * k;
            if (c < p.Cin) mr[k][0] = p.gn_stats[(b * 32 + c / cpg) * 2], mr[k][1] = p.gn_stats[(b * 32 + c / cpg) * 2 + 1];
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int c = tid + 256 * k;
            if (c < p.Cin) {
                const float sc = mr[k][1] * p.gn_gamma[c];
                gn_tab[0][c] = sc;
                gn_tab[1][c] = p.gn_beta[c] - mr[k][0] * sc;
            }
        }
        __syncthreads();
    }
    hparams(s0, true);
#pragma unroll
    for (int j = 0; j < NH; ++j) hprep_piece(j);
    hstore(0);
    if constexpr (SGAM_HPEEL && SGAM_HDIRECT && BM != 256) {
        // (a workgroup that walks ONE slab — split-K plans of the 16^2 maps — has no second halo to request and, in the folding
        // form, no second fold of the chunk statistics to compute: one uniform branch, outside the slab loop)
        if (s0 + 1 < s1) hload(s0 + 1, true);
    } else {
        hload(s0 + 1, s0 + 1 < s1);
    }
    __syncthreads();

    // A fragments are read FD steps (of TM MFMAs) ahead into a ring of FD + 1 register sets.  One step ahead covers 32 TM cycles of
    // MFMA work (128 at TM = 4, 64 at TM = 2: less than an LDS round trip); two to four steps were measured (scripts/r04p.sh):
    // -4 % on the 64^2 x 256 layer behind a cache flush, nothing in the frame — the small-map launches (one wavefront per SIMD) are
    // the sum of a 3 us prologue, a 2 us epilogue and slabs in which staging arithmetic, weight loads and MFMAs of ONE wavefront
    // follow each other (ablations: 17.2 us; 14.0 without staging, 15.1 without weight loads, 11.5 without both), not LDS latency
    constexpr int FD = TM >= 4 ? SGAM_HFD4 : SGAM_HFD2;
    u32x4 fa[FD + 1][TM];
    const unsigned short *hb = smem;
    auto afrag = [&](const int set, const int tap, const int kk) {          // upsampling form: per-lane source pixel
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned short *ah = hb + (((a_py[i] + ky - 1) >> 1) + 1) * LP + (((a_px[i] + kx - 1) >> 1) + 1) * XLD + frag_k;
            fa[set][i] = *reinterpret_cast<const u32x4 *>(ah + kk * 16);
        }
    };
#define HDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
    unsigned a_lds[TM];
    auto afrag_asm = [&](const int set, const int tap, const int kk) {
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int i = 0; i < TM; ++i) HDS_READ(fa[set][i], a_lds[i], 2 * (ky * LP + kx * XLD + kk * 16));
    };
    // all reads but the newest `younger` (a multiple of TM <= FD TM: the sets issued for later steps) have landed; the fragment
    // registers are tied to the wait so that no MFMA moves above it.  EVERY count up to the asserted maximum of 12 is spelled, so
    // whatever read-ahead depth the build parameters select (SGAM_HFD2 / SGAM_HFD4) finds its wait: a count without an arm would
    // emit no s_waitcnt at all and the MFMAs would consume fragments that have not landed.
#define HAWAIT(n_) else if (younger == (n_)) { if constexpr (TM == 4) asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fa[set][2]), "+v"(fa[set][3])); \
                                                 else if constexpr (TM == 2) asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(fa[set][0]), "+v"(fa[set][1])); \
                                                 else asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(fa[set][0])); }
    auto await = [&](const int set, const int younger) {
        static_assert(TM == 1 || TM == 2 || TM == 4, "fragment waits are spelled for 1, 2 or 4 row tiles");
        static_assert(FD >= 1 && FD * TM <= 12, "wait counts are spelled for 0 .. 12 outstanding reads");
        if (false) {}
        HAWAIT(0) HAWAIT(1) HAWAIT(2) HAWAIT(3) HAWAIT(4) HAWAIT(5) HAWAIT(6) HAWAIT(7) HAWAIT(8) HAWAIT(9) HAWAIT(10) HAWAIT(11) HAWAIT(12)
    };
#undef HAWAIT
    // the residual tile (SGAM_HPEEL): requested where the second-to-last slab's halo load would stand — that load is dead, its place
    // in the in-order queue is free, and a slab and a half of MFMAs cover the trip — instead of at the head of the epilogue, where
    // every workgroup of the launch waits for it at once.  Without a residual (or with split-K, where the combine adds it) the
    // descriptor is empty and the loads return zeros without touching memory.
    constexpr bool PEEL = SGAM_HPEEL && SGAM_HDIRECT && BM != 256;
    constexpr bool RPF = PEEL && (SGAM_HRPF == 2 || (SGAM_HRPF == 1 && TM >= 4));
    u32x2 rq[TM][TN][4];                      // residual: four 4-channel units per (row tile, channel tile)
    auto rload = [&]() {
        const int pl_ = lane & 31, hh_ = lane >> 5;
        const bool live = p.res && p.ksplit == 1;
        const unsigned r_bytes_ = live ? (unsigned)(((int64_t)(p.M - 1) * p.ldr + p.n_valid) * 2) : 0u;
        const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc((void *)p.res, 0, (int)r_bytes_, 0x00020000);
        const int wn0_ = n0 + wn * (BN / WGN_);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int trow = wm * (BM / WGM_) + i * 32 + pl_;
            const int mrow_ = (b * p.Ho + ty0 + (trow >> TWS)) * p.Wo + tx0 + (trow & (TW - 1));
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int n4 = wn0_ + j * 32 + hh_ * 16 + k * 4;
                    rq[i][j][k] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(
                                                                rr_, (int)hsel(n4 < p.n_valid, (unsigned)(mrow_ * p.ldr + n4) * 2u, 0xFFFFFFF0u), 0, 0));
                }
        }
    };
    // one slab: nine taps of TM TN MFMAs per k-step.  RB = ring phase of its tap 0 (NBR = 6: 0, 3, 0, ...).  MODE 0: run-time flags
    // decide whether the next slab / the one behind it exist (dead loads go out of range); the peeled forms know: 1 = two more
    // slabs follow, 2 = one more follows (stages it, then requests the residual instead of a halo), 3 = the last (nothing to
    // stage or to request: a quarter of the staging arithmetic of a four-slab tile used to run on zeros here)
    auto slab = [&](const int sl, auto rb_, auto mode_) {
        constexpr int RB = decltype(rb_)::value, MODE = decltype(mode_)::value;
        const bool has_next = MODE == 0 ? sl + 1 < s1 : MODE != 3;
        const bool has_next2 = MODE == 0 ? sl + 2 < s1 : MODE == 1;
        hb = smem + hcur * HPL;
        if constexpr (!UPS) {
            const unsigned hb_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short *)hb;
#pragma unroll
            for (int i = 0; i < TM; ++i) a_lds[i] = hb_lds + 2u * (unsigned)a_base[i];
        }
        constexpr int HLT = (SGAM_HLT >= NH && SGAM_HLT <= 8) ? SGAM_HLT : NH + 1;       // the staged pieces take taps HLT - NH .. HLT - 1
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int set = (tap + RB) % NBR;
            if constexpr (!(SGAM_HABLATE & 8) && NBR != 9) {
                const int tt = tap + NBR - 1;
                if (tt < 9) bload((tt + RB) % NBR, tt, sl, true);
                else if constexpr (MODE != 3) bload((tt + RB) % NBR, tt - 9, sl + 1, has_next);
            }
            if constexpr (!(SGAM_HABLATE & 16)) {
                // (the barrier keeps the scheduler from hoisting the staging arithmetic to the head of the slab body, in front of
                // this iteration's first loads: the wait it then needs counts loads across the loop's back edge and comes out as
                // vmcnt(0) — the whole vector-memory queue drained at the top of every slab)
                if (SGAM_HSB == 2 || (SGAM_HSB == 1 && tap == 1)) __builtin_amdgcn_sched_barrier(0);
                if constexpr (MODE != 3) {
                    if (tap >= HLT - NH && tap < HLT) hprep_piece(tap - (HLT - NH));        // next slab's halo, one piece per tap
                    if (tap == HLT) {
                        hstore(hcur ^ 1);
                        if constexpr (MODE == 2) {
                            if constexpr (RPF) rload();
                        } else {
                            hload(sl + 2, has_next2);
                        }
                    }
                }
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int q = tap * 2 + kk;
                if constexpr (!UPS) {
                    if (q == 0) {
#pragma unroll
                        for (int d = 0; d < FD; ++d) afrag_asm(d, d >> 1, d & 1);
                    }
                    if (q + FD < 18) afrag_asm((q + FD) % (FD + 1), (q + FD) >> 1, (q + FD) & 1);
                    await(q % (FD + 1), TM * (17 - q < FD ? 17 - q : FD));
                } else {
                    if (q == 0) afrag(0, 0, 0);
                    if (q < 17) afrag((q + 1) % (FD + 1), (q + 1) >> 1, (q + 1) & 1);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (SGAM_HABLATE & 1) acc[i][j][0] += __builtin_bit_cast(float, fa[q % (FD + 1)][i][0] ^ bq[set][j][kk][0]);
                        else if constexpr (SGAM_HDIRECT) acc[i][j] = HH<HT>::mfma(bq[set][j][kk], fa[q % (FD + 1)][i], acc[i][j]);
                        else acc[i][j] = HH<HT>::mfma(fa[q % (FD + 1)][i], bq[set][j][kk], acc[i][j]);
                    }
            }
            if constexpr (!(SGAM_HABLATE & 8) && NBR == 9) bload(tap, tap, sl + 1, has_next);      // this tap's set, next slab
        }
        __syncthreads();
        hcur ^= 1;
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    if constexpr (SGAM_HABLATE & 4) {
    } else if constexpr (!PEEL) {
        for (int sl = s0; sl < s1; sl += SUN) {
            slab(sl, I0{}, I0{});
            if constexpr (SUN == 2) slab(sl + 1, I3{}, I0{});      // (unreachable: the ring of six needs the peeled loop, asserted above)
        }
    } else if constexpr (SUN == 1) {
        int sl = s0;
        for (; sl + 2 < s1; ++sl) slab(sl, I0{}, I1{});
        if (s1 - s0 >= 2) slab(s1 - 2, I0{}, I2{});
        else if constexpr (RPF) rload();
        slab(s1 - 1, I0{}, I3{});
    } else {
        // ring of six: the ring phase of a slab's tap 0 alternates 0, 3, 0, ...  Two shapes of slab loop are spelled — every
        // further (phase, mode) copy of the slab body costs registers (a dispatch for ANY slab count compiled to 250+ VGPRs, or to
        // 168 with 200 - 500 bytes of scratch): the folding form's workgroups walk one or two slabs (host: sgam_conv2d_h16_gn_foldable),
        // every other launch that takes this ring an even number (host: `deep`)
        if constexpr (GNF) {
            if (s1 - s0 >= 2) {
                slab(s0, I0{}, I2{});
                slab(s0 + 1, I3{}, I3{});
            } else {
                slab(s0, I0{}, I3{});
            }
        } else {
            int sl = s0;
            for (; sl + 2 < s1; sl += 2) {
                slab(sl, I0{}, I1{});
                slab(sl + 1, I3{}, I1{});
            }
            slab(s1 - 2, I0{}, I2{});
            slab(s1 - 1, I3{}, I3{});
        }
    }
    __syncthreads();

    if constexpr (SGAM_HABLATE & 2) {          // keep the accumulators alive through one store that never happens
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) keep += acc[i][j][e];
        if (keep == 12345.678f) reinterpret_cast<float *>(p.out)[tid] = keep;
        return;
    }
    const int n_lim = p.n_valid;
    const unsigned osz = p.out_f32 ? 4u : 2u;
    const unsigned o_bytes = (unsigned)(((int64_t)(p.M - 1) * p.ldc + n_lim) * osz);
    const unsigned r_bytes = p.res ? (unsigned)(((int64_t)(p.M - 1) * p.ldr + p.n_valid) * 2) : 0u;
    const unsigned bias_bytes = p.bias ? (unsigned)(p.N * 4) : 0u;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)o_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)p.res, 0, (int)r_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)p.bias, 0, (int)bias_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const int wn0 = n0 + wn * (BN / WGN_);
#if SGAM_HDIRECT
    // ---- epilogue, transposed product: in the 32 x 32 accumulator layout lane (pixel = lane & 31, half = lane >> 5) holds
    // MFMA rows 8 (e / 4) + 4 half + e % 4, e = 0..15; the weight rows were packed in the order that makes those the
    // channels 16 half + e of the 32-channel tile (pack_weight_h16_frag_kernel): SIXTEEN CONSECUTIVE CHANNELS OF ONE PIXEL
    // per lane and accumulator tile — 32 bytes of a 16-bit row (64 of an fp32 one) go out as 16-byte stores, and bias,
    // residual and the GroupNorm statistics of the output are applied in registers: no trip through LDS.
    const int pl = lane & 31, hh = lane >> 5;
    int mrow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int trow = wm * (BM / WGM_) + i * 32 + pl;
        mrow[i] = (b * p.Ho + ty0 + (trow >> TWS)) * p.Wo + tx0 + (trow & (TW - 1));
    }
    if (p.ksplit > 1) {
        // split-K: the raw fp32 partial tile of this K range; bias, residual, rounding and the statistics belong to the
        // combine (h16_splitk_reduce_kernel), which adds the ranges in a fixed order
        float *wsz = p.ws + (int64_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 v = {acc[i][j][4 * k], acc[i][j][4 * k + 1], acc[i][j][4 * k + 2], acc[i][j][4 * k + 3]};
                    *reinterpret_cast<f32x4 *>(wsz + (int64_t)mrow[i] * p.N + wn0 + j * 32 + hh * 16 + k * 4) = v;
                }
        return;
    }
    if constexpr (!RPF) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int n4 = wn0 + j * 32 + hh * 16 + k * 4;
                    rq[i][j][k] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(
                                                                rr, (int)hsel(n4 < n_lim, (unsigned)(mrow[i] * p.ldr + n4) * 2u, OOB), 0, 0));
                }
    }
    float us[RH][TN][4], uss[RH][TN][4];      // per (row half, 4-channel unit): sum, sum of squares over this lane's pixels
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nb = wn0 + j * 32 + hh * 16;
        f32x4 bv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  rb, (int)hsel(nb + 4 * k < n_lim, (unsigned)(nb + 4 * k) * 4u, OOB), 0, 0));
#pragma unroll
            for (int r = 0; r < RH; ++r) us[r][j][k] = uss[r][j][k] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            u32x4 o16[2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int n4 = nb + 4 * k;
                const bool ok = n4 < n_lim;
                const bool st_ok = ok && !(SGAM_HABLATE & 32);        // (32: every store lands out of range and is dropped)
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * k + e] + bv[k][e];
                v[0] += HH<HT>::to_f((unsigned short)(rq[i][j][k][0] & 0xFFFFu));
                v[1] += HH<HT>::to_f((unsigned short)(rq[i][j][k][0] >> 16));
                v[2] += HH<HT>::to_f((unsigned short)(rq[i][j][k][1] & 0xFFFFu));
                v[3] += HH<HT>::to_f((unsigned short)(rq[i][j][k][1] >> 16));
                if (p.out_f32) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro,
                                                           (int)hsel(st_ok, (unsigned)(mrow[i] * p.ldc + n4) * 4u, OOB), 0, 0);
                } else {
                    const unsigned short h0 = HH<HT>::from_f(v[0]), h1 = HH<HT>::from_f(v[1]), h2 = HH<HT>::from_f(v[2]),
                                         h3 = HH<HT>::from_f(v[3]);
                    o16[k >> 1][(k & 1) * 2] = (unsigned)h0 | ((unsigned)h1 << 16);
                    o16[k >> 1][(k & 1) * 2 + 1] = (unsigned)h2 | ((unsigned)h3 << 16);
                    // the statistics describe the STORED (rounded) tensor: that is what the next GroupNorm normalises
                    v = f32x4{HH<HT>::to_f(h0), HH<HT>::to_f(h1), HH<HT>::to_f(h2), HH<HT>::to_f(h3)};
                }
                if (ok) {
                    constexpr int TMH = TM / RH > 0 ? TM / RH : 1;
                    hstat_add(us[RH == 1 ? 0 : i / TMH][j][k], uss[RH == 1 ? 0 : i / TMH][j][k], v);
                }
            }
            if (!p.out_f32) {
                if (nb + 16 <= n_lim) {               // the usual case: two 16-byte stores
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2)
                        __builtin_amdgcn_raw_buffer_store_b128(
                            o16[q2], ro, (int)hsel(!(SGAM_HABLATE & 32), (unsigned)(mrow[i] * p.ldc + nb + 8 * q2) * 2u, OOB), 0, 0);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const u32x2 o8 = {o16[k >> 1][(k & 1) * 2], o16[k >> 1][(k & 1) * 2 + 1]};
                        __builtin_amdgcn_raw_buffer_store_b64(
                            o8, ro, (int)hsel(nb + 4 * k < n_lim && !(SGAM_HABLATE & 32), (unsigned)(mrow[i] * p.ldc + nb + 4 * k) * 2u, OOB),
                            0, 0);
                    }
                }
            }
        }
    }
    if (p.gn_partial) {
        // over the 32 pixels of a lane half (xor shuffles stay inside it), then lane 0 of each half leaves its 4 x TN units
#pragma unroll
        for (int r = 0; r < RH; ++r)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int off = 16; off >= 1; off >>= 1) {
                        us[r][j][k] += __shfl_xor(us[r][j][k], off, 64);
                        uss[r][j][k] += __shfl_xor(uss[r][j][k], off, 64);
                    }
        float *sl = reinterpret_cast<float *>(smem) + wave * (RH * 2 * TN * 8);    // wave-private: [row half][unit = 8 j + 4 half + k][2]
        if (pl == 0) {
#pragma unroll
            for (int r = 0; r < RH; ++r)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        sl[((r * TN + j) * 8 + hh * 4 + k) * 2] = us[r][j][k];
                        sl[((r * TN + j) * 8 + hh * 4 + k) * 2 + 1] = uss[r][j][k];
                    }
        }
        const int c4_per_group = p.gn_cpg / 4;
        const int groups_here = (TN * 8) / c4_per_group;
        if (lane < groups_here * RH) {
            const int r = lane / groups_here, gl = lane - r * groups_here;
            double ds = 0.0, dss = 0.0;
            for (int k = 0; k < c4_per_group; ++k) {
                ds += (double)sl[(r * TN * 8 + gl * c4_per_group + k) * 2];
                dss += (double)sl[(r * TN * 8 + gl * c4_per_group + k) * 2 + 1];
            }
            const int g = (wn0 / p.gn_cpg) + gl;
            const int groups = p.N / p.gn_cpg;
            if (g < groups) {
                // chunk = (tile, row half): the wavefronts that share a row half own different channels, so each
                // (chunk = tile * 2 + row half, group) is written by exactly one lane of one wavefront
                const int chunks_per_b = tiles_img * 2;
                double *o = p.gn_partial + (((int64_t)b * chunks_per_b + t_img * 2 + (RH == 1 ? wm : r)) * groups + g) * 2;
                o[0] = ds;
                o[1] = dss;
            }
        }
    }
#else
    // ---- epilogue: wave-private LDS transpose, then every lane owns 4 consecutive channels of one pixel
    float *region = reinterpret_cast<float *>(smem) + wave * (WM * LDR);
    const int col_l = lane & 31, row_h = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = wn0 + j * 32 + col_l;
            const float bias_n = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                              rb, (int)hsel(n < n_lim, (unsigned)n * 4u, OOB), 0, 0));
#pragma unroll
            for (int e = 0; e < 16; ++e)
                region[(i * 32 + (e & 3) + 8 * (e >> 2) + row_h) * LDR + j * 32 + col_l] = acc[i][j][e] + bias_n;
        }
    constexpr int C4 = WN / 4;          // float4 chunks per row: 16
    constexpr int RPP = 64 / C4;        // rows per pass: 4
    const int c4 = lane % C4, rr0 = lane / C4;
    const int n4 = wn0 + c4 * 4;
    const bool n_ok = n4 < n_lim;
    float gs = 0.f, gss = 0.f;
#pragma unroll
    for (int pass = 0; pass < WM / RPP; ++pass) {
        const int row = rr0 + pass * RPP;
        const int trow = wm * (BM / 2) + row;
        const int m = (b * p.Ho + ty0 + (trow >> TWS)) * p.Wo + tx0 + (trow & (TW - 1));
        const bool ok = n_ok && m < p.M;
        const bool st_ok = ok && !(SGAM_HABLATE & 32);        // (32: every store lands out of range and is dropped)
        f32x4 v = *reinterpret_cast<const f32x4 *>(region + row * LDR + c4 * 4);
        const u32x2 rq = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(
                                                       rr, (int)hsel(ok, (unsigned)(m * p.ldr + n4) * 2u, OOB), 0, 0));
        v[0] += HH<HT>::to_f((unsigned short)(rq[0] & 0xFFFFu));
        v[1] += HH<HT>::to_f((unsigned short)(rq[0] >> 16));
        v[2] += HH<HT>::to_f((unsigned short)(rq[1] & 0xFFFFu));
        v[3] += HH<HT>::to_f((unsigned short)(rq[1] >> 16));
        if (p.out_f32) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, (int)hsel(st_ok, (unsigned)(m * p.ldc + n4) * 4u, OOB),
                                                   0, 0);
        } else {
            u32x2 o;
            unsigned short h0 = HH<HT>::from_f(v[0]), h1 = HH<HT>::from_f(v[1]), h2 = HH<HT>::from_f(v[2]), h3 = HH<HT>::from_f(v[3]);
            o[0] = (unsigned)h0 | ((unsigned)h1 << 16);
            o[1] = (unsigned)h2 | ((unsigned)h3 << 16);
            __builtin_amdgcn_raw_buffer_store_b64(o, ro, (int)hsel(st_ok, (unsigned)(m * p.ldc + n4) * 2u, OOB), 0, 0);
            // the statistics describe the STORED (rounded) tensor: that is what the next GroupNorm normalises
            v = f32x4{HH<HT>::to_f(h0), HH<HT>::to_f(h1), HH<HT>::to_f(h2), HH<HT>::to_f(h3)};
        }
        if (ok) {
            gs += (v[0] + v[1]) + (v[2] + v[3]);
            gss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
    }
    if (p.gn_partial) {
        float *sl = region;                   // wave-private: [64 lanes][2]
        sl[lane * 2] = gs;
        sl[lane * 2 + 1] = gss;
        const int c4_per_group = p.gn_cpg / 4;
        const int groups_here = C4 / c4_per_group;
        if (lane < groups_here) {
            double ds = 0.0, dss = 0.0;
            for (int r = 0; r < RPP; ++r)
                for (int k = 0; k < c4_per_group; ++k) {
                    const int l = r * C4 + lane * c4_per_group + k;
                    ds += (double)sl[l * 2];
                    dss += (double)sl[l * 2 + 1];
                }
            const int g = (wn0 / p.gn_cpg) + lane;
            const int groups = p.N / p.gn_cpg;
            if (g < groups) {
                // chunk = (tile, row half, column half): two wavefronts share a row half but own different channels, so
                // each (chunk = tile * 2 + wm, group) is written by exactly one lane of one wavefront
                const int chunks_per_b = tiles_img * 2;
                double *o = p.gn_partial + (((int64_t)b * chunks_per_b + t_img * 2 + wm) * groups + g) * 2;
                o[0] = ds;
                o[1] = dss;
            }
        }
    }
#endif
}

// split-K combine: out = round16(sum_z ws[z] + bias + residual), ranges added in the order z = 0, 1, 2, ...; thread = 4
// channels of one pixel, workgroup = 1024 outputs = 1024 / N whole rows = one chunk of output statistics (the scheme of
// splitk_reduce_f32x_kernel, conv_f32x.hip)
template <int HT>
__global__ __launch_bounds__(256) void h16_splitk_reduce_kernel(const HHParams p) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = p.N / 4;
    const bool live = q < (int64_t)p.M * nq;
    const int m = live ? (int)(q / nq) : 0;
    const int n = live ? (int)(q - (int64_t)m * nq) * 4 : 0;
    float gs = 0.f, gss = 0.f;
    if (live && n < p.n_valid) {
        const float *w0 = p.ws + (int64_t)m * p.N + n;
        const int64_t zs = (int64_t)p.M * p.N;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *reinterpret_cast<const f32x4 *>(p.bias + n);
        u32x2 rq = {0u, 0u};
        if (p.res) rq = *reinterpret_cast<const u32x2 *>(p.res + (int64_t)m * p.ldr + n);
        f32x4 sum = *reinterpret_cast<const f32x4 *>(w0);
        int z = 1;
        for (; z + 4 <= p.ksplit; z += 4) {              // four loads in flight (a plain loop is load -> wait -> add)
            const f32x4 a = *reinterpret_cast<const f32x4 *>(w0 + z * zs), b2 = *reinterpret_cast<const f32x4 *>(w0 + (z + 1) * zs);
            const f32x4 c = *reinterpret_cast<const f32x4 *>(w0 + (z + 2) * zs), d = *reinterpret_cast<const f32x4 *>(w0 + (z + 3) * zs);
            sum += a;
            sum += b2;
            sum += c;
            sum += d;
        }
        for (; z < p.ksplit; ++z) sum += *reinterpret_cast<const f32x4 *>(w0 + z * zs);
        f32x4 v = sum + bv;
        if (p.res) {
            v[0] += HH<HT>::to_f((unsigned short)(rq[0] & 0xFFFFu));
            v[1] += HH<HT>::to_f((unsigned short)(rq[0] >> 16));
            v[2] += HH<HT>::to_f((unsigned short)(rq[1] & 0xFFFFu));
            v[3] += HH<HT>::to_f((unsigned short)(rq[1] >> 16));
        }
        if (p.out_f32) {
            *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(p.out) + (int64_t)m * p.ldc + n) = v;
        } else {
            const unsigned short h0 = HH<HT>::from_f(v[0]), h1 = HH<HT>::from_f(v[1]), h2 = HH<HT>::from_f(v[2]), h3 = HH<HT>::from_f(v[3]);
            const u32x2 o = {(unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16)};
            *reinterpret_cast<u32x2 *>(reinterpret_cast<unsigned short *>(p.out) + (int64_t)m * p.ldc + n) = o;
            v = f32x4{HH<HT>::to_f(h0), HH<HT>::to_f(h1), HH<HT>::to_f(h2), HH<HT>::to_f(h3)};     // statistics of the STORED tensor
        }
        gs = (v[0] + v[1]) + (v[2] + v[3]);
        gss = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    if (!p.gn_partial) return;
    // host guarantees: n_valid == N, 1024 % N == 0, whole workgroups inside one image
    const int c4n = p.gn_cpg / 4;                      // lanes per (row, group): 1, 2, 4 or 8 neighbours
    for (int o = 1; o < c4n; o <<= 1) {
        gs += __shfl_xor(gs, o, 64);
        gss += __shfl_xor(gss, o, 64);
    }
    __shared__ float sh[8][32][2];                     // [row in workgroup][group]
    const int rows = 1024 / p.N, row = threadIdx.x / nq, g = (threadIdx.x - row * nq) / c4n;
    if ((threadIdx.x % c4n) == 0) {
        sh[row][g][0] = gs;
        sh[row][g][1] = gss;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        double ds = 0.0, dss = 0.0;
        for (int r = 0; r < rows; ++r) {
            ds += (double)sh[r][threadIdx.x][0];
            dss += (double)sh[r][threadIdx.x][1];
        }
        double *o = p.gn_partial + ((int64_t)blockIdx.x * 32 + threadIdx.x) * 2;     // chunk = workgroup (image-major)
        o[0] = ds;
        o[1] = dss;
    }
}

// the combine of the SMALL maps (16^2, 32^2), group-major: a workgroup owns TR = 1024 / TC rows x TC channels (whole 64- / 32-
// byte row pieces), so an image falls into hw / TR <= 16 chunks of statistics — few enough for the consuming conv to fold by
// itself (GNF), which removes the fold launch between two convolutions (conv_f32x.hip: splitk_reduce_gm_f32x_kernel).
template <int HT, int TC>
__global__ __launch_bounds__(256) void h16_splitk_reduce_gm_kernel(const HHParams p) {
    constexpr int TR = 1024 / TC, TPR = TC / 4;
    const int col_tiles = p.N / TC;
    const int rt = blockIdx.x / col_tiles, ct = blockIdx.x - rt * col_tiles;
    const int row = threadIdx.x / TPR, c4 = threadIdx.x - row * TPR;
    const int m = rt * TR + row, n = ct * TC + c4 * 4;            // host: M % TR == 0, N % TC == 0, n_valid == N
    const float *w0 = p.ws + (int64_t)m * p.N + n;
    const int64_t zs = (int64_t)p.M * p.N;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *reinterpret_cast<const f32x4 *>(p.bias + n);
    u32x2 rq = {0u, 0u};
    if (p.res) rq = *reinterpret_cast<const u32x2 *>(p.res + (int64_t)m * p.ldr + n);
    f32x4 sum = *reinterpret_cast<const f32x4 *>(w0);
    int z = 1;
    for (; z + 4 <= p.ksplit; z += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(w0 + z * zs), b2 = *reinterpret_cast<const f32x4 *>(w0 + (z + 1) * zs);
        const f32x4 c = *reinterpret_cast<const f32x4 *>(w0 + (z + 2) * zs), d = *reinterpret_cast<const f32x4 *>(w0 + (z + 3) * zs);
        sum += a;
        sum += b2;
        sum += c;
        sum += d;
    }
    for (; z < p.ksplit; ++z) sum += *reinterpret_cast<const f32x4 *>(w0 + z * zs);
    f32x4 v = sum + bv;
    if (p.res) {
        v[0] += HH<HT>::to_f((unsigned short)(rq[0] & 0xFFFFu));
        v[1] += HH<HT>::to_f((unsigned short)(rq[0] >> 16));
        v[2] += HH<HT>::to_f((unsigned short)(rq[1] & 0xFFFFu));
        v[3] += HH<HT>::to_f((unsigned short)(rq[1] >> 16));
    }
    if (p.out_f32) {
        *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(p.out) + (int64_t)m * p.ldc + n) = v;
    } else {
        const unsigned short h0 = HH<HT>::from_f(v[0]), h1 = HH<HT>::from_f(v[1]), h2 = HH<HT>::from_f(v[2]), h3 = HH<HT>::from_f(v[3]);
        const u32x2 o = {(unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16)};
        *reinterpret_cast<u32x2 *>(reinterpret_cast<unsigned short *>(p.out) + (int64_t)m * p.ldc + n) = o;
        v = f32x4{HH<HT>::to_f(h0), HH<HT>::to_f(h1), HH<HT>::to_f(h2), HH<HT>::to_f(h3)};     // statistics of the STORED tensor
    }
    float gs = (v[0] + v[1]) + (v[2] + v[3]);
    float gss = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    if (!p.gn_partial) return;
    const int lpg = p.gn_cpg / 4;                       // lanes per (row, group): 1, 2, 4 or 8 neighbours (cpg <= TC)
    for (int o = 1; o < lpg; o <<= 1) {
        gs += __shfl_xor(gs, o, 64);
        gss += __shfl_xor(gss, o, 64);
    }
    __shared__ float sh[256][2];                        // [row][group in tile]: TR * (TC / cpg) = 1024 / cpg <= 256 entries
    const int gt = TC / p.gn_cpg, gl = c4 / lpg;
    if ((c4 % lpg) == 0) {
        sh[row * gt + gl][0] = gs;
        sh[row * gt + gl][1] = gss;
    }
    __syncthreads();
    if ((int)threadIdx.x < gt) {
        double ds = 0.0, dss = 0.0;
        for (int r = 0; r < TR; ++r) {
            ds += (double)sh[r * gt + threadIdx.x][0];
            dss += (double)sh[r * gt + threadIdx.x][1];
        }
        // chunk = row tile inside the image (image-major: rt counts over the whole batch, hw % TR == 0)
        double *o = p.gn_partial + ((int64_t)rt * 32 + ct * gt + threadIdx.x) * 2;
        o[0] = ds;
        o[1] = dss;
    }
}

template <int HT>
__global__ void pack_weight_h16_frag_kernel(const float *w, unsigned short *o, int Cout, int Cin, int KH, int KW, int Cout_pad,
                                            int Cin_pad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int taps = KH * KW;
    const int64_t total = (int64_t)Cout_pad * taps * Cin_pad;
    if (i >= total) return;
    const int c = (int)(i % Cin_pad);
    const int t = (int)((i / Cin_pad) % taps);
    const int n = (int)(i / ((int64_t)Cin_pad * taps));
    float v = 0.f;
    if (n < Cout && c < Cin) v = w[((int64_t)n * Cin + c) * taps + t];
    const int64_t k = (int64_t)t * Cin_pad + c, slabs = (int64_t)taps * Cin_pad / 32;
    const int64_t slab = k >> 5, kin = k & 31;
    // row of the 32-row tile: plain (pixels = MFMA rows) or, for the transposed product, the MFMA row whose accumulator slot
    // makes channel c = 16 half + e slot e of lane half `half`: row = 8 (e / 4) + 4 half + e % 4
    const int c32 = n & 31;
    const int r32 = SGAM_HDIRECT ? 8 * ((c32 & 15) >> 2) + 4 * (c32 >> 4) + (c32 & 3) : c32;
    const int64_t piece = (((kin >> 4) * 2) + ((kin >> 3) & 1)) * 32 + r32;
    o[((((int64_t)(n >> 5)) * slabs + slab) * 128 + piece) * 8 + (kin & 7)] = HH<HT>::from_f(v);
}

bool hh_shape(const sgam_conv_desc *d, int bm) {
    const int up = d->upsample2x ? 2 : 1;
    const bool tile_ok = (bm == 256 && d->Wo % 16 == 0 && d->Ho % 16 == 0) || (bm == 128 && d->Wo % 16 == 0) || (bm == 64 && d->Wo % 8 == 0);
    return tile_ok && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 && d->Ho == up * d->Hi &&
           d->Wo == up * d->Wi && d->Ho % 8 == 0 && d->Cin % 32 == 0 && d->N % 128 == 0 && d->lda % 8 == 0 && d->ldb % 32 == 0 &&
           d->ldb >= 9 * d->Cin && d->n_valid % 4 == 0 && d->ldc % 4 == 0 && d->ldr % 4 == 0 && d->bias_per_row == 0;
}

// plan for this descriptor: tile rows (128 when that still fills the chip or the caller's plan asks for it, else 64) and
// the split of the K slabs over grid.y for maps too small to fill the chip with whole-K workgroups; bm = 0: not a halo shape
struct HHPlan {
    int bm = 0, ksplit = 1, slabs_per_split = 0;
};

HHPlan hh_plan(const sgam_conv_desc *d) {
    HHPlan pl;
    if (!d || d->B <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->N <= 0 || d->Cin <= 0) return pl;
    const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
    int bm = (M / 128) * (d->N / 128) >= 224 ? 128 : 64;
    if (d->plan_bm == 256 || d->plan_bm == 128 || d->plan_bm == 64) bm = d->plan_bm;
    if (bm == 256 && !hh_shape(d, 256)) bm = 128;
    if (bm == 128 && !hh_shape(d, 128)) bm = 64;
    if (!hh_shape(d, bm)) return pl;
    const int slabs = d->Cin / 32;
    const int64_t tiles = (M / bm) * (d->N / 128);
    int ks = 1;
    if (SGAM_HDIRECT) {
        if (d->plan_ksplit > 0) ks = d->plan_ksplit;
        else if (tiles < 128) ks = (int)((256 + tiles - 1) / tiles);
        if (ks > slabs) ks = slabs;
        if (ks > 32) ks = 32;
    } else if (d->plan_ksplit > 1 || tiles < 128) {
        return pl;                                  // (the LDS-transpose build has no split-K epilogue)
    }
    pl.bm = bm;
    pl.slabs_per_split = (slabs + ks - 1) / ks;
    pl.ksplit = (slabs + pl.slabs_per_split - 1) / pl.slabs_per_split;
    return pl;
}

int hh_bm(const sgam_conv_desc *d) { return hh_plan(d).bm; }

}  // namespace

extern "C" int32_t sgam_conv2d_h16_uses_halo(const sgam_conv_desc *d) { return hh_bm(d) ? 1 : 0; }

// bytes of fp32 split-K partials sgam_conv2d_halo_nhwc_h16 needs for this descriptor (0: whole-K workgroups), -1: not a
// halo shape
extern "C" int64_t sgam_conv2d_halo_h16_workspace_bytes(const sgam_conv_desc *d) {
    const HHPlan pl = hh_plan(d);
    if (!pl.bm) return -1;
    return pl.ksplit > 1 ? (int64_t)pl.ksplit * d->B * d->Ho * d->Wo * d->N * 4 : 0;
}

// chunks of output statistics per image the halo kernel leaves (sgam_conv2d_halo_nhwc_h16 with gn_partial), 0 = none:
// one per (tile, wavefront row) from the epilogue, one per 1024 outputs (1024 / N whole rows) from the split-K combine
// channels per workgroup tile of the group-major combine for this descriptor (32, 16 or 8), 0 = the row-major combine: a
// group (N / 32 channels) must fit a tile and an image must fall into at most 16 row tiles (conv_f32x.hip: red_tc_for)
static int hh_red_tc_for(const sgam_conv_desc *d) {
    static const int on = [] { const char *e = getenv("SGAM_GN_FOLD"); return (e && e[0] == '0') ? 0 : 1; }();
    const int hw = d->Ho * d->Wo, cpg = d->N / 32;
    if (!on || d->N % 128 != 0 || d->n_valid != d->N || d->N > 1024) return 0;
    for (int tc = 32; tc >= 8; tc >>= 1) {
        const int tr = 1024 / tc;
        // the fold inside the combine is an xor butterfly over cpg / 4 lanes of whole groups: cpg must divide the tile and be a power of two
        // (N = 384, 768 ... — a ch_mult with a 3 — keep the row-major combine and the two-pass statistics)
        if (tc >= cpg && tc % cpg == 0 && (cpg & (cpg - 1)) == 0 && hw % tr == 0 && hw / tr <= 16) return tc;
    }
    return 0;
}

extern "C" int32_t sgam_conv2d_h16_stats_chunks(const sgam_conv_desc *d) {
    const HHPlan pl = hh_plan(d);
    if (!pl.bm || d->n_valid != d->N) return 0;
    const int hw = d->Ho * d->Wo;
    if (pl.ksplit == 1) return (hw / pl.bm) * 2;
    if (const int tc = hh_red_tc_for(d)) return hw / (1024 / tc);
    if (d->N > 1024 || 1024 % d->N != 0 || ((int64_t)hw * d->N) % 1024 != 0) return 0;
    return (int32_t)((int64_t)hw * d->N / 1024);
}

extern "C" int sgam_pack_conv_weight_h16_frag(const float *w_oihw, void *w_frag, int32_t ht, int32_t Cout, int32_t Cin, int32_t KH,
                                              int32_t KW, int32_t Cout_pad, int32_t Cin_pad, void *stream) {
    if (!w_oihw || !w_frag || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || Cout_pad < Cout || Cin_pad < Cin || Cin_pad % 32 ||
        Cout_pad % 32 || (ht != 0 && ht != 1))
        return SGAM_EINVAL;
    const int64_t total = (int64_t)Cout_pad * KH * KW * Cin_pad;
    const dim3 g(sgam_cdiv(total, 256));
    if (ht == 0)
        SGAM_KLAUNCH(pack_weight_h16_frag_kernel<0>, g, dim3(256), 0, sgam_stream(stream), w_oihw, (unsigned short *)w_frag, Cout,
                     Cin, KH, KW, Cout_pad, Cin_pad);
    else
        SGAM_KLAUNCH(pack_weight_h16_frag_kernel<1>, g, dim3(256), 0, sgam_stream(stream), w_oihw, (unsigned short *)w_frag, Cout,
                     Cin, KH, KW, Cout_pad, Cin_pad);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// 1 when sgam_conv2d_halo_gnp_nhwc_h16 can take the statistics of its input as `chunks_in` chunk partials per image and fold
// them itself: the 64-row tile of the halo kernel, a workgroup that walks at most two channel slabs (it repeats the fold per
// slab), groups of >= 8 channels (a thread's eight staged channels are one group), at most 16 chunks
extern "C" int32_t sgam_conv2d_h16_gn_foldable(const sgam_conv_desc *d, int32_t chunks_in) {
    static const int on = [] { const char *e = getenv("SGAM_GN_FOLD"); return (e && e[0] == '0') ? 0 : 1; }();
    if (!on || !d || chunks_in < 1 || chunks_in > 16 || d->upsample2x || d->Cin % 256 != 0) return 0;
    const HHPlan pl = hh_plan(d);
    return (pl.bm == 64 && pl.slabs_per_split <= 2) ? 1 : 0;
}

static int hh_conv_impl(const sgam_conv_desc *d, int32_t ht, const void *x, const float *gn_mean_rstd, const double *gn_partial_in,
                        int32_t chunks_in, float gn_eps, const float *gn_gamma, const float *gn_beta, int32_t gn_swish,
                        const void *w_frag, const float *bias, const void *residual, void *out, int32_t out_f32, double *gn_partial,
                        void *workspace, int64_t workspace_bytes, void *stream) {
    const HHPlan pl = hh_plan(d);
    const int bm = pl.bm;
    if (!bm || !x || !w_frag || !out || (ht != 0 && ht != 1)) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(w_frag) || (((uintptr_t)out) & 7u) || (residual && (((uintptr_t)residual) & 7u)))
        return SGAM_EALIGN;
    if (gn_partial_in && (gn_mean_rstd || sgam_conv2d_h16_gn_foldable(d, chunks_in) != 1 || !sgam_aligned16(gn_partial_in)))
        return SGAM_EINVAL;
    const bool gn = gn_mean_rstd != nullptr || gn_partial_in != nullptr;
    if (gn && (!gn_gamma || !gn_beta || !sgam_aligned16(gn_gamma) || !sgam_aligned16(gn_beta) || d->upsample2x || d->Cin % 128 ||
               d->Cin > SGAM_HGN_MAXC))
        return SGAM_EINVAL;
    if (gn_partial && sgam_conv2d_h16_stats_chunks(d) <= 0) return SGAM_EINVAL;
    if (pl.ksplit > 1 && (!workspace || workspace_bytes < sgam_conv2d_halo_h16_workspace_bytes(d) || !sgam_aligned16(workspace) ||
                          (bias && !sgam_aligned16(bias))))
        return SGAM_EINVAL;
    HHParams p;
    p.x = (const unsigned short *)x; p.w = (const unsigned short *)w_frag; p.res = (const unsigned short *)residual;
    p.bias = bias; p.out = out;
    p.B = d->B; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.N = d->N; p.ups = d->upsample2x ? 1 : 0;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr; p.n_valid = d->n_valid; p.out_f32 = out_f32 ? 1 : 0;
    p.M = d->B * d->Ho * d->Wo;
    p.slabs = d->Cin / 32;
    p.ksplit = pl.ksplit; p.slabs_per_split = pl.slabs_per_split; p.ws = (float *)workspace;
    const int64_t xb = (((int64_t)d->B * d->Hi * d->Wi - 1) * d->lda + d->Cin) * 2;
    const int64_t wb = (int64_t)d->N * d->ldb * 2;
    if (xb >= (1ll << 32) - 64 || wb >= (1ll << 32) - 256) return SGAM_EINVAL;
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
    p.gn_stats = gn_mean_rstd; p.gn_gamma = gn_gamma; p.gn_beta = gn_beta; p.gn_swish = gn_swish ? 1 : 0;
    p.gn_partial = gn_partial; p.gn_cpg = d->N / 32;
    p.gn_partial_in = gn_partial_in; p.gn_chunks_in = chunks_in; p.gn_eps = gn_eps;
    p.gn_inv_n = 1.0f / ((float)d->Hi * (float)d->Wi * (float)(d->Cin / 32));
    p.gx = p.M / bm; p.gy = d->N / 128;
    static const int swz = [] { const char *e = getenv("SGAM_XCD_SWIZZLE"); return (e && e[0] == '0') ? 0 : 1; }();
    p.xcd_swizzle = swz;
    const dim3 grid((unsigned)((int64_t)p.gx * p.gy), (unsigned)pl.ksplit);
    hipStream_t s = sgam_stream(stream);
    if (sgam_i_prof_on) sgam_i_prof_shape(p.M, d->n_valid, 9 * d->Cin, 1);
    if (sgam_i_prof_on)
        sgam_i_prof_work(2.0 * p.M * d->n_valid * (double)(9 * d->Cin),
                         2.0 * ((double)d->B * d->Hi * d->Wi * d->Cin + (double)d->n_valid * 9 * d->Cin + (double)p.M * d->n_valid));
    // ring depth of this launch: the ring of six is spelled for even slab counts per workgroup (and for the one or two of the folding form)
    const bool even = (pl.slabs_per_split % 2 == 0) && (p.slabs % pl.slabs_per_split == 0);
    const bool deep = even && ((bm == 128 && SGAM_HNBR == 6) || (bm == 64 && SGAM_HNBR64 == 6));
    // (launch sites spell the template arguments the way the kernel timeline / bench.py name the instantiations: the defaulted
    // tail — SW = true, GNF = false, NB = 3 — is left off)
#define HH_LAUNCH(BM_, HT_)                                                                                                      \
    do {                                                                                                                         \
        if (p.ups) SGAM_KLAUNCH((conv3x3_h16_halo_kernel<BM_, 128, HT_, false, true>), grid, dim3(256), 0, s, p);                \
        else if (gn && p.gn_swish) SGAM_KLAUNCH((conv3x3_h16_halo_kernel<BM_, 128, HT_, true, false>), grid, dim3(256), 0, s, p); \
        else if (gn) SGAM_KLAUNCH((conv3x3_h16_halo_kernel<BM_, 128, HT_, true, false, false>), grid, dim3(256), 0, s, p);        \
        else SGAM_KLAUNCH((conv3x3_h16_halo_kernel<BM_, 128, HT_, false, false>), grid, dim3(256), 0, s, p);                     \
    } while (0)
#define HH_LAUNCH6(BM_, HT_)                                                                                                                \
    do {                                                                                                                                    \
        if (p.ups) SGAM_KLAUNCH((conv3x3_h16_halo_kernel<BM_, 128, HT_, false, true, true, false, 6>), grid, dim3(256), 0, s, p);           \
        else if (gn && p.gn_swish) SGAM_KLAUNCH((conv3x3_h16_halo_kernel<BM_, 128, HT_, true, false, true, false, 6>), grid, dim3(256), 0, s, p); \
        else if (gn) SGAM_KLAUNCH((conv3x3_h16_halo_kernel<BM_, 128, HT_, true, false, false, false, 6>), grid, dim3(256), 0, s, p);        \
        else SGAM_KLAUNCH((conv3x3_h16_halo_kernel<BM_, 128, HT_, false, false, true, false, 6>), grid, dim3(256), 0, s, p);                \
    } while (0)
    if (gn_partial_in) {
#if SGAM_HNBRF == 6
        if (ht == 0 && p.gn_swish) SGAM_KLAUNCH((conv3x3_h16_halo_kernel<64, 128, 0, true, false, true, true, 6>), grid, dim3(256), 0, s, p);
        else if (ht == 0) SGAM_KLAUNCH((conv3x3_h16_halo_kernel<64, 128, 0, true, false, false, true, 6>), grid, dim3(256), 0, s, p);
        else if (p.gn_swish) SGAM_KLAUNCH((conv3x3_h16_halo_kernel<64, 128, 1, true, false, true, true, 6>), grid, dim3(256), 0, s, p);
        else SGAM_KLAUNCH((conv3x3_h16_halo_kernel<64, 128, 1, true, false, false, true, 6>), grid, dim3(256), 0, s, p);
#else
        if (ht == 0 && p.gn_swish) SGAM_KLAUNCH((conv3x3_h16_halo_kernel<64, 128, 0, true, false, true, true>), grid, dim3(256), 0, s, p);
        else if (ht == 0) SGAM_KLAUNCH((conv3x3_h16_halo_kernel<64, 128, 0, true, false, false, true>), grid, dim3(256), 0, s, p);
        else if (p.gn_swish) SGAM_KLAUNCH((conv3x3_h16_halo_kernel<64, 128, 1, true, false, true, true>), grid, dim3(256), 0, s, p);
        else SGAM_KLAUNCH((conv3x3_h16_halo_kernel<64, 128, 1, true, false, false, true>), grid, dim3(256), 0, s, p);
#endif
    } else if (bm == 256) {
        if (ht == 0) HH_LAUNCH(256, 0); else HH_LAUNCH(256, 1);
    } else if (bm == 128) {
#if SGAM_HNBR == 6
        if (deep) { if (ht == 0) HH_LAUNCH6(128, 0); else HH_LAUNCH6(128, 1); } else
#endif
        if (ht == 0) HH_LAUNCH(128, 0); else HH_LAUNCH(128, 1);
    } else {
#if SGAM_HNBR64 == 6
        if (deep) { if (ht == 0) HH_LAUNCH6(64, 0); else HH_LAUNCH6(64, 1); } else
#endif
        if (ht == 0) HH_LAUNCH(64, 0); else HH_LAUNCH(64, 1);
    }
    (void)deep;
#undef HH_LAUNCH
#undef HH_LAUNCH6
    SGAM_LAUNCH_CHECK();
    if (pl.ksplit > 1) {
        const int64_t q = (int64_t)p.M * (d->N / 4);
        if (sgam_i_prof_on) sgam_i_prof_work(0.0, (double)pl.ksplit * p.M * d->N * 4.0 + 2.0 * p.M * d->n_valid);
        const int tc = gn_partial ? hh_red_tc_for(d) : 0;
#define HH_RED(HT_)                                                                                                       \
    do {                                                                                                                  \
        if (tc == 32) SGAM_KLAUNCH((h16_splitk_reduce_gm_kernel<HT_, 32>), dim3((unsigned)(q / 256)), dim3(256), 0, s, p);       \
        else if (tc == 16) SGAM_KLAUNCH((h16_splitk_reduce_gm_kernel<HT_, 16>), dim3((unsigned)(q / 256)), dim3(256), 0, s, p);  \
        else if (tc == 8) SGAM_KLAUNCH((h16_splitk_reduce_gm_kernel<HT_, 8>), dim3((unsigned)(q / 256)), dim3(256), 0, s, p);    \
        else SGAM_KLAUNCH(h16_splitk_reduce_kernel<HT_>, dim3(sgam_cdiv(q, 256)), dim3(256), 0, s, p);                    \
    } while (0)
        if (ht == 0) HH_RED(0); else HH_RED(1);
#undef HH_RED
        SGAM_LAUNCH_CHECK();
    }
    return SGAM_OK;
}

extern "C" int sgam_conv2d_halo_nhwc_h16(const sgam_conv_desc *d, int32_t ht, const void *x, const float *gn_mean_rstd,
                                         const float *gn_gamma, const float *gn_beta, int32_t gn_swish, const void *w_frag,
                                         const float *bias, const void *residual, void *out, int32_t out_f32, double *gn_partial,
                                         void *workspace, int64_t workspace_bytes, void *stream) {
    return hh_conv_impl(d, ht, x, gn_mean_rstd, nullptr, 0, 0.f, gn_gamma, gn_beta, gn_swish, w_frag, bias, residual, out, out_f32,
                        gn_partial, workspace, workspace_bytes, stream);
}

// sgam_conv2d_halo_nhwc_h16 with the statistics of x still as its producer's chunk partials [B][chunks_in][32][2] (fp64 {sum,
// sumsq}): the kernel folds them (needs sgam_conv2d_h16_gn_foldable(d, chunks_in) == 1)
extern "C" int sgam_conv2d_halo_gnp_nhwc_h16(const sgam_conv_desc *d, int32_t ht, const void *x, const double *gn_partial_in,
                                             int32_t chunks_in, float gn_eps, const float *gn_gamma, const float *gn_beta,
                                             int32_t gn_swish, const void *w_frag, const float *bias, const void *residual, void *out,
                                             int32_t out_f32, double *gn_partial, void *workspace, int64_t workspace_bytes,
                                             void *stream) {
    if (!gn_partial_in) return SGAM_EINVAL;
    return hh_conv_impl(d, ht, x, nullptr, gn_partial_in, chunks_in, gn_eps, gn_gamma, gn_beta, gn_swish, w_frag, bias, residual, out,
                        out_f32, gn_partial, workspace, workspace_bytes, stream);
}
