// conv_f32x.hip — fp32 convolution / GEMM on the fp16 matrix cores with fp32-class accuracy ("split" mode).
//
// The fp32-in MFMA of conv_gemm.hip runs at 1/16 of the 16-bit matrix rate and that kernel is pinned at ~75 % of
// it.  Here every fp32 operand is split exactly into two fp16 pieces,
//        a = a_hi + a_lo,   a_hi = fp16(a) (RNE),  a_lo = fp16(a - a_hi)        (22 significant bits)
// and the product is evaluated as three fp16 MFMAs with fp32 accumulation,
//        a.b  ~=  a_hi.b_hi + a_hi.b_lo + a_lo.b_hi                              (dropped a_lo.b_lo <= 2^-22 |a||b|)
// Products of fp16 values are exact in fp32, so the only errors are the 2^-22-class representation residue and the
// usual fp32 accumulation round-off: the result is as accurate as an fp32 GEMM with a different summation order
// (and is held to the same acceptance tests: bit-exact codebook indices on the margin-guarded fixtures, RGB-D within
// 1e-4 of the reference) at 3/16 of the fp32-MFMA issue time.  Three kernels share the arithmetic: the generic
// implicit-GEMM kernel (any conv / GEMM shape; A and B tiles through LDS per (tap, slab)), and the halo-staged 3x3
// kernel with its nearest-2x upsampling form (the bulk of the FLOPs; A halo staged once per channel slab, weight
// fragments straight from L2 to registers).
//
//   * weights are split OFFLINE (sgam_pack_conv_weight_f32x) after multiplying by a power of two `w_scale` that
//     lifts max|w| to (512, 1024]: both pieces stay in fp16's normal range; the accumulator is multiplied by the
//     exact inverse in the epilogue;
//   * activations are split while they are staged into LDS (the A operand arrives as fp32 NHWC exactly as in
//     conv_gemm.hip: same descriptor, same addressing, same bounds-checked buffer loads).  |x| must stay below
//     65504 (GroupNorm keeps the VQGAN's activations within a few hundred); pieces below fp16's normal range lose
//     relative, not absolute, accuracy (<= 3e-8 per element).
//   * weights live in HBM in MFMA-fragment order (frag_index): the B operand of one MFMA is one contiguous kilobyte;
//   * generic kernel LDS: per K slab of 32, four fp16 planes A_hi, A_lo [BM][40], B_hi, B_lo [BN][40] (80-byte rows:
//     16-byte aligned and conflict-free for the 16-lane ds_read_b128 groups), double buffered = 80 KiB for 128x128
//     (two workgroups per CU).  One ds_read_b128 = the 8 halfs a lane feeds to one 32x32x16 MFMA.
#include <stdlib.h>

#include <type_traits>

#include "sgam_common.h"

#ifndef SGAM_XGN_MAXC
#define SGAM_XGN_MAXC 1024   // most input channels the fused GroupNorm of the halo kernels takes (scale / shift table in LDS)
#endif


namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct XParams {
    const float *x, *bias, *res;
    const unsigned short *w;   // hi / lo fp16 halves of scale * w in MFMA-fragment order (see frag_index)
    float *out, *ws;
    int B, Hi, Wi, Cin, Ho, Wo, N, KH, KW, stride, pad_t, pad_l, ups;
    int lda, ldb, ldc, ldr, n_valid, bias_per_row;
    int M, ksplit, iters_total, iters_per_split;
    int gx, gy, xcd_swizzle;   // logical grid (M tiles, N tiles); the launch is 1-D, see xcd_block()
    unsigned x_bytes, w_plane_bytes;
    // optional (halo kernels): GroupNorm(32 groups) applied to x while it is staged: per-(image, group) {mean, rstd}
    // [B][32][2] + the affine parameters [Cin]; the per-channel scale / shift are formed in the kernel
    const float *gn_stats, *gn_gamma, *gn_beta;
    int gn_swish;        // ... followed by swish
    // ... or, instead of gn_stats, the producer's partial sums [B][gn_chunks_in][32][2] (fp64 {sum, sumsq} per chunk and group),
    // folded by the consumer itself (GNF kernels: a few chunks, one or two channel slabs per workgroup — no fold launch)
    const double *gn_partial_in;
    double gn_inv_n;     // 1 / (pixels per image x channels per group)
    int gn_chunks_in;
    float gn_eps;
    int red_tc;          // group-major split-K combine: channels per workgroup tile (32, 16 or 8); 0 = row-major combine

    double *gn_partial;  // optional: per-(row-half of the tile, group) {sum, sumsq} of the OUTPUT for the next GroupNorm
    int gn_cpg;          // channels per group of that GroupNorm (N / 32)
    int32_t *range_flag; // optional: set to 1 when an output is not finite (an operand left fp16's range, see sgam_hip.h)
    float inv_w_scale;   // 1 / (a_scale * w_scale), an exact power of two
    float a_scale;       // power of two applied to the A operand before the split (e.g. 1024 for softmax probabilities)
};

// cache policy of the STREAMED operands (activation patches in, residual in, outputs out): 2 = non-temporal.  The weight
// panel of a layer (0.6 - 9 MB) is re-read by every workgroup and should own the L2s; the activations pass through once.
#ifndef SGAM_XNT
#define SGAM_XNT 0
#endif
#ifndef SGAM_XPF_BIG
#define SGAM_XPF_BIG 1
#endif
#ifndef SGAM_XPF_SMALL
#define SGAM_XPF_SMALL 2
#endif
#ifndef SGAM_XSB
#define SGAM_XSB 1
#endif
#ifndef SGAM_XBK_SMALL
#define SGAM_XBK_SMALL 64
#endif
#ifndef SGAM_XWK_SMALL
#define SGAM_XWK_SMALL 2
#endif
// K slab per pipeline step: 32 fp32 elements for the 128-wide tiles; the 64x64 tile (the latency-bound layers) takes
// 64 with EIGHT wavefronts — two groups of four, each doing half of the slab's MFMA k-steps on its own accumulators,
// combined through LDS before the epilogue — so that a barrier round trip buys twice the work and every SIMD has two
// wavefronts to overlap LDS / VALU / MFMA latencies with.  LDS row stride = slab + 8 halfs (80 / 144 bytes): the
// 16-lane groups of ds_read_b128 land on distinct banks.
constexpr int xbk_of(int bm, int bn) { return (bm == 64 && bn == 64) ? SGAM_XBK_SMALL : 32; }
constexpr int wk_of(int bm, int bn) { return (bm == 64 && bn == 64) ? SGAM_XWK_SMALL : 1; }

__device__ __forceinline__ unsigned xsel(bool c, unsigned a, unsigned b) {
    const unsigned m = 0u - (unsigned)c;
    return (a & m) | (b & ~m);
}

// exact two-piece split of four fp32 values -> 4 hi halfs (8 bytes) + 4 lo halfs (8 bytes)
__device__ __forceinline__ void split4(const f32x4 v, u32x2 &hi, u32x2 &lo) {
    unsigned short h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 fh = (_Float16)v[e];
        const _Float16 fl = (_Float16)(v[e] - (float)fh);
        h[e] = __builtin_bit_cast(unsigned short, fh);
        l[e] = __builtin_bit_cast(unsigned short, fl);
    }
    hi[0] = (unsigned)h[0] | ((unsigned)h[1] << 16);
    hi[1] = (unsigned)h[2] | ((unsigned)h[3] << 16);
    lo[0] = (unsigned)l[0] | ((unsigned)l[1] << 16);
    lo[1] = (unsigned)l[2] | ((unsigned)l[3] << 16);
}

__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// PF = register prefetch distance in K slabs: the loads of slab t+PF are in flight while slab t is multiplied.  The
// small tiles run on the latency-bound layers (16^2..64^2 maps, split-K: a few slabs per workgroup, one or two
// workgroups per CU), where a distance of 1 leaves every slab waiting a full memory round trip.
// XCD-aware workgroup -> tile map.  The dispatcher deals consecutive workgroups round-robin to the 8 XCDs (observed:
// workgroup b runs on XCD b % 8), each with a private 4 MB L2, so neighbouring tiles — which share filter-tap halos of
// the input and the weight panel — would each fetch their operands into a different L2.  The launch is 1-D and
// workgroup L takes logical tile L' = (L % 8) * T/8 + L / 8 (bijective form for T % 8 != 0): every XCD owns a
// contiguous run of tiles, M-tile index fastest.  Placement only affects speed, never results.
__device__ __forceinline__ void xcd_block(const XParams &p, int &bx, int &by, int &bz) {
    const unsigned L = blockIdx.x, T = gridDim.x;
    unsigned Lp = L;
    if (p.xcd_swizzle) {
        const unsigned q = T >> 3, r = T & 7u, xcd = L & 7u, idx = L >> 3;
        Lp = xcd * q + (xcd < r ? xcd : r) + idx;
    }
    bx = (int)(Lp % (unsigned)p.gx);
    const unsigned t = Lp / (unsigned)p.gx;
    by = (int)(t % (unsigned)p.gy);
    bz = (int)(t / (unsigned)p.gy);
}

// per-channel {scale, shift} of the fused input GroupNorm for channels c .. c+3 (one group: 32 groups of >= 4 channels):
// scale = rstd * gamma, shift = beta - mean * scale — the same expressions, in the same order, as the stand-alone
// GroupNorm kernels, so fused and two-pass results are bit-identical.  t0 = {s0, h0, s1, h1}, t1 = {s2, h2, s3, h3}.
template <bool GNF = false>
__device__ __forceinline__ void gn_scale_shift(const XParams &p, int b, int c, f32x4 &t0, f32x4 &t1) {
    const int g = c / (p.Cin / 32);
    float mean, rstd;
    if constexpr (GNF) {
        // fold the producer's chunk partials of this group and finish like gn_finalize_stats_kernel.  The caller's staging map
        // gives the eight lanes l, l ^ 8, l ^ 16, l ^ 32 ... of a wavefront the same channel quad, hence the same group: lane
        // sub = l >> 3 fetches chunks sub and sub + 8 (both loads in flight at once — a per-chunk loop would be a chain of
        // dependent L2 round trips: +4 .. 8 us per launch, measured), a three-step xor butterfly adds them in a fixed order
        const int sub = (threadIdx.x & 63) >> 3;
        const double *q = p.gn_partial_in + ((int64_t)b * p.gn_chunks_in * 32 + g) * 2;
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const f64x2 z = {0.0, 0.0};
        const f64x2 a = sub < p.gn_chunks_in ? *reinterpret_cast<const f64x2 *>(q + (int64_t)sub * 64) : z;
        const f64x2 c = sub + 8 < p.gn_chunks_in ? *reinterpret_cast<const f64x2 *>(q + (int64_t)(sub + 8) * 64) : z;
        double s = a[0] + c[0], ss = a[1] + c[1];
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            s += __shfl_xor(s, o, 64);
            ss += __shfl_xor(ss, o, 64);
        }
        const double m = s * p.gn_inv_n;
        double var = ss * p.gn_inv_n - m * m;
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
    } else {
        mean = p.gn_stats[(b * 32 + g) * 2];
        rstd = p.gn_stats[(b * 32 + g) * 2 + 1];
    }
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(p.gn_gamma + c), be = *reinterpret_cast<const f32x4 *>(p.gn_beta + c);
    float sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sc[e] = rstd * ga[e];
        sh[e] = be[e] - mean * sc[e];
    }
    t0 = f32x4{sc[0], sh[0], sc[1], sh[1]};
    t1 = f32x4{sc[2], sh[2], sc[3], sh[3]};
}

#ifndef SGAM_XLB64
#define SGAM_XLB64 2       // workgroups per CU the 64-row halo tile is compiled for.  3 caps it at 168 registers (three wavefronts per SIMD): the
#endif                     //    peeled GroupNorm form then spills 12 bytes and measured 21.2 against 20.3 us in the frame; 2 lets it take 172
#ifndef SGAM_XRWARM
#define SGAM_XRWARM 0      // halo kernels: L2 warm-up of the residual tile from the second-to-last slab (see rwarm).  Measured: no effect —
#endif                     //    340.7 / 340.8 / 340.0 against 341.1 / 340.4 / 340.6 frames/s (the tile is L2 / MALL resident inside the frame): off

#ifndef SGAM_XNBR64
#define SGAM_XNBR64 6      // weight-fragment ring of the 64-row halo tile's GroupNorm launches on grids of <= 2 workgroups per CU: 3 or 6 sets
#endif
#ifndef SGAM_XPEEL
#define SGAM_XPEEL 1       // halo kernels: the last two slabs of a workgroup peeled (no staging of a slab that does not exist)
#endif
// ---- epilogue shared by the tile kernels: each wavefront transposes its (32 TM) x (32 TN) fp32 tile through a private
// LDS region so that a lane ends up with 4 CONSECUTIVE output channels of one pixel: residual comes in and the result
// leaves as 16-byte accesses, 16 lanes covering a 256-byte row segment (the MFMA D layout alone gives 4-byte accesses:
// 64 store + 64 load instructions per 32x32 tile instead of 4 + 4).  The caller guarantees every wavefront is done
// reading the operand LDS.  rowmap(tile row) -> global output pixel index.
// WGM = wavefronts along M (2: the 2 x 2 grid, 1: four wavefronts side by side along N).
template <int BM, int BN, int WGM, class RowMap>
__device__ __forceinline__ void xepilogue(const XParams &p, f32x16 (&acc)[BM / (32 * WGM)][BN / (32 * (4 / WGM))], float *smem_f,
                                          int wave, int lane, int bx, int bz, int n0, RowMap rowmap) {
    constexpr int WGN = 4 / WGM;
    constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
    constexpr int WM = 32 * TM, WN = 32 * TN, LDR = WN + 4;
    const int wm = WGM == 4 ? wave : (WGM == 2 ? wave >> 1 : 0), wn = WGM == 4 ? 0 : (WGM == 2 ? (wave & 1) : wave);
    float *region = smem_f + wave * (WM * LDR);
    const bool to_ws = p.ws != nullptr;
    const int n_lim = to_ws ? p.N : p.n_valid;
    const int ldo = to_ws ? p.N : p.ldc;
    float *obase = to_ws ? p.ws + (int64_t)bz * p.M * p.N : p.out;
    const unsigned o_bytes = (unsigned)(((int64_t)(p.M - 1) * ldo + n_lim) * 4);
    const unsigned r_bytes = (p.res && !to_ws) ? (unsigned)(((int64_t)(p.M - 1) * p.ldr + p.n_valid) * 4) : 0u;
    const unsigned bias_bytes = (p.bias && !to_ws) ? (unsigned)((p.bias_per_row ? p.M : p.N) * 4) : 0u;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)obase, 0, (int)o_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)p.res, 0, (int)r_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)p.bias, 0, (int)bias_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const float inv = to_ws ? 1.0f : p.inv_w_scale;
    const int col_l = lane & 31;
    const int row_h = 4 * (lane >> 5);
    const int wn0 = n0 + wn * (BN / WGN);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = wn0 + j * 32 + col_l;
            const float bias_n = (p.bias_per_row || to_ws) ? 0.f
                                     : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                           rb, (int)xsel(n < n_lim, (unsigned)n * 4u, OOB), 0, 0));
#pragma unroll
            for (int e = 0; e < 16; ++e)
                region[(i * 32 + (e & 3) + 8 * (e >> 2) + row_h) * LDR + j * 32 + col_l] = acc[i][j][e] * inv + bias_n;
        }
    // read back row-major: 16 lanes x float4 cover 64 columns; 4 rows per pass (wave-private region: no barrier)
    constexpr int C4 = WN / 4;          // float4 chunks per row: 8 or 16
    constexpr int RPP = 64 / C4;        // rows per pass: 8 or 4
    const int c4 = lane % C4, rr0 = lane / C4;
    const int n4 = wn0 + c4 * 4;
    const bool n_ok = n4 < n_lim;       // n_valid is a multiple of 4
    float gs = 0.f, gss = 0.f;   // this lane's share of the output statistics (4 channels x WM/RPP pixels)
    bool bad = false;            // a non-finite output: an operand overflowed the fp16 hi half (or fp32 itself overflowed)
#pragma unroll
    for (int pass = 0; pass < WM / RPP; ++pass) {
        const int row = rr0 + pass * RPP;
        const int m = rowmap(wm * (BM / WGM) + row);     // global output pixel of this tile row (>= M: outside)
        const bool ok = n_ok && m < p.M;
        f32x4 v = *reinterpret_cast<const f32x4 *>(region + row * LDR + c4 * 4);
        const f32x4 rv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                       rr, (int)xsel(ok, (unsigned)(m * p.ldr + n4) * 4u, OOB), 0, SGAM_XNT));
        if (p.bias_per_row) {
            const float bm = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                           rb, (int)xsel(ok, (unsigned)m * 4u, OOB), 0, 0));
            v += bm;
        }
        v += rv;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, (int)xsel(ok, (unsigned)(m * ldo + n4) * 4u, OOB), 0, SGAM_XNT);
        if (ok) {
            const float t4 = (v[0] + v[1]) + (v[2] + v[3]);
            gs += t4;
            gss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            bad |= sgam_not_finite(t4);
        }
    }
    if (bad && !to_ws && p.range_flag) atomicOr(p.range_flag, 1);      // rare path; partial sums are checked by the combine
    if (p.gn_partial && !to_ws) {
        // GroupNorm statistics of the tensor just written, for free: a float4 never straddles a group (cpg is a
        // multiple of 4).  Lanes -> LDS, then one lane per group of this wavefront's column range folds, in a fixed
        // order and in fp64, the lanes that hold that group; chunk index = (m-tile, row-half) of the output.
        float *sl = region;                   // reuse the wave-private region: [64 lanes][2]
        sl[lane * 2] = gs;
        sl[lane * 2 + 1] = gss;
        const int c4_per_group = p.gn_cpg / 4;
        const int groups_here = C4 / c4_per_group;       // groups inside this wavefront's WN columns
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
                const int chunk = bx * 2 + wm;          // every output row belongs to exactly one chunk
                const int hw = p.Ho * p.Wo;
                const int b = (bx * BM) / hw;                    // host guarantees a tile never straddles two images
                const int chunks_per_b = ((hw + BM - 1) / BM) * 2;
                const int cb = chunk - b * chunks_per_b;
                double *o = p.gn_partial + (((int64_t)b * chunks_per_b + cb) * groups + g) * 2;
                o[0] = ds;
                o[1] = dss;
                if (WGM == 1) {
                    o[groups * 2] = 0.0;
                    o[groups * 2 + 1] = 0.0;
                }
            }
        }
    }
}

template <int BM, int BN, bool UPS, bool ASCALE>
__global__ __launch_bounds__(256 * wk_of(BM, BN)) void conv_gemm_f32x_kernel(const XParams p) {
    constexpr int XBK = xbk_of(BM, BN), WK = wk_of(BM, BN), NT = 256 * WK;
    constexpr int XLD = XBK + 8;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int PF = (BM * BN >= 128 * 128) ? SGAM_XPF_BIG : SGAM_XPF_SMALL;
    constexpr int AC = XBK / 4;            // float4 columns of an A slab row
    static_assert(NT / AC == 32, "the A staging map assumes 32 rows per pass");
    constexpr int AR = BM / 32;            // float4 rows of A per thread
    constexpr int SL = XBK / 32;           // 32-element K slabs per pipeline step
    constexpr int NPB = (BN / 32) * SL * 256 / NT;   // 16-byte B pieces per thread (see b_piece)
    constexpr int KS = XBK / 16;           // MFMA k-steps per slab
    static_assert(KS % WK == 0, "k-steps split evenly over the wavefront groups");
    constexpr int PLANE_A = BM * XLD, PLANE_B = BN * XLD;
    constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;   // halfs per pipeline stage
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = (tid >> 6) & 3;       // position in the 2x2 wavefront grid of the tile
    const int wk = tid >> 8;               // k-step group (0 when WK == 1)
    const int wm = wave >> 1, wn = wave & 1;
    int bx, by, bz;
    xcd_block(p, bx, by, bz);
    const int m0 = bx * BM;
    const int n0 = by * BN;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
    // B operand: [N][ldb / 32][2][32] halfs — the hi and the lo halves of a 32-element K slab share one 128-byte line
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)p.w_plane_bytes, 0x00020000);

    const int it0 = bz * p.iters_per_split;
#ifndef SGAM_XABLATE
#define SGAM_XABLATE 0
#endif
    const int it1 = SGAM_XABLATE == 1 ? it0 : min(p.iters_total, it0 + p.iters_per_split);

    // A staging: thread -> (float4 column col4 of the 32-wide slab, rows row_in_pass + 32 r)
    const int col4 = tid % AC;
    const int row_in_pass = tid / AC;
    // B staging: the weights are stored in MFMA-fragment order — per (32-row tile, 32-element slab) 256 pieces of 16
    // bytes, piece = ((plane * 2 + k-step) * 2 + k-half) * 32 + row — so thread t takes piece t & 255 of block t >> 8 (+
    // NT / 256 per further piece): every wavefront load is one contiguous kilobyte.
    const int Hl = p.ups ? 2 * p.Hi : p.Hi;
    const int Wl = p.ups ? 2 * p.Wi : p.Wi;

    // per staged row: byte offset of the tap-(0,0) input pixel and a bit mask of the taps that fall inside the image
    // (non-upsampled convs: the per-step address is rowoff + a wave-uniform tap offset, validity is one bit test);
    // the nearest-2x upsampling conv keeps the generic coordinate arithmetic.
    int a_iy0[AR], a_ix0[AR], a_base[AR];
    unsigned a_rowoff[AR], a_tapmask[AR];
#pragma unroll
    for (int r = 0; r < AR; ++r) {
        const int m = m0 + row_in_pass + 32 * r;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int hw = p.Ho * p.Wo;
        const int b = mm / hw;
        const int rem = mm - b * hw;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        a_iy0[r] = ok ? oy * p.stride - p.pad_t : -(1 << 28);
        a_ix0[r] = ox * p.stride - p.pad_l;
        a_base[r] = b * p.Hi * p.Wi;
        a_rowoff[r] = (unsigned)((a_base[r] + a_iy0[r] * p.Wi + a_ix0[r]) * p.lda + col4 * 4) * 4u;   // may wrap: only used when valid
        unsigned mk = 0;
        for (int t = 0; t < p.KH * p.KW; ++t) {
            const int iy = a_iy0[r] + t / p.KW, ix = a_ix0[r] + t % p.KW;
            mk |= (((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) ? 1u : 0u) << t;
        }
        a_tapmask[r] = mk;
    }
    unsigned b_off[NPB];      // byte offset of the piece at slab 0 (or out of range), and where it lands in an LDS stage
    int b_lds[NPB], b_k[NPB];
    const unsigned slabs_per_row = (unsigned)p.ldb / 32u;
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        const int L = tid + NT * j, blk = L >> 8, pi = L & 255;
        const int nt = blk / SL, sl = blk - nt * SL;
        const int plane = pi >> 7, kk = (pi >> 6) & 1, kh = (pi >> 5) & 1, row = pi & 31;
        const int n = n0 + nt * 32 + row;
        b_off[j] = n < p.N ? ((((unsigned)(n >> 5) * slabs_per_row + (unsigned)sl) * 256u + (unsigned)pi) * 16u) : 0xC0000000u;
        b_lds[j] = plane * PLANE_B + (nt * 32 + row) * XLD + sl * 32 + kk * 16 + kh * 8;
        b_k[j] = sl * 32 + kk * 16 + kh * 8;
    }

    const int taps = p.KH * p.KW;
    int ch = it0 / taps;
    int tap = it0 - ch * taps;
    int ky = tap / p.KW;
    int kx = tap - ky * p.KW;

    f32x4 areg[PF][AR];
    u32x4 bp[PF][NPB];
    auto issue_loads = [&](const int st, bool live) {
        if (SGAM_XABLATE == 2) live = false;
        const int coff = ch * XBK + col4 * 4;
        const bool k_ok = live && coff < p.Cin;
        if constexpr (UPS) {
#pragma unroll
            for (int r = 0; r < AR; ++r) {
                const int iy = a_iy0[r] + ky, ix = a_ix0[r] + kx;
                const bool ok = k_ok && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;
                const int py = iy >> 1, px = ix >> 1;
                const unsigned off = (unsigned)((a_base[r] + py * p.Wi + px) * p.lda + coff) * 4u;
                areg[st][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)xsel(ok, off, p.x_bytes), 0, 0));
            }
        } else {
            const unsigned tap_off = (unsigned)((ky * p.Wi + kx) * p.lda + ch * XBK) * 4u;   // wave-uniform
#pragma unroll
            for (int r = 0; r < AR; ++r) {
                const bool ok = k_ok && ((a_tapmask[r] >> tap) & 1u);
                areg[st][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                            rx, (int)xsel(ok, a_rowoff[r] + tap_off, p.x_bytes), 0, 0));
            }
        }
        const unsigned koff = (unsigned)(tap * p.Cin + ch * XBK) * 128u;    // 4096 bytes per (row tile, 32-element slab)
#pragma unroll
        for (int j = 0; j < NPB; ++j) {
            const bool kb_ok = live && (ch * XBK + b_k[j]) < p.Cin;
            bp[st][j] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)xsel(kb_ok, b_off[j] + koff, p.w_plane_bytes), 0, 0);
        }
        ++tap;
        if (++kx == p.KW) {
            kx = 0;
            ++ky;
        }
        if (tap == taps) {
            tap = 0; ky = 0; kx = 0;
            ++ch;
        }
    };
    auto store_lds = [&](const int st, int buf) {
        unsigned short *ah = smem + buf * STAGE;
        unsigned short *al = ah + PLANE_A;
        unsigned short *bhp = al + PLANE_A;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            u32x2 hi, lo;
            if (SGAM_XABLATE == 4) {
                const u32x4 raw = __builtin_bit_cast(u32x4, areg[st][r]);
                hi[0] = raw[0]; hi[1] = raw[1]; lo[0] = raw[2]; lo[1] = raw[3];
            } else
            split4(ASCALE ? areg[st][r] * p.a_scale : areg[st][r], hi, lo);
            const int o = (row_in_pass + 32 * r) * XLD + col4 * 4;
            *reinterpret_cast<u32x2 *>(ah + o) = hi;
            *reinterpret_cast<u32x2 *>(al + o) = lo;
        }
#pragma unroll
        for (int j = 0; j < NPB; ++j) *reinterpret_cast<u32x4 *>(bhp + b_lds[j]) = bp[st][j];
    };

    // one accumulator per output tile; a wavefront that owns a single tile keeps the cross terms (a_lo b_hi + a_hi b_lo)
    // in a second one so that consecutive MFMAs never wait on each other's result
    constexpr bool SEPACC = (TM * TN == 1);
    f32x16 acc[TM][TN], accs[1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) accs[0][e] = 0.f;

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 8;

    // prologue: slabs it0 .. it0+PF-1 in flight, slab it0 into LDS, its register stage re-armed with slab it0+PF
#pragma unroll
    for (int s = 0; s < PF; ++s) issue_loads(s, it0 + s < it1);
    store_lds(0, 0);
    issue_loads(0, it0 + PF < it1);
    __syncthreads();

    int it = it0;
    while (it < it1) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {       // unrolled by PF: the register stage index is a compile-time constant
            if (it < it1) {
                const int nst = (s + 1) % PF;    // stage holding slab it+1
                const int buf = (it - it0) & 1;
                const unsigned short *ah = smem + buf * STAGE + (wm * (BM / 2) + frag_row) * XLD + frag_k;
                const unsigned short *al = ah + PLANE_A;
                const unsigned short *bhp = smem + buf * STAGE + 2 * PLANE_A + (wn * (BN / 2) + frag_row) * XLD + frag_k;
                const unsigned short *blp = bhp + PLANE_B;
#pragma unroll
                for (int kq = 0; kq < KS / WK; ++kq) {
                    const int kk = kq * WK + wk;          // this wavefront group's k-step inside the slab
                    if (kq == (KS / WK) / 2) {
                        store_lds(nst, buf ^ 1);                      // slab it+1 -> the other LDS buffer
                        issue_loads(nst, it + 1 + PF < it1);          // slab it+1+PF -> the freed register stage
#if SGAM_XSB
                        __builtin_amdgcn_sched_barrier(0);            // do not let the scheduler sink the prefetch
#endif
                    }
                    u32x4 fah[TM], fal[TM], fbh[TN], fbl[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        fah[i] = *reinterpret_cast<const u32x4 *>(ah + i * 32 * XLD + kk * 16);
                        fal[i] = *reinterpret_cast<const u32x4 *>(al + i * 32 * XLD + kk * 16);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        fbh[j] = *reinterpret_cast<const u32x4 *>(bhp + j * 32 * XLD + kk * 16);
                        fbl[j] = *reinterpret_cast<const u32x4 *>(blp + j * 32 * XLD + kk * 16);
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            // small terms first so that they are not swamped one by one by the running sum
                            if (SGAM_XABLATE == 3) {
                                acc[i][j][0] += __builtin_bit_cast(float, fal[i][0] ^ fbh[j][0] ^ fah[i][1] ^ fbl[j][1]);
                                continue;
                            }
                            if constexpr (SEPACC) {
                                accs[0] = mfma16(fal[i], fbh[j], accs[0]);
                                acc[i][j] = mfma16(fah[i], fbh[j], acc[i][j]);
                                accs[0] = mfma16(fah[i], fbl[j], accs[0]);
                            } else {
                                acc[i][j] = mfma16(fal[i], fbh[j], acc[i][j]);
                                acc[i][j] = mfma16(fah[i], fbl[j], acc[i][j]);
                                acc[i][j] = mfma16(fah[i], fbh[j], acc[i][j]);
                            }
                        }
                }
                __syncthreads();
                ++it;
            }
        }
    }

    // ---- epilogue (the loop's final barrier guarantees every wavefront is done reading the operand slabs)
    if constexpr (SEPACC) acc[0][0] += accs[0];
    constexpr int WM = 32 * TM, WN = 32 * TN, LDR = WN + 4;
    static_assert(4 * WM * LDR * 4 <= 2 * STAGE * 2, "epilogue staging must fit the operand LDS");
    if constexpr (WK > 1) {
        // the second k-step group hands its accumulators over through LDS (lane-linear, behind the transposition
        // regions) and retires; the first group adds them in a fixed order
        constexpr int XCH = 4 * WM * LDR;      // floats used by the four transposition regions
        static_assert((XCH + 4 * TM * TN * 1024) * 4 <= 2 * STAGE * 2, "k-group exchange must fit the operand LDS");
        float *xch = reinterpret_cast<float *>(smem) + XCH + wave * (TM * TN * 1024);
        if (wk == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) xch[((i * TN + j) * 16 + e) * 64 + lane] = acc[i][j][e];
        }
        __syncthreads();
        // (the second group's wavefronts END here while the first group still executes __syncthreads() in xepilogue: that is
        // defined on this hardware — s_barrier counts the wavefronts of the workgroup that have not terminated — and it is the only
        // place the library relies on it; a port to a part whose barrier counts launched wavefronts must keep them alive instead)
        if (wk == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] += xch[((i * TN + j) * 16 + e) * 64 + lane];
    }
    xepilogue<BM, BN, 2>(p, acc, reinterpret_cast<float *>(smem), wave, lane, bx, bz, n0, [&](int row) { return m0 + row; });
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with a HALO-staged A operand (the bulk of the network's FLOPs).
// The generic kernel above fetches, splits and stages the A tile once per filter tap: nine global->LDS passes over
// (almost) the same pixels for every 32-channel slab.  Here the workgroup owns an 8 x 16 (8 x 8) patch of output pixels;
// per slab it stages the 10 x 18 (10 x 10) halo of that patch ONCE (global loads, the fp32 -> hi/lo split and the LDS
// writes of the A side drop 6.4x) and the nine taps read their MFMA A fragments from it at a wavefront-uniform offset.
// (A first generation of this kernel kept the weight tile in LDS, double-buffered per tap with one barrier per tap; it
// is in the history — this second generation replaced it.)  The B (weight) fragments never touch LDS.  In the [N][K/32][hi 32 | lo 32] layout the
// MFMA B operand of a lane (8 consecutive halfs of one output channel's K slab) is one aligned 16-byte piece of a
// 128-byte line, so every wavefront pulls the fragments of its own 64 output channels for the next tap straight into
// registers (8 x buffer_load_dwordx4, each 128-byte line consumed whole across the four (k-step, plane) pieces; the
// wavefront that shares the channel range and the co-resident workgroup hit the same lines in the vector L1).  With B
// out of LDS the only shared operand is the halo, which is double-buffered: ONE barrier per channel slab (nine taps)
// instead of one per tap, no ds_write of weights, half the ds_reads, and wavefronts drift freely inside a slab so the
// two workgroups of a CU interleave their load and MFMA phases.
// UPS: the conv runs on the nearest-2x upsampled input without materialising it — the staged halo is the SOURCE patch
// ((TH/2 + 2) x (TW/2 + 2) pixels: a third of the pixels), and a lane finds the source pixel of (patch pixel, tap) as
// ((p + k - 1) >> 1) + 1 per axis.
template <int BM, int BN, bool GN, bool UPS = false, bool GNF = false, int NB = 3>
__global__ __launch_bounds__(256, (BM == 64 && !GNF) ? SGAM_XLB64 : 2) void conv3x3_f32x_halo2_kernel(const XParams p) {
    static_assert(!GNF || GN, "GNF = GroupNorm statistics folded from the producer's chunk partials: a GN kernel");
#ifndef SGAM_XWGM
#define SGAM_XWGM 1
#endif
    // wavefront layout: 1 = four wavefronts side by side along N (each owns all BM rows x BN / 4 channels: every A fragment
    // is read from LDS by all four), 2 = a 2 x 2 grid (each owns BM / 2 rows x BN / 2 channels: half the LDS reads of A,
    // twice the weight-fragment loads, which two wavefronts share in the vector L1)
    // BN = 32 (narrow outputs: the decoder's conv_out, 128 -> 4 channels, weights padded to ONE 32-channel tile instead of a
    // 128-channel tile of mostly-zero columns): the four wavefronts stack along M, each 32 rows x 32 channels
    // BN = 64 (whole-K workgroups for the 64 x 64 maps: 64 tiles x N / 64 = 256 workgroups WITHOUT a split-K plan and its combine
    // launch): a 2 x 2 grid, each wavefront 32 rows x 32 channels
    constexpr int WGM_ = BN == 32 ? 4 : (BN == 64 ? 2 : SGAM_XWGM), WGN_ = 4 / WGM_;
    constexpr int TH = 8, TW = BM / 8, TWS = (TW == 16) ? 4 : 3;
    constexpr int HROWS = UPS ? TH / 2 + 2 : TH + 2, HWID = UPS ? TW / 2 + 2 : TW + 2, HR = HROWS * HWID;
    static_assert(!(UPS && GN), "no GroupNorm precedes an upsampling conv");
    static_assert(BM == 128 || BM == 64, "8 x 16 or 8 x 8 output patches");
    constexpr int XBK = 32, XLD = XBK + 8;
    constexpr int TM = BM / (32 * WGM_), TN = BN / (32 * WGN_);
    // halo pixel (hy, hx) sits at hy * LP + hx * XLD halfs; the line pitch LP is padded (720 -> 768, 400 -> 448) so that
    // the 16-lane groups of ds_read_b128, which straddle two or more patch rows, land on 16 distinct bank quads for
    // every tap offset (found by enumeration over the hardware's lane groups)
    constexpr int LP = UPS ? ((TW == 16) ? 408 : 240) : ((TW == 16) ? 768 : 448);
    constexpr int HPL = HROWS * LP;                     // halfs per halo plane
    constexpr int HBUF = 2 * HPL;                       // one halo buffer: hi plane, lo plane
    constexpr int NH = (HR * 8 + 255) / 256;            // float4 halo loads per thread
    constexpr int OP_BYTES = 2 * HBUF * 2;
    constexpr int EPI_BYTES = 4 * (32 * TM) * (32 * TN + 4) * 4;
    // fused GroupNorm (not the folding form): per-channel {scale, shift} pairs of the input, formed ONCE per workgroup behind the
    // operand buffers (the 128 x 128 tile has 12 KB to spare there under its epilogue's transpose region) and read back per
    // slab.  Round 3 formed them per slab from global loads issued right behind the halo loads of slab s + 2: the vector-memory
    // queue returns in order, so the wait in front of `rstd * gamma` drained the halo loads' trip to L2 / HBM as well (the
    // listing showed s_waitcnt vmcnt(1), vmcnt(0) 24 MFMAs after the loads) — every wavefront, every slab.
    constexpr int TAB_BYTES = (GN && !GNF) ? SGAM_XGN_MAXC * 8 : 0;
    constexpr int SM_BYTES = (OP_BYTES + TAB_BYTES) > EPI_BYTES ? (OP_BYTES + TAB_BYTES) : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned short smem[SM_BYTES / 2];
    float *gn_tab = reinterpret_cast<float *>(smem + OP_BYTES / 2);        // [Cin][{scale, shift}]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = WGM_ == 4 ? wave : (WGM_ == 2 ? wave >> 1 : 0), wn = WGM_ == 4 ? 0 : (WGM_ == 2 ? (wave & 1) : wave);
    int bx, by, bz;
    xcd_block(p, bx, by, bz);
    const int n0 = by * BN;
    const int tiles_x = p.Wo / TW, tiles_img = tiles_x * (p.Ho / TH);
    const int b = bx / tiles_img;
    const int t_img = bx - b * tiles_img;
    const int ty0 = (t_img / tiles_x) * TH, tx0 = (t_img % tiles_x) * TW;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)p.w_plane_bytes, 0x00020000);

    const int it0 = bz * p.iters_per_split;
    const int it1 = min(p.iters_total, it0 + p.iters_per_split);

    unsigned h_off[NH];
    int h_lds[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const int idx = tid + 256 * j;
        const int row = idx >> 3, col4 = idx & 7;
        const int hy = row / HWID, hx = row - hy * HWID;
        const int iy = (UPS ? ty0 / 2 : ty0) + hy - 1, ix = (UPS ? tx0 / 2 : tx0) + hx - 1;     // source pixel of this halo cell
        const bool ok = row < HR && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        h_off[j] = ok ? (unsigned)(((b * p.Hi + iy) * p.Wi + ix) * p.lda + col4 * 4) * 4u : 0xFFFFFFFFu;
        h_lds[j] = row < HR ? hy * LP + hx * XLD + col4 * 4 : -1;
    }
    // B fragments: lane -> output channel n (plan guarantees N % BN == 0), k-half (lane >> 5) of every 16-element k-step
    unsigned bf_off[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        // fragment-ordered weights: lane l reads piece (plane, k-step) * 64 + l of its 32-row tile: one contiguous KB
        const int nt = (n0 + wn * (BN / WGN_) + j * 32) >> 5;
        bf_off[j] = ((unsigned)nt * ((unsigned)p.ldb / 32u) * 256u + (unsigned)lane) * 16u;
    }

    f32x4 hreg[NH];
    f32x4 gt0, gt1;
    auto hload_issue = [&](int ch, bool live) {      // !live: out-of-range offsets (zeros come back, no memory traffic)
        // the slab's channel offset is wave-uniform: it rides in the load's scalar offset (no VALU per load); the bounds
        // check only sees the vector offset, so a dead load / a padding pixel (h_off = ~0) stays out of range
        const unsigned coff = (unsigned)ch * (XBK * 4u);
#pragma unroll
        for (int j = 0; j < NH; ++j)
            hreg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(live ? h_off[j] : 0xFFFFFFFFu), (int)coff, SGAM_XNT));
    };
    auto hparams = [&](int ch, bool live) {
        if constexpr (GN && GNF) {
            gn_scale_shift<GNF>(p, b, (live ? ch : 0) * XBK + (tid & 7) * 4, gt0, gt1);
        } else if constexpr (GN) {
            const float *t = gn_tab + 2 * ((live ? ch : 0) * XBK + (tid & 7) * 4);
            gt0 = *reinterpret_cast<const f32x4 *>(t);
            gt1 = *reinterpret_cast<const f32x4 *>(t + 4);
        }
    };
    auto hload = [&](int ch, bool live) {
        hload_issue(ch, live);
        hparams(ch, live);
    };
    auto hprep_piece = [&](const int j) {
        {
            u32x2 hi, lo;
            f32x4 v = hreg[j];
            if constexpr (GN) {
                v[0] = v[0] * gt0[0] + gt0[1];
                v[1] = v[1] * gt0[2] + gt0[3];
                v[2] = v[2] * gt1[0] + gt1[1];
                v[3] = v[3] * gt1[2] + gt1[3];
                if (p.gn_swish) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = sgam_swish(v[e]);
                }
                if (h_off[j] == 0xFFFFFFFFu) v = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            split4(v, hi, lo);
            hreg[j] = __builtin_bit_cast(f32x4, u32x4{hi[0], hi[1], lo[0], lo[1]});
        }
    };
    auto hprep = [&]() {
#pragma unroll
        for (int j = 0; j < NH; ++j) hprep_piece(j);
    };
    auto hstore = [&](int hb) {
        unsigned short *halo = smem + hb * HBUF;
#pragma unroll
        for (int j = 0; j < NH; ++j) {
            const u32x4 q = __builtin_bit_cast(u32x4, hreg[j]);
            if (h_lds[j] >= 0) {
                *reinterpret_cast<u32x2 *>(halo + h_lds[j]) = u32x2{q[0], q[1]};
                *reinterpret_cast<u32x2 *>(halo + HPL + h_lds[j]) = u32x2{q[2], q[3]};
            }
        }
    };

    // B fragments: NB = 3 register sets rotate with the tap (9 taps = 3 full turns, so a slab ends where it began); the
    // fragments are requested TWO taps ahead — one tap of MFMAs (384 cycles on the 64-row tile) does not cover an L2
    // round trip when a SIMD holds a single wavefront.  NB = 6 (round 5, the 64-row tile's launches of <= 2 workgroups per CU):
    // FIVE taps ahead; the ring phase of a slab's tap 0 then alternates 0, 3, 0 ... and the slab loop is spelled for the
    // slab counts those launches have (h16_halo.hip has the same ring and the measurements)
    static_assert(NB == 3 || (NB == 6 && SGAM_XPEEL), "ring of three, or of six with the peeled slab loop");
    u32x4 bq[NB][TN][2][2];                // [(slab phase + tap) % NB][n tile][k-step][hi, lo]
    auto bload = [&](const int set, int tap, int ch, bool live) {
        const unsigned koff = (unsigned)(tap * p.Cin + ch * XBK) * 128u;   // 4096 bytes per (row tile, slab); scalar offset
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const unsigned vo = live ? bf_off[j] : 0xFFFFFFF0u;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    bq[set][j][kk][pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)vo, (int)(koff + (unsigned)((pl * 2 + kk) * 1024)), 0);
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

    // The host aligns split-K ranges to whole slabs for this kernel (it0, it1 multiples of 9).  One slab = ONE basic
    // block: nine taps unrolled, no branches (range ends are handled with out-of-range load offsets), so the scheduler
    // can thread the VALU work on the NEXT slab's halo (GroupNorm, swish, hi/lo split: one float4 piece per tap) and
    // the weight-fragment loads of the next tap through the 216 MFMAs of the running slab.
    const int s0 = it0 / 9, s1 = SGAM_XABLATE == 1 ? it0 / 9 : it1 / 9;
    int hcur = 0;
    hload_issue(s0, s0 < s1);
#pragma unroll
    for (int t = 0; t < NB - 1; ++t) bload(t, t, s0, s0 < s1);
    if constexpr (GN && !GNF) {
        // (behind the first halo and weight loads, so that its own round trip overlaps theirs; same expressions and order as
        // gn_scale_shift / the stand-alone GroupNorm kernels)
        const int cpg = p.Cin / 32;
        constexpr int CPT = SGAM_XGN_MAXC / 256;                   // channels per thread, at most
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
                gn_tab[2 * c] = sc;
                gn_tab[2 * c + 1] = p.gn_beta[c] - mr[k][0] * sc;
            }
        }
        __syncthreads();
    }
    hparams(s0, s0 < s1);
    hprep();
    hstore(0);
    if constexpr (SGAM_XPEEL) {
        if (s0 + 1 < s1) hload(s0 + 1, true);          // (a one-slab workgroup has no second halo / second fold of the chunk statistics)
    } else {
        hload(s0 + 1, s0 + 1 < s1);
    }
    __syncthreads();

    u32x4 fa[2][TM][2];                    // [step parity][m tile][hi, lo]
    if constexpr (SGAM_XABLATE == 27) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[a][i][0] = fa[a][i][1] = u32x4{(unsigned)lane, 0x3c003c00u, (unsigned)tid, 0x38003800u};
    }
    const unsigned short *hb = smem;
    auto afrag = [&](const int set, const int tap, const int kk) {
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned short *ah;
            if constexpr (UPS) ah = hb + (((a_py[i] + ky - 1) >> 1) + 1) * LP + (((a_px[i] + kx - 1) >> 1) + 1) * XLD + frag_k;
            else ah = hb + ky * LP + kx * XLD + a_base[i];
            fa[set][i][0] = *reinterpret_cast<const u32x4 *>(ah + kk * 16);
            fa[set][i][1] = *reinterpret_cast<const u32x4 *>(ah + HPL + kk * 16);
        }
    };
    // The A fragments of the plain (non-upsampling) kernel come out of LDS through EXPLICIT ds_read_b128 statements, one
    // step (12 or 6 MFMAs) ahead of their use, retired by a counted wait: left to the compiler the reads are sunk next to
    // the MFMA that consumes them (one register quad re-used for all of them), and every MFMA then waits out an LDS round
    // trip (rocprof: matrix pipe 47 % busy).  Address = per-row base register + compile-time (tap, k-step, plane) offset.
#define XDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
    unsigned a_lds[TM];                    // LDS byte address of (this lane's row of m tile i, tap (0,0), k-step 0, hi plane)
    auto afrag_asm = [&](const int set, const int tap, const int kk) {
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            XDS_READ(fa[set][i][0], a_lds[i], 2 * (ky * LP + kx * XLD + kk * 16));
            XDS_READ(fa[set][i][1], a_lds[i], 2 * (ky * LP + kx * XLD + kk * 16 + HPL));
        }
    };
    // all reads but the newest 2 * TM (= the set issued for the NEXT step) have landed; the fragment registers are tied to
    // the wait so that no MFMA moves above it
    auto await = [&](const int set, const bool next_in_flight) {
        if constexpr (TM == 4) {
            if (next_in_flight)
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fa[set][0][0]), "+v"(fa[set][0][1]), "+v"(fa[set][1][0]), "+v"(fa[set][1][1]),
                             "+v"(fa[set][2][0]), "+v"(fa[set][2][1]), "+v"(fa[set][3][0]), "+v"(fa[set][3][1]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[set][0][0]), "+v"(fa[set][0][1]), "+v"(fa[set][1][0]), "+v"(fa[set][1][1]),
                             "+v"(fa[set][2][0]), "+v"(fa[set][2][1]), "+v"(fa[set][3][0]), "+v"(fa[set][3][1]));
        } else if constexpr (TM == 2) {
            if (next_in_flight)
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[set][0][0]), "+v"(fa[set][0][1]), "+v"(fa[set][1][0]), "+v"(fa[set][1][1]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[set][0][0]), "+v"(fa[set][0][1]), "+v"(fa[set][1][0]), "+v"(fa[set][1][1]));
        } else {
            static_assert(TM == 1 || TM == 2 || TM == 4, "wait counts are spelled for 1, 2 or 4 row tiles");
            if (next_in_flight) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[set][0][0]), "+v"(fa[set][0][1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[set][0][0]), "+v"(fa[set][0][1]));
        }
    };
    constexpr bool XASM = !UPS && (SGAM_XABLATE != 30);
    // L2 warm-up of the residual tile (SGAM_XRWARM, peeled loops): every 128-byte line of the rows this workgroup will add in its
    // epilogue is touched ONCE where the second-to-last slab's (dead) halo request stood — no younger weight load is held up longer than
    // a halo request would have, a slab and a half of MFMAs cover the trip, and the epilogue's float4 loads (issued by all 512
    // workgroups of a one-wave launch at once: 33.5 MB at B = 1 on the 256^2 layers) find the lines in this XCD's L2.  Without a
    // residual (or with split-K partial tiles, whose combine adds it) the descriptor is empty: nothing is fetched.
    constexpr int NWARM = SGAM_XRWARM ? (BM * (BN / 32) + 255) / 256 : 0;
    unsigned rwarm_v[NWARM > 0 ? NWARM : 1];
    auto rwarm = [&]() {
        if constexpr (NWARM > 0) {
            const bool live = p.res != nullptr && p.ws == nullptr;
            const unsigned r_bytes_ = live ? (unsigned)(((int64_t)(p.M - 1) * p.ldr + p.n_valid) * 4) : 0u;
            const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc((void *)p.res, 0, (int)r_bytes_, 0x00020000);
#pragma unroll
            for (int k = 0; k < NWARM; ++k) {
                const int idx = tid + 256 * k, row = idx / (BN / 32), q = idx - row * (BN / 32);
                const int m = (b * p.Ho + ty0 + (row >> TWS)) * p.Wo + tx0 + (row & (TW - 1));
                const int n = n0 + q * 32;
                rwarm_v[k] = __builtin_amdgcn_raw_buffer_load_b32(
                    rr_, (int)xsel(row < BM && n < p.n_valid, (unsigned)(m * p.ldr + n) * 4u, 0xFFFFFFF0u), 0, 0);
            }
        }
    };
    // one slab.  MODE 0: run-time flags say whether a next slab / the one behind it exist (dead loads go out of range and the
    // staging arithmetic runs on the zeros they return); the peeled forms (SGAM_XPEEL, round 5) know: 1 = two more slabs follow,
    // 2 = one more follows (stage it, request nothing), 3 = the last — nothing to stage: a workgroup of a split-K plan on the
    // 16^2 / 32^2 maps walks one or two slabs, so half or all of its in-loop GroupNorm / swish / split work ran on zeros
    auto slab = [&](const int sl, auto mode_, auto rb_) {
        constexpr int MODE = decltype(mode_)::value, RB = decltype(rb_)::value;      // RB: ring phase of this slab's tap 0
        const bool has_next = MODE == 0 ? sl + 1 < s1 : MODE != 3;
        const bool has_next2 = MODE == 0 ? sl + 2 < s1 : MODE == 1;
        hb = smem + hcur * HBUF;
        if constexpr (XASM) {
            const unsigned hb_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short *)hb;
#pragma unroll
            for (int i = 0; i < TM; ++i) a_lds[i] = hb_lds + 2u * (unsigned)a_base[i];
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int set = (tap + RB) % NB;
            // timing experiments (results are wrong): 21 no weight loads, 25 no halo staging, 24 neither, 27 neither and no A-fragment
            // reads (the bare MFMA stream), 26 everything but the MFMAs
            constexpr bool NOB = SGAM_XABLATE == 21 || SGAM_XABLATE == 24 || SGAM_XABLATE == 27;
            constexpr bool NOH = SGAM_XABLATE == 24 || SGAM_XABLATE == 25 || SGAM_XABLATE == 27 || MODE == 3;
            constexpr bool NOA = SGAM_XABLATE == 27, NOM = SGAM_XABLATE == 26;
            if (!NOB) {
                const int tt = tap + NB - 1;
                if (tt < 9) bload((tt + RB) % NB, tt, sl, true);
                else if constexpr (MODE != 3) bload((tt + RB) % NB, tt - 9, sl + 1, has_next);
            }
            // (28: the staging arithmetic and LDS stores run, on stale registers, without the in-loop halo LOADS; 29: the loads are
            // issued, nothing is done with them)
            if (!NOH && SGAM_XABLATE != 29 && tap >= 1 && tap <= NH) hprep_piece(tap - 1);    // next slab's halo, one piece per tap
            if (!NOH && tap == NH + 1) {
                if (SGAM_XABLATE != 29) hstore(hcur ^ 1);               // idle buffer: nobody reads it during this slab
                if constexpr (MODE != 2) {
                    if (SGAM_XABLATE != 28) hload(sl + 2, has_next2);
                } else {
                    rwarm();
                }
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int q = tap * 2 + kk;                 // step inside the slab: 0 .. 17
                // A fragments are read one step ahead (register double buffer fa[q & 1]); step 0 reads its own
                if constexpr (NOA) {
                } else if constexpr (XASM) {
                    if (q == 0) afrag_asm(0, 0, 0);
                    if (q < 17) afrag_asm((q + 1) & 1, (q + 1) >> 1, (q + 1) & 1);
                    await(q & 1, q < 17);
                } else {
                    if (q == 0) afrag(0, 0, 0);
                    if (q < 17) afrag((q + 1) & 1, (q + 1) >> 1, (q + 1) & 1);
                }
                // term-major order: the three MFMAs of one accumulator are TM * TN instructions apart
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if constexpr (NOM) acc[i][j][term] += __builtin_bit_cast(float, fa[q & 1][i][term == 0 ? 1 : 0][0] ^ bq[set][j][kk][term == 1 ? 1 : 0][0]);
                            else acc[i][j] = mfma16(fa[q & 1][i][term == 0 ? 1 : 0], bq[set][j][kk][term == 1 ? 1 : 0], acc[i][j]);
                        }
            }
        }
        __syncthreads();                                               // next halo visible; old one free for re-use
        hcur ^= 1;
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    if constexpr (!SGAM_XPEEL) {
        for (int sl = s0; sl < s1; ++sl) slab(sl, I0{}, I0{});
    } else if constexpr (NB == 3) {
        if (s0 < s1) {
            int sl = s0;
            for (; sl + 2 < s1; ++sl) slab(sl, I1{}, I0{});
            if (s1 - s0 >= 2) slab(s1 - 2, I2{}, I0{});
            slab(s1 - 1, I3{}, I0{});
        }
    } else if constexpr (GNF) {                       // (host: a folding workgroup walks one or two slabs)
        if (s1 - s0 >= 2) {
            slab(s0, I2{}, I0{});
            slab(s0 + 1, I3{}, I3{});
        } else if (s0 < s1) {
            slab(s0, I3{}, I0{});
        }
    } else if (s0 < s1) {                             // (host: an even number of slabs per workgroup)
        int sl = s0;
        for (; sl + 2 < s1; sl += 2) {
            slab(sl, I1{}, I0{});
            slab(sl + 1, I1{}, I3{});
        }
        slab(s1 - 2, I2{}, I0{});
        slab(s1 - 1, I3{}, I3{});
    }
    __syncthreads();                                  // every wavefront is done with the halo: LDS becomes the epilogue's
    if constexpr (NWARM > 0 && SGAM_XPEEL) {
#pragma unroll
        for (int k = 0; k < NWARM; ++k) asm volatile("" ::"v"(rwarm_v[k]));          // (the touches are loads with a destination: retire them here)
    }

    xepilogue<BM, BN, WGM_>(p, acc, reinterpret_cast<float *>(smem), wave, lane, bx, bz, n0, [&](int row) {
        return (b * p.Ho + ty0 + (row >> TWS)) * p.Wo + tx0 + (row & (TW - 1));
    });
}


// fixed-order split-K reduction (partials are still weight-scaled) + un-scale + bias + residual; optionally the GroupNorm
// statistics of what it writes: a workgroup covers 1024 / N whole output rows (N in {128, 256, 512, 1024}), lanes of one
// (row, group) are neighbours -> shuffle fold, rows -> LDS fold, one {sum, sumsq} pair per (workgroup = chunk, group).
__global__ __launch_bounds__(256) void splitk_reduce_f32x_kernel(const XParams p) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = p.N / 4;
    const bool live = q < (int64_t)p.M * nq;
    const int m = live ? (int)(q / nq) : 0;
    const int n = live ? (int)(q - (int64_t)m * nq) * 4 : 0;
    float gs = 0.f, gss = 0.f;
    if (live) {
        // the partial slabs are summed in the order z = 0, 1, 2, ... with four loads in flight at a time (a plain loop
        // compiles to load -> wait -> add: one L2 round trip per slab); bias / residual are fetched before they are used
        const float *w0 = p.ws + (int64_t)m * p.N + n;
        const int64_t zs = (int64_t)p.M * p.N;
        float bv[4], rv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = n + e < p.n_valid;
            bv[e] = (ok && p.bias) ? (p.bias_per_row ? p.bias[m] : p.bias[n + e]) : 0.f;
            rv[e] = (ok && p.res) ? p.res[(int64_t)m * p.ldr + n + e] : 0.f;
        }
        f32x4 s = *reinterpret_cast<const f32x4 *>(w0);
        int z = 1;
        for (; z + 4 <= p.ksplit; z += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(w0 + z * zs), b = *reinterpret_cast<const f32x4 *>(w0 + (z + 1) * zs);
            const f32x4 c = *reinterpret_cast<const f32x4 *>(w0 + (z + 2) * zs), d = *reinterpret_cast<const f32x4 *>(w0 + (z + 3) * zs);
            s += a;
            s += b;
            s += c;
            s += d;
        }
        for (; z < p.ksplit; ++z) s += *reinterpret_cast<const f32x4 *>(w0 + z * zs);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (n + e >= p.n_valid) continue;
            float v = s[e] * p.inv_w_scale;
            if (p.bias) v += bv[e];
            if (p.res) v += rv[e];
            p.out[(int64_t)m * p.ldc + n + e] = v;
            gs += v;
            gss += v * v;
        }
        if (p.range_flag && sgam_not_finite(gs)) atomicOr(p.range_flag, 1);
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

// GROUP-MAJOR form of the combine for the small maps (16^2, 32^2): a workgroup owns TR = 1024 / TC whole-column runs of TC
// channels (a 128- / 64- / 32-byte piece of TR consecutive output rows) instead of 1024 / N whole rows, so that a GroupNorm group
// is covered by hw / TR <= 16 workgroups instead of hw N / 1024 (128 - 256): few enough chunk partials for the CONSUMING
// convolution to fold them itself (GNF kernels) — the 32-workgroup fold launch between the two disappears.  Same fixed
// summation orders (slabs z = 0, 1, ...; rows top to bottom), so results do not depend on scheduling.
template <int TC>
__global__ __launch_bounds__(256) void splitk_reduce_gm_f32x_kernel(const XParams p) {
    constexpr int TR = 1024 / TC, TPR = TC / 4;
    const int col_tiles = p.N / TC;
    const int rt = blockIdx.x / col_tiles, ct = blockIdx.x - rt * col_tiles;
    const int row = threadIdx.x / TPR, c4 = threadIdx.x - row * TPR;
    const int m = rt * TR + row, n = ct * TC + c4 * 4;            // host: M % TR == 0, N % TC == 0, n_valid == N
    const float *w0 = p.ws + (int64_t)m * p.N + n;
    const int64_t zs = (int64_t)p.M * p.N;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, rv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = p.bias_per_row ? f32x4{p.bias[m], p.bias[m], p.bias[m], p.bias[m]} : *reinterpret_cast<const f32x4 *>(p.bias + n);
    if (p.res) rv = *reinterpret_cast<const f32x4 *>(p.res + (int64_t)m * p.ldr + n);
    f32x4 s = *reinterpret_cast<const f32x4 *>(w0);
    int z = 1;
    for (; z + 4 <= p.ksplit; z += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(w0 + z * zs), b = *reinterpret_cast<const f32x4 *>(w0 + (z + 1) * zs);
        const f32x4 c = *reinterpret_cast<const f32x4 *>(w0 + (z + 2) * zs), d = *reinterpret_cast<const f32x4 *>(w0 + (z + 3) * zs);
        s += a;
        s += b;
        s += c;
        s += d;
    }
    for (; z < p.ksplit; ++z) s += *reinterpret_cast<const f32x4 *>(w0 + z * zs);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = s[e] * p.inv_w_scale;
        if (p.bias) v[e] += bv[e];
        if (p.res) v[e] += rv[e];
    }
    *reinterpret_cast<f32x4 *>(p.out + (int64_t)m * p.ldc + n) = v;
    float gs = (v[0] + v[1]) + (v[2] + v[3]);
    float gss = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    if (p.range_flag && sgam_not_finite(gs)) atomicOr(p.range_flag, 1);
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

// B-operand storage = MFMA-fragment order: [row tile of 32][K slab of 32][plane hi/lo][k-step][k-half][row][8 halfs], i.e.
// per (row tile, slab) 256 pieces of 16 bytes with piece = ((plane * 2 + k-step) * 2 + k-half) * 32 + row — exactly what
// the 64 lanes of a wavefront need for one v_mfma_f32_32x32x16_f16 B operand sit in one contiguous kilobyte.
__device__ __forceinline__ int64_t frag_index(int64_t n, int64_t k, int64_t slabs_per_row) {     // index of the HI half
    const int64_t slab = k >> 5, kin = k & 31;
    const int64_t piece = (((kin >> 4) * 2) + ((kin >> 3) & 1)) * 32 + (n & 31);
    return (((n >> 5) * slabs_per_row + slab) * 256 + piece) * 8 + (kin & 7);
}

// [Cout][Cin][KH][KW] fp32 -> fragment order of scale * w, K = (tap, Cin_pad) with Cin_pad % 32 == 0, Cout_pad % 32 == 0,
// zero padded
__global__ void pack_weight_f32x_kernel(const float *w, unsigned short *o, int Cout, int Cin, int KH, int KW, int Cout_pad,
                                        int Cin_pad, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int taps = KH * KW;
    const int64_t total = (int64_t)Cout_pad * taps * Cin_pad;
    if (i >= total) return;
    const int c = (int)(i % Cin_pad);
    const int t = (int)((i / Cin_pad) % taps);
    const int n = (int)(i / ((int64_t)Cin_pad * taps));
    float v = 0.f;
    if (n < Cout && c < Cin) v = w[((int64_t)n * Cin + c) * taps + t] * scale;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    const int64_t k = (int64_t)t * Cin_pad + c;
    const int64_t q = frag_index(n, k, (int64_t)taps * Cin_pad / 32);
    o[q] = __builtin_bit_cast(unsigned short, h);
    o[q + 2 * 2 * 32 * 8] = __builtin_bit_cast(unsigned short, l);      // the lo plane: 128 pieces further
}

// generic [N][K] fp32 matrix (row stride ld) -> fragment order over [Np][Kp], both rounded up to 32 (zero filled): the B
// operand of activation x activation GEMMs
__global__ void split_rows_f32x_kernel(const float *x, unsigned short *o, int N, int Np, int K, int Kp, int ld, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)Np * Kp;
    if (i >= total) return;
    const int n = (int)(i / Kp), k = (int)(i - (int64_t)n * Kp);
    const float v = (k < K && n < N) ? x[(int64_t)n * ld + k] * scale : 0.f;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    const int64_t q = frag_index(n, k, Kp / 32);
    o[q] = __builtin_bit_cast(unsigned short, h);
    o[q + 2 * 2 * 32 * 8] = __builtin_bit_cast(unsigned short, l);
}

struct XPlan {
    int bm, bn, ksplit, iters_total, iters_per_split;
};

// shapes the halo-staged 3x3 kernels take: 3x3 / stride 1 / pad 1 (plain or nearest-2x upsampled input), 8 x 16 (8 x 8) output
// patches, whole 32-channel slabs
static bool halo_shape(const sgam_conv_desc *d, int bm, int bn) {
    static const int halo_on = [] { const char *e = getenv("SGAM_F32X_HALO"); return (e && e[0] == '0') ? 0 : 1; }();
    static const int h64_on = [] { const char *e = getenv("SGAM_F32X_HALO64"); return (e && e[0] == '0') ? 0 : 1; }();
    const bool tile_ok = (bm == 128 && bn == 128 && d->Wo % 16 == 0) || (bm == 64 && bn == 128 && d->Wo % 8 == 0) ||
                         (bm == 128 && bn == 32 && d->Wo % 16 == 0 && d->N == 32 && !d->upsample2x) ||
                         (bm == 64 && bn == 64 && h64_on && d->Wo % 8 == 0 && d->N % 64 == 0 && !d->upsample2x);
    const int up = d->upsample2x ? 2 : 1;
    return halo_on && tile_ok && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 &&
           d->Ho == up * d->Hi && d->Wo == up * d->Wi && d->Ho % 8 == 0 && d->Cin % 32 == 0;
}

XPlan make_xplan(const sgam_conv_desc *d) {
    const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
    XPlan pl;
    auto blocks = [&](int bm, int bn) { return (int64_t)sgam_cdiv(M, bm) * sgam_cdiv(d->N, bn); };
    // conv_out (128 -> 4, padded to ONE 32-channel tile): the halo kernel at any frame size — the only other plan such a descriptor
    // admits is the generic (64, 64) kernel, which also costs the fused GroupNorm + swish a stand-alone pass; small frames
    // fill the chip through the split-K rule below like every other halo plan
    if (d->N == 32 && d->plan_bm == 0 && halo_shape(d, 128, 32)) { pl.bm = 128; pl.bn = 32; }
    else if (d->N % 128 == 0 && blocks(128, 128) >= 224) { pl.bm = 128; pl.bn = 128; }
    else if (d->N % 128 == 0 && blocks(64, 128) >= 224) { pl.bm = 64; pl.bn = 128; }
    else { pl.bm = 64; pl.bn = 64; }
    if (d->plan_bm > 0 && d->plan_bn > 0) { pl.bm = d->plan_bm; pl.bn = d->plan_bn; }
    const int xbk = halo_shape(d, pl.bm, pl.bn) ? 32 : xbk_of(pl.bm, pl.bn);     // (the halo kernels walk 32-channel slabs)
    pl.iters_total = d->KH * d->KW * ((d->Cin + xbk - 1) / xbk);
    const int64_t nb = blocks(pl.bm, pl.bn);
    int ks = 1;
    if (d->plan_ksplit > 0) {
        ks = d->plan_ksplit;
        if (ks > pl.iters_total) ks = pl.iters_total;
    } else if (nb < 192) {
        ks = (int)((384 + nb - 1) / nb);
        const int max_by_iters = pl.iters_total / 4;
        if (ks > max_by_iters) ks = max_by_iters;
        if (ks > 32) ks = 32;
        if (ks < 1) ks = 1;
    }
    pl.iters_per_split = (pl.iters_total + ks - 1) / ks;
    if (halo_shape(d, pl.bm, pl.bn)) pl.iters_per_split = (pl.iters_per_split + 8) / 9 * 9;   // whole channel slabs (9 taps)
    pl.ksplit = (pl.iters_total + pl.iters_per_split - 1) / pl.iters_per_split;
    return pl;
}

int xvalidate(const sgam_conv_desc *d) {
    if (!d) return SGAM_EINVAL;
    if (d->B <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->N <= 0) return SGAM_EINVAL;
    if (d->Cin <= 0 || d->Cin % 8 != 0 || d->N % 4 != 0) return SGAM_EINVAL;
    if (d->KH <= 0 || d->KW <= 0 || d->stride <= 0) return SGAM_EINVAL;
    if (d->lda < d->Cin || d->lda % 4 != 0) return SGAM_EALIGN;
    if (d->ldb < d->KH * d->KW * d->Cin || d->ldb % 32 != 0) return SGAM_EALIGN;          // whole 32-element K slabs per row
    if (d->KH * d->KW > 1 && d->Cin % 32 != 0) return SGAM_EALIGN;                          // taps start on a slab boundary
    if (d->n_valid <= 0 || d->n_valid > d->N || d->ldc < d->n_valid) return SGAM_EINVAL;
    if (d->n_valid % 4 != 0 || d->ldc % 4 != 0 || d->ldr % 4 != 0) return SGAM_EALIGN;   // 16-byte epilogue accesses
    if (d->plan_bm != 0 || d->plan_bn != 0) {
        const bool ok = (d->plan_bm == 128 && d->plan_bn == 128) || (d->plan_bm == 64 && d->plan_bn == 128) ||
                        (d->plan_bm == 128 && d->plan_bn == 32 && d->N == 32) ||
                        (d->plan_bm == 64 && d->plan_bn == 64);
        if (!ok || (d->plan_bn == 128 && d->N % 128 != 0)) return SGAM_EINVAL;
    }
    if (d->plan_ksplit < 0 || d->plan_ksplit > 64) return SGAM_EINVAL;
    return SGAM_OK;
}

}  // namespace

extern "C" int64_t sgam_conv2d_f32x_workspace_bytes(const sgam_conv_desc *d) {
    if (xvalidate(d) != SGAM_OK) return -1;
    const XPlan pl = make_xplan(d);
    if (pl.ksplit <= 1) return 0;
    return (int64_t)pl.ksplit * d->B * d->Ho * d->Wo * d->N * (int64_t)sizeof(float);
}

extern "C" int sgam_conv2d_f32x_plan(const sgam_conv_desc *d, int32_t *bm, int32_t *bn, int32_t *ksplit) {
    const int rc = xvalidate(d);
    if (rc != SGAM_OK) return rc;
    const XPlan pl = make_xplan(d);
    if (bm) *bm = pl.bm;
    if (bn) *bn = pl.bn;
    if (ksplit) *ksplit = pl.ksplit;
    return SGAM_OK;
}

struct XExtra {            // optional fusions around the product
    const float *gn_stats = nullptr, *gn_gamma = nullptr, *gn_beta = nullptr;   // GroupNorm(+swish) of the input (halo kernels)
    int gn_swish = 0;
    const double *gn_partial_in = nullptr;   // ... with the statistics still as the producer's chunk partials (folded in the kernel)
    int gn_chunks_in = 0;
    float gn_eps = 1e-6f;
    double *gn_partial = nullptr;     // statistics of the output: per-chunk partial sums (epilogue or split-K combine)
};

static int conv_f32x_impl(const sgam_conv_desc *d, const float *x, float a_scale, const void *w_planes, float w_scale,
                          const float *bias, const float *residual, float *out, void *workspace, int64_t workspace_bytes,
                          void *stream, const XExtra &ex);

static bool halo_eligible(const sgam_conv_desc *d, const XPlan &pl, float a_scale) {
    return a_scale == 1.0f && halo_shape(d, pl.bm, pl.bn);
}

// 1 when this descriptor runs on a halo-staged 3x3 kernel (plain or nearest-2x upsampled input), 0: generic kernel
extern "C" int32_t sgam_conv2d_f32x_uses_halo(const sgam_conv_desc *d) {
    if (xvalidate(d) != SGAM_OK) return 0;
    return halo_eligible(d, make_xplan(d), 1.0f) ? 1 : 0;
}

extern "C" int32_t sgam_conv2d_f32x_gn_fusable(const sgam_conv_desc *d) {
    if (xvalidate(d) != SGAM_OK || d->upsample2x || d->Cin > SGAM_XGN_MAXC) return 0;
    return halo_eligible(d, make_xplan(d), 1.0f) ? 1 : 0;
}

// channels per workgroup tile of the group-major combine for this descriptor (32, 16 or 8), 0 = keep the row-major combine:
// a group (N / 32 channels) must fit a tile and an image must fall into at most 16 row tiles
static int red_tc_for(const sgam_conv_desc *d) {
    static const int on = [] { const char *e = getenv("SGAM_GN_FOLD"); return (e && e[0] == '0') ? 0 : 1; }();
    const int hw = d->Ho * d->Wo, cpg = d->N / 32;
    if (!on || d->N % 128 != 0 || d->n_valid != d->N || d->N > 1024) return 0;
    // (extending this to the 64 x 64 maps — 32 chunks of 128 rows x 8 channels, consumers that walk four slabs — was measured:
    // 344 -> 337 frames/s; 32-byte row pieces make the combine slower than the fold launch it saves)
    for (int tc = 32; tc >= 8; tc >>= 1) {
        const int tr = 1024 / tc;
        // the fold inside the combine is an xor butterfly over cpg / 4 lanes of whole groups: cpg must divide the tile and be a power of two
        // (N = 384, 768 ... — a ch_mult with a 3 — keep the row-major combine and the two-pass statistics)
        if (tc >= cpg && tc % cpg == 0 && (cpg & (cpg - 1)) == 0 && hw % tr == 0 && hw / tr <= 16) return tc;
    }
    return 0;
}

extern "C" int32_t sgam_conv2d_f32x_stats_chunks(const sgam_conv_desc *d) {
    if (xvalidate(d) != SGAM_OK) return -1;
    const XPlan pl = make_xplan(d);
    const int hw = d->Ho * d->Wo;
    if (d->N % 128 != 0 || d->n_valid != d->N) return 0;                           // 32 groups of >= 4 channels, complete rows
    if (pl.ksplit == 1) {
        // from the conv epilogue: one chunk per (tile, wavefront row)
        if (d->B > 1 && hw % pl.bm != 0) return 0;                                 // a tile must not straddle two images
        return ((hw + pl.bm - 1) / pl.bm) * 2;
    }
    // from the split-K combine: group-major tiles on the small maps (hw / TR <= 16 chunks, foldable by the consumer), else
    // one chunk per workgroup of 1024 outputs = 1024 / N whole rows
    if (const int tc = red_tc_for(d)) return hw / (1024 / tc);
    if (d->N > 1024 || 1024 % d->N != 0 || ((int64_t)hw * d->N) % 1024 != 0) return 0;
    return (int32_t)((int64_t)hw * d->N / 1024);
}

// 1 when a launch of this descriptor can deliver the GroupNorm statistics of its output (per-chunk partial sums,
// sgam_conv2d_f32x_stats_chunks of them per image: from the epilogue without split-K, from the combine with it), else 0
extern "C" int32_t sgam_conv2d_f32x_stats_mode(const sgam_conv_desc *d) {
    return sgam_conv2d_f32x_stats_chunks(d) > 0 ? 1 : 0;
}

static int check_stats_out(const sgam_conv_desc *d, const XExtra &ex) {
    return (ex.gn_partial && sgam_conv2d_f32x_stats_mode(d) != 1) ? SGAM_EINVAL : SGAM_OK;
}

extern "C" int sgam_conv2d_gn_nhwc_f32x(const sgam_conv_desc *d, const float *x, const float *gn_mean_rstd, const float *gn_gamma,
                                        const float *gn_beta, int32_t gn_swish, const void *w_planes, float w_scale,
                                        const float *bias, const float *residual, float *out, double *gn_partial,
                                        void *workspace, int64_t workspace_bytes, void *stream) {
    if (!gn_mean_rstd || !gn_gamma || !gn_beta || !sgam_aligned16(gn_gamma) || !sgam_aligned16(gn_beta) ||
        sgam_conv2d_f32x_gn_fusable(d) != 1 || d->Cin % 128 != 0)
        return SGAM_EINVAL;
    XExtra ex;
    ex.gn_stats = gn_mean_rstd; ex.gn_gamma = gn_gamma; ex.gn_beta = gn_beta; ex.gn_swish = gn_swish ? 1 : 0;
    ex.gn_partial = gn_partial;
    if (check_stats_out(d, ex) != SGAM_OK) return SGAM_EINVAL;
    return conv_f32x_impl(d, x, 1.0f, w_planes, w_scale, bias, residual, out, workspace, workspace_bytes, stream, ex);
}

// 1 when a convolution of this descriptor can normalise its input from `chunks_in` chunk partials per image by itself (the
// folding form of the 64-row halo kernel: at most 16 chunks, at most two channel slabs per workgroup, so that the fold — which is
// repeated per slab — stays a few hundred cycles in the prologue), 0: fold them first (sgam_groupnorm_stats_from_partials_f32)
extern "C" int32_t sgam_conv2d_f32x_gn_foldable(const sgam_conv_desc *d, int32_t chunks_in) {
    static const int on = [] { const char *e = getenv("SGAM_GN_FOLD"); return (e && e[0] == '0') ? 0 : 1; }();
    if (!on || chunks_in < 1 || chunks_in > 16 || sgam_conv2d_f32x_gn_fusable(d) != 1 || d->Cin % 128 != 0) return 0;
    const XPlan pl = make_xplan(d);
    return (pl.bm == 64 && pl.bn == 128 && pl.iters_per_split <= 18) ? 1 : 0;
}

// sgam_conv2d_gn_nhwc_f32x with the statistics of x still as its producer's chunk partials [B][chunks_in][32][2] (fp64 {sum,
// sumsq}): the kernel folds them (needs sgam_conv2d_f32x_gn_foldable(d, chunks_in) == 1)
extern "C" int sgam_conv2d_gnp_nhwc_f32x(const sgam_conv_desc *d, const float *x, const double *gn_partial_in, int32_t chunks_in,
                                         float eps, const float *gn_gamma, const float *gn_beta, int32_t gn_swish, const void *w_planes,
                                         float w_scale, const float *bias, const float *residual, float *out, double *gn_partial,
                                         void *workspace, int64_t workspace_bytes, void *stream) {
    if (!gn_partial_in || !gn_gamma || !gn_beta || !sgam_aligned16(gn_partial_in) || !sgam_aligned16(gn_gamma) ||
        !sgam_aligned16(gn_beta) || !(eps > 0.f) || sgam_conv2d_f32x_gn_foldable(d, chunks_in) != 1)
        return SGAM_EINVAL;
    XExtra ex;
    ex.gn_partial_in = gn_partial_in;
    ex.gn_chunks_in = chunks_in; ex.gn_eps = eps;
    ex.gn_gamma = gn_gamma; ex.gn_beta = gn_beta; ex.gn_swish = gn_swish ? 1 : 0;
    ex.gn_partial = gn_partial;
    if (check_stats_out(d, ex) != SGAM_OK) return SGAM_EINVAL;
    return conv_f32x_impl(d, x, 1.0f, w_planes, w_scale, bias, residual, out, workspace, workspace_bytes, stream, ex);
}

extern "C" int sgam_conv2d_nhwc_f32x(const sgam_conv_desc *d, const float *x, float a_scale, const void *w_planes,
                                     float w_scale, const float *bias, const float *residual, float *out,
                                     void *workspace, int64_t workspace_bytes, void *stream) {
    return conv_f32x_impl(d, x, a_scale, w_planes, w_scale, bias, residual, out, workspace, workspace_bytes, stream, XExtra());
}

extern "C" int sgam_conv2d_stats_nhwc_f32x(const sgam_conv_desc *d, const float *x, float a_scale, const void *w_planes,
                                           float w_scale, const float *bias, const float *residual, float *out,
                                           double *gn_partial, void *workspace, int64_t workspace_bytes, void *stream) {
    XExtra ex;
    ex.gn_partial = gn_partial;
    if (!gn_partial || check_stats_out(d, ex) != SGAM_OK) return SGAM_EINVAL;
    return conv_f32x_impl(d, x, a_scale, w_planes, w_scale, bias, residual, out, workspace, workspace_bytes, stream, ex);
}

static int conv_f32x_impl(const sgam_conv_desc *d, const float *x, float a_scale, const void *w_planes, float w_scale,
                          const float *bias, const float *residual, float *out, void *workspace, int64_t workspace_bytes,
                          void *stream, const XExtra &ex) {
    const int rc = xvalidate(d);
    if (rc != SGAM_OK) return rc;
    if (!x || !w_planes || !out || !(w_scale > 0.f) || !(a_scale > 0.f)) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(w_planes)) return SGAM_EALIGN;
    const XPlan pl = make_xplan(d);
    XParams p;
    p.x = x; p.w = (const unsigned short *)w_planes; p.bias = bias; p.res = residual; p.out = out; p.ws = nullptr;
    p.B = d->B; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.N = d->N;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.ups = d->upsample2x ? 1 : 0;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr; p.n_valid = d->n_valid; p.bias_per_row = d->bias_per_row;
    p.M = d->B * d->Ho * d->Wo;
    p.ksplit = pl.ksplit; p.iters_total = pl.iters_total; p.iters_per_split = pl.iters_per_split;
    p.inv_w_scale = 1.0f / (w_scale * a_scale);
    p.range_flag = sgam_i_range_flag;
    p.a_scale = a_scale;
    p.gn_partial = ex.gn_partial;
    p.gn_cpg = d->N / 32;
    p.gn_stats = ex.gn_stats; p.gn_gamma = ex.gn_gamma; p.gn_beta = ex.gn_beta;
    p.gn_swish = ex.gn_swish;
    p.gn_partial_in = ex.gn_partial_in; p.gn_chunks_in = ex.gn_chunks_in; p.gn_eps = ex.gn_eps;
    p.gn_inv_n = 1.0 / ((double)d->Hi * d->Wi * (d->Cin / 32));
    p.red_tc = 0;

    const int64_t xb = (((int64_t)d->B * d->Hi * d->Wi - 1) * d->lda + d->Cin) * 4;
    const int64_t wb = (int64_t)((d->N + 31) / 32 * 32) * d->ldb * 4;   // fragment order over [N rounded up to 32][ldb]
    if (xb >= (1ll << 32) - 64 || wb >= (1ll << 32) - 256) return SGAM_EINVAL;
    p.x_bytes = (unsigned)xb; p.w_plane_bytes = (unsigned)wb;
    if (pl.ksplit > 1) {
        const int64_t need = (int64_t)pl.ksplit * p.M * p.N * (int64_t)sizeof(float);
        if (!workspace || workspace_bytes < need || !sgam_aligned16(workspace)) return SGAM_EWORKSPACE;
        p.ws = (float *)workspace;
    }
    p.gx = sgam_cdiv(p.M, pl.bm);
    p.gy = sgam_cdiv(p.N, pl.bn);
    static const int swz = [] { const char *e = getenv("SGAM_XCD_SWIZZLE"); return (e && e[0] == '0') ? 0 : 1; }();
    p.xcd_swizzle = swz;
    const dim3 grid((unsigned)((int64_t)p.gx * p.gy * pl.ksplit));   // 1-D: the kernel maps it XCD-aware (xcd_block)
    hipStream_t s = sgam_stream(stream);
#define XLAUNCH(BM_, BN_)                                                                                                   \
    do {                                                                                                                    \
        const dim3 blk(256 * wk_of(BM_, BN_));                                                                              \
        if (p.ups) SGAM_KLAUNCH((conv_gemm_f32x_kernel<BM_, BN_, true, false>), grid, blk, 0, s, p);                  \
        else if (a_scale != 1.0f) SGAM_KLAUNCH((conv_gemm_f32x_kernel<BM_, BN_, false, true>), grid, blk, 0, s, p);   \
        else SGAM_KLAUNCH((conv_gemm_f32x_kernel<BM_, BN_, false, false>), grid, blk, 0, s, p);                       \
    } while (0)
    if (p.ups && a_scale != 1.0f) return SGAM_EINVAL;
    if (d->KH * d->KW > 32) return SGAM_EINVAL;   // tap validity mask is 32 bits
    const bool halo = halo_eligible(d, pl, a_scale);
    const bool gn_tab_on = ex.gn_stats != nullptr;                           // statistics per (image, group): table-filling GN kernels
    if ((gn_tab_on || ex.gn_partial_in) && !halo) return SGAM_EINVAL;
    if (gn_tab_on && d->Cin > SGAM_XGN_MAXC) return SGAM_EINVAL;             // the scale / shift table of the fused GroupNorm (LDS)
    if (ex.gn_partial_in && (pl.bm != 64 || p.ups)) return SGAM_EINVAL;      // folding consumers: the 64-row halo kernel
    // algorithmic work of this launch: 2 M N K fp32 FLOP; bytes = input + weights + output once
    if (sgam_i_prof_on) sgam_i_prof_shape(p.M, d->n_valid, d->KH * d->KW * d->Cin, pl.ksplit);
    if (sgam_i_prof_on)
        sgam_i_prof_work(2.0 * p.M * d->n_valid * (double)(d->KH * d->KW * d->Cin),
                         4.0 * ((double)d->B * d->Hi * d->Wi * d->Cin + (double)d->n_valid * d->KH * d->KW * d->Cin +
                                (double)p.M * d->n_valid * (residual ? 2 : 1)));
    if (halo) {
        if (p.ups) {
            if (pl.bm == 128) SGAM_KLAUNCH((conv3x3_f32x_halo2_kernel<128, 128, false, true>), grid, dim3(256), 0, s, p);
            else SGAM_KLAUNCH((conv3x3_f32x_halo2_kernel<64, 128, false, true>), grid, dim3(256), 0, s, p);
        } else if (pl.bm == 128 && pl.bn == 32) {
            if (gn_tab_on) SGAM_KLAUNCH((conv3x3_f32x_halo2_kernel<128, 32, true>), grid, dim3(256), 0, s, p);
            else SGAM_KLAUNCH((conv3x3_f32x_halo2_kernel<128, 32, false>), grid, dim3(256), 0, s, p);
        } else if (pl.bm == 128) {
            if (gn_tab_on) SGAM_KLAUNCH((conv3x3_f32x_halo2_kernel<128, 128, true>), grid, dim3(256), 0, s, p);
            else SGAM_KLAUNCH((conv3x3_f32x_halo2_kernel<128, 128, false>), grid, dim3(256), 0, s, p);
        } else if (pl.bn == 64) {
            if (p.gn_partial_in) return SGAM_EINVAL;
            if (gn_tab_on) SGAM_KLAUNCH((conv3x3_f32x_halo2_kernel<64, 64, true>), grid, dim3(256), 0, s, p);
            else SGAM_KLAUNCH((conv3x3_f32x_halo2_kernel<64, 64, false>), grid, dim3(256), 0, s, p);
        } else {
            // ring of six for the launches that leave a SIMD one or two wavefronts (grid <= 2 workgroups per CU) and whose slab count the
            // kernel spells: even, or the one / two slabs of the folding form
            const int slabs_wg = p.iters_per_split / 9;
            const bool deep6 = SGAM_XNBR64 == 6 && SGAM_XPEEL && (int64_t)grid.x * grid.y * grid.z <= 2 * 256 && p.iters_per_split % 9 == 0 &&
                               (p.gn_partial_in ? slabs_wg <= 2 : (slabs_wg % 2 == 0 && (p.iters_total / 9) % slabs_wg == 0));
#if SGAM_XNBR64 == 6 && SGAM_XPEEL
            if (deep6 && p.gn_partial_in) SGAM_KLAUNCH((conv3x3_f32x_halo2_kernel<64, 128, true, false, true, 6>), grid, dim3(256), 0, s, p);
            else if (deep6 && gn_tab_on) SGAM_KLAUNCH((conv3x3_f32x_halo2_kernel<64, 128, true, false, false, 6>), grid, dim3(256), 0, s, p);
            else
#endif
            if (p.gn_partial_in) SGAM_KLAUNCH((conv3x3_f32x_halo2_kernel<64, 128, true, false, true>), grid, dim3(256), 0, s, p);
            else if (gn_tab_on) SGAM_KLAUNCH((conv3x3_f32x_halo2_kernel<64, 128, true>), grid, dim3(256), 0, s, p);
            else SGAM_KLAUNCH((conv3x3_f32x_halo2_kernel<64, 128, false>), grid, dim3(256), 0, s, p);
            (void)deep6;
        }
    } else if (pl.bm == 128 && pl.bn == 128) XLAUNCH(128, 128);
    else if (pl.bm == 64 && pl.bn == 128) XLAUNCH(64, 128);
    else XLAUNCH(64, 64);
#undef XLAUNCH
    SGAM_LAUNCH_CHECK();
    if (pl.ksplit > 1) {
        const int64_t q = (int64_t)p.M * (p.N / 4);
        const int tc = p.gn_partial ? red_tc_for(d) : 0;
        if (tc == 32) SGAM_KLAUNCH(splitk_reduce_gm_f32x_kernel<32>, dim3((unsigned)(q / 256)), dim3(256), 0, s, p);
        else if (tc == 16) SGAM_KLAUNCH(splitk_reduce_gm_f32x_kernel<16>, dim3((unsigned)(q / 256)), dim3(256), 0, s, p);
        else if (tc == 8) SGAM_KLAUNCH(splitk_reduce_gm_f32x_kernel<8>, dim3((unsigned)(q / 256)), dim3(256), 0, s, p);
        else SGAM_KLAUNCH(splitk_reduce_f32x_kernel, dim3(sgam_cdiv(q, 256)), dim3(256), 0, s, p);
        SGAM_LAUNCH_CHECK();
    }
    return SGAM_OK;
}

extern "C" int sgam_pack_conv_weight_f32x(const float *w_oihw, void *w_planes, float w_scale, int32_t Cout, int32_t Cin,
                                          int32_t KH, int32_t KW, int32_t Cout_pad, int32_t Cin_pad, void *stream) {
    if (!w_oihw || !w_planes || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || Cout_pad < Cout || Cin_pad < Cin || Cin_pad % 32 || Cout_pad % 32) return SGAM_EINVAL;
    const int64_t total = (int64_t)Cout_pad * KH * KW * Cin_pad;
    SGAM_KLAUNCH(pack_weight_f32x_kernel, dim3(sgam_cdiv(total, 256)), dim3(256), 0, sgam_stream(stream), w_oihw,
                       (unsigned short *)w_planes, Cout, Cin, KH, KW, Cout_pad, Cin_pad, w_scale);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_split_rows_f32x(const float *x, void *planes, float scale, int32_t N, int32_t K, int32_t ld, void *stream) {
    if (!x || !planes || N <= 0 || K <= 0 || ld < K) return SGAM_EINVAL;
    const int Kp = (K + 31) / 32 * 32, Np = (N + 31) / 32 * 32;
    const int64_t total = (int64_t)Np * Kp;
    SGAM_KLAUNCH(split_rows_f32x_kernel, dim3(sgam_cdiv(total, 256)), dim3(256), 0, sgam_stream(stream), x,
                       (unsigned short *)planes, N, Np, K, Kp, ld, scale);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// diagnostic (not part of the ABI header): resident workgroups per CU of the three tile variants
extern "C" int sgam_debug_f32x_occupancy(int which) {
    int n = -1;
    hipError_t e;
    if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_gemm_f32x_kernel<128, 128, false, false>, 256, 0);
    else if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_gemm_f32x_kernel<64, 128, false, false>, 256, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_gemm_f32x_kernel<64, 64, false, false>, 256 * wk_of(64, 64), 0);
    return e == hipSuccess ? n : -(int)e;
}
