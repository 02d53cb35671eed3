// gemm_gn_f32x.hip — out[M][N] = GroupNorm(x)[M][K] . W[N][K]^T + bias on the split-fp32 path (gfx950 only): the fused q | k | v
// projection of the VQGAN AttnBlock (reference modules/diffusionmodules/model.py:168-175: h_ = self.norm(x); q = self.q(h_);
// k = self.k(h_); v = self.v(h_) — three 1x1 convolutions of the normalised tensor, here one GEMM against the stacked weights).
//
// Why its own kernel.  The generic implicit-GEMM kernel (conv_f32x.hip) walks K in 32-wide slabs with a barrier each: for a
// 1x1 convolution a slab is 12 MFMAs per wavefront, i.e. the loop is one barrier + one staging round trip per 384 matrix
// cycles, and the normalisation needs a pass of its own in front (gn_apply: a read + a write of the activation + a launch).
// Here a workgroup stages its whole 64 x K_chunk panel of x ONCE (K_chunk = 256: the C = 256 blocks need a single barrier,
// the C = 512 ones two), normalising (x - mean_g) rstd_g gamma_c + beta_c and splitting into fp16 hi / lo halves on the way
// into LDS; the weights come pre-split in MFMA-fragment order (sgam_split_rows_f32x / SplitWeight) straight from L2 into
// registers, one k-step ahead.  Arithmetic as everywhere on this path: a product = hi.hi + hi.lo + lo.hi on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation.
//
// Tile: 64 rows x 128 columns, four wavefronts side by side (each 64 x 32: two row tiles share every weight fragment).  In
// the 32 x 32 accumulator layout a lane holds one column and 16 rows, so the 32 lanes of a half-wave write 128 contiguous
// bytes of an output row: stores go out directly (no LDS transpose).
//
// The same kernel without the normalisation (GN = false) serves the other 1x1 convolutions and plain GEMMs of the path whose
// shapes fit — AttnBlock.proj_out (+ residual), ResnetBlock.nin_shortcut, quant_conv / post_quant_conv — with the bias /
// residual epilogue and the per-chunk statistics of the output that the next GroupNorm consumes (one chunk per 64-row tile:
// a lane sums its column over the 32 rows it holds, the cpg lanes of a group and the two lane halves are joined by xor
// shuffles; same [B][chunk][32][2] fp64 layout as the conv kernels').  K_chunk = 128 for K = 128.
#include "sgam_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int GBM = 64, GBN = 128;

struct GemmGnParams {
    const float *x;            // [M][lda]
    const unsigned short *w;   // fragment-ordered hi / lo planes: [N / 32][K / 32][256 pieces][8 halfs]
    const float *bias;         // [N] or NULL
    const float *res;          // [M][ldr] or NULL
    float *out;                // [M][ldc]
    double *gn_partial;        // [B][M / HW * HW / 64][32][2] or NULL: statistics of the output per 64-row chunk
    int ldr;
    const float *mean_rstd;    // [B][32][2]
    const float *gamma, *beta; // [K]
    int M, N, K, lda, ldc, HW; // HW rows per image (GroupNorm statistics are per image)
    float inv_w_scale;
    int32_t *range_flag;
};

__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <bool GN, int KC>
__global__ __launch_bounds__(256, 2) void gemm_gn_f32x_kernel(const GemmGnParams p) {
    constexpr int LDK = KC + 8;                                     // LDS row pitch in halfs: rows 4 banks apart
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][GBM][LDK];         // hi plane, lo plane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * GBM, n0 = blockIdx.y * GBN + wave * 32;
    const int b = m0 / p.HW;                                       // host guarantees HW % 64 == 0: a tile lies in one image
    const int cpg = p.K / 32;                                      // channels per group of the INPUT's GroupNorm (32 groups)
    const int slabs = p.K / 32;
    const int lr = lane & 31, lh = lane >> 5;

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // weight fragments of (row tile n0 / 32, slab s, k-step t): pieces ((plane * 2 + t) * 2 + lh) * 32 + lr
    const unsigned short *wt = p.w + (int64_t)(n0 >> 5) * slabs * 2048;
    auto wfrag = [&](int kstep, u32x4 &hi, u32x4 &lo) {            // kstep = global 16-wide k-step
        const unsigned short *q = wt + (int64_t)(kstep >> 1) * 2048 + (((kstep & 1) * 2 + lh) * 32 + lr) * 8;
        hi = *reinterpret_cast<const u32x4 *>(q);
        lo = *reinterpret_cast<const u32x4 *>(q + 1024);           // lo plane: + 128 pieces
    };

    // Round 5: the panel is LATENCY-bound, not bandwidth- or matrix-bound (1.6 GFLOP in 13 us at n = 4096): the staging loop used to
    // keep four 16-byte loads in flight per thread and round (four to sixteen dependent trips to L2 / HBM per panel) and every
    // weight fragment was requested ONE k-step (192 matrix cycles) ahead of its use — an L2 round trip is ~700.  Now all of a
    // thread's panel loads of a chunk are in flight at once (16 x 16 bytes; the next chunk's are requested before this chunk's
    // MFMAs), and the weight fragments run WD = 4 k-steps ahead in a register ring whose first sets are requested before the panel.
    constexpr int NIT = GBM * (KC / 4) / 256;                       // 16-byte pieces of the panel per thread: 16 (KC = 256), 8 (128)
    constexpr int WD = 4;                                           // weight read-ahead in k-steps
    constexpr int KS = KC / 16;                                     // k-steps per chunk
    static_assert(KS % WD == 0, "the ring index of a k-step must not depend on the chunk");
    const int prow = tid / (KC / 4), pc4 = (tid % (KC / 4)) * 4;    // this thread's (row, 4 channels) in round 0; round `it` adds 256 / (KC / 4) rows
    f32x4 xr[NIT];
    auto xload = [&](int k0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            xr[it] = *reinterpret_cast<const f32x4 *>(p.x + (int64_t)(m0 + prow + it * (256 / (KC / 4))) * p.lda + k0 + pc4);
    };
    u32x4 wh[WD], wl[WD];
    const int nks = p.K / 16;
#pragma unroll
    for (int t = 0; t < WD; ++t) wfrag(t < nks ? t : nks - 1, wh[t], wl[t]);
    xload(0);
    for (int k0 = 0; k0 < p.K; k0 += KC) {
        if (k0) __syncthreads();                                    // the previous panel has been consumed
        // ---- stage the 64 x KC panel: normalise, split, store.  Thread -> (row, 4 consecutive channels): a half-wave of 64
        // threads covers one row (256 channels), so global reads are whole rows and LDS writes are conflict free
        {
            const int c = k0 + pc4;
            float mean = 0.f, rstd = 1.f;
            f32x4 ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
            if constexpr (GN) {
                const int g = c / cpg;
                mean = p.mean_rstd[(b * 32 + g) * 2], rstd = p.mean_rstd[(b * 32 + g) * 2 + 1];
                ga = *reinterpret_cast<const f32x4 *>(p.gamma + c);
                be = *reinterpret_cast<const f32x4 *>(p.beta + c);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = prow + it * (256 / (KC / 4));
                f32x4 v = xr[it];
                if constexpr (GN) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean) * rstd * ga[e] + be[e];
                }
                unsigned hi[2], lo[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float a0 = v[2 * e], a1 = v[2 * e + 1];
                    const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
                    const _Float16 l0 = (_Float16)(a0 - (float)h0), l1 = (_Float16)(a1 - (float)h1);
                    hi[e] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                    lo[e] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                }
                *reinterpret_cast<unsigned long long *>(&sA[0][row][pc4]) = (unsigned long long)hi[0] | ((unsigned long long)hi[1] << 32);
                *reinterpret_cast<unsigned long long *>(&sA[1][row][pc4]) = (unsigned long long)lo[0] | ((unsigned long long)lo[1] << 32);
            }
        }
        if (k0 + KC < p.K) xload(k0 + KC);                          // the next chunk's panel: its trip overlaps this chunk's MFMAs
        __syncthreads();
        // ---- KS k-steps of 16: A fragments from LDS (row = 32 i + lr, k = 16 t + 8 lh + 0..7), weights WD steps ahead
        const int ks0 = k0 / 16;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const u32x4 bh = wh[t % WD], bl = wl[t % WD];
            {
                const int nx = ks0 + t + WD;                        // refill this slot for k-step t + WD (clamped: a dead load of the last)
                wfrag(nx < nks ? nx : nks - 1, wh[t % WD], wl[t % WD]);
            }
            // (pins the request in front of this step's MFMAs: left free, the scheduler sinks every weight load next to its use to
            // save registers, and the ring collapses to the one-step read-ahead it replaced)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4 ah = *reinterpret_cast<const u32x4 *>(&sA[0][i * 32 + lr][t * 16 + lh * 8]);
                const u32x4 al = *reinterpret_cast<const u32x4 *>(&sA[1][i * 32 + lr][t * 16 + lh * 8]);
                acc[i] = mfma16(ah, bh, acc[i]);
                acc[i] = mfma16(ah, bl, acc[i]);
                acc[i] = mfma16(al, bh, acc[i]);
            }
        }
    }
    // ---- epilogue: lane = column n0 + lr, rows 8 (e / 4) + 4 lh + e % 4 of each 32-row tile; a half-wave writes 128 B of a row
    const int n = n0 + lr;
    const float bias = p.bias ? p.bias[n] : 0.f;
    float gs = 0.f, gss = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = m0 + i * 32 + 8 * (e >> 2) + 4 * lh + (e & 3);
            float v = acc[i][e] * p.inv_w_scale + bias;
            if (p.res) v += p.res[(int64_t)row * p.ldr + n];
            p.out[(int64_t)row * p.ldc + n] = v;
            gs += v;
            gss += v * v;
        }
    if (p.range_flag && sgam_not_finite(gs)) atomicOr(p.range_flag, 1);           // an operand left fp16's range
    if (p.gn_partial) {
        // statistics of the output for the next GroupNorm: groups of cpo = N / 32 adjacent columns (4, 8 or 16 lanes) x the
        // two lane halves; one chunk per 64-row tile
        const int cpo = p.N / 32;
        double ds = (double)gs, dss = (double)gss;
        for (int o = 1; o < cpo; o <<= 1) {
            ds += __shfl_xor(ds, o, 64);
            dss += __shfl_xor(dss, o, 64);
        }
        ds += __shfl_xor(ds, 32, 64);
        dss += __shfl_xor(dss, 32, 64);
        if (lh == 0 && (lr % cpo) == 0) {
            const int chunks_per_b = p.HW / GBM, chunk = (m0 - b * p.HW) / GBM;
            double *o = p.gn_partial + (((int64_t)b * chunks_per_b + chunk) * 32 + n / cpo) * 2;
            o[0] = ds;
            o[1] = dss;
        }
    }
}

}  // namespace

// 1 when (M, N, K, HW) fit the kernel: whole 64-row tiles inside one image, whole 128-column tiles, K a multiple of 256 (or 128)
extern "C" int32_t sgam_gemm_gn_f32x_fits(int32_t M, int32_t N, int32_t K, int32_t HW) {
    return (M > 0 && N > 0 && K > 0 && HW > 0 && M % GBM == 0 && HW % GBM == 0 && M % HW == 0 && N % GBN == 0 && (K % 256 == 0 || K == 128)) ? 1
                                                                                                                                         : 0;
}

// out = [GroupNorm](x) . W^T (+ bias) (+ residual); mean_rstd / gamma / beta NULL = no normalisation; gn_partial (optional):
// [B][HW / 64][32][2] fp64 partial statistics of `out` (needs N % 128 == 0 as always, N / 32 a power of two <= 32)
static int gemm_panel_impl(const float *x, int32_t lda, const float *mean_rstd, const float *gamma,
                           const float *beta, const void *w_planes, float w_scale, const float *bias, const float *residual, int32_t ldr,
                           float *out, int32_t ldc, double *gn_partial, int32_t M, int32_t N, int32_t K, int32_t HW,
                           void *stream) {
    const bool gn = mean_rstd != nullptr;
    if (!x || !w_planes || !out || sgam_gemm_gn_f32x_fits(M, N, K, HW) != 1 || lda < K || ldc < N || !(w_scale > 0.f) ||
        (gn && (!gamma || !beta)) || (residual && ldr < N))
        return SGAM_EINVAL;
    const int cpo = N / 32;
    if (gn_partial && (cpo > 32 || (cpo & (cpo - 1)))) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || (gn && (!sgam_aligned16(gamma) || !sgam_aligned16(beta))) || !sgam_aligned16(w_planes) || lda % 4)
        return SGAM_EALIGN;
    GemmGnParams p;
    p.x = x; p.w = (const unsigned short *)w_planes; p.bias = bias; p.res = residual; p.ldr = ldr; p.out = out; p.gn_partial = gn_partial;
    p.mean_rstd = mean_rstd; p.gamma = gamma; p.beta = beta;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.HW = HW; p.inv_w_scale = 1.0f / w_scale; p.range_flag = sgam_i_range_flag;
    if (sgam_i_prof_on) sgam_i_prof_shape(M, N, K, 1);
    if (sgam_i_prof_on) sgam_i_prof_work(2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    const dim3 grid(M / GBM, N / GBN);
    hipStream_t s = sgam_stream(stream);
    if (gn && K == 128) SGAM_KLAUNCH((gemm_gn_f32x_kernel<true, 128>), grid, dim3(256), 0, s, p);   // C = 128 AttnBlocks (ch * ch_mult = 128 levels)
    else if (gn) SGAM_KLAUNCH((gemm_gn_f32x_kernel<true, 256>), grid, dim3(256), 0, s, p);
    else if (K == 128) SGAM_KLAUNCH((gemm_gn_f32x_kernel<false, 128>), grid, dim3(256), 0, s, p);
    else SGAM_KLAUNCH((gemm_gn_f32x_kernel<false, 256>), grid, dim3(256), 0, s, p);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_gemm_panel_f32x(const float *x, int32_t lda, const float *mean_rstd, const float *gamma, const float *beta,
                                    const void *w_planes, float w_scale, const float *bias, const float *residual, int32_t ldr, float *out,
                                    int32_t ldc, double *gn_partial, int32_t M, int32_t N, int32_t K, int32_t HW, void *stream) {
    return gemm_panel_impl(x, lda, mean_rstd, gamma, beta, w_planes, w_scale, bias, residual, ldr, out, ldc, gn_partial, M, N, K, HW, stream);
}

extern "C" int sgam_gemm_gn_f32x(const float *x, int32_t lda, const float *mean_rstd, const float *gamma, const float *beta,
                                 const void *w_planes, float w_scale, const float *bias, float *out, int32_t ldc, int32_t M, int32_t N,
                                 int32_t K, int32_t HW, void *stream) {
    if (!mean_rstd) return SGAM_EINVAL;
    return sgam_gemm_panel_f32x(x, lda, mean_rstd, gamma, beta, w_planes, w_scale, bias, nullptr, 0, out, ldc, nullptr, M, N, K, HW, stream);
}
