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
#include "sgam_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int GBM = 64, GBN = 128, KC = 256, LDK = KC + 8;        // LDS row pitch in halfs: 528 B, rows 4 banks apart

struct GemmGnParams {
    const float *x;            // [M][lda]
    const unsigned short *w;   // fragment-ordered hi / lo planes: [N / 32][K / 32][256 pieces][8 halfs]
    const float *bias;         // [N] or NULL
    float *out;                // [M][ldc]
    const float *mean_rstd;    // [B][32][2]
    const float *gamma, *beta; // [K]
    int M, N, K, lda, ldc, HW; // HW rows per image (GroupNorm statistics are per image)
    float inv_w_scale;
    int32_t *range_flag;
};

__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void gemm_gn_f32x_kernel(const GemmGnParams p) {
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][GBM][LDK];         // hi plane, lo plane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * GBM, n0 = blockIdx.y * GBN + wave * 32;
    const int b = m0 / p.HW;                                       // host guarantees HW % 64 == 0: a tile lies in one image
    const int cpg = p.K / 32;                                      // channels per group (32 groups)
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

    for (int k0 = 0; k0 < p.K; k0 += KC) {
        if (k0) __syncthreads();                                    // the previous panel has been consumed
        // ---- stage the 64 x KC panel: normalise, split, store.  Thread -> (row, 4 consecutive channels): a half-wave of 64
        // threads covers one row (256 channels), so global reads are whole rows and LDS writes are conflict free
#pragma unroll 4
        for (int it = 0; it < GBM * (KC / 4) / 256; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx >> 6, c4 = (idx & 63) * 4;
            const int c = k0 + c4;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(p.x + (int64_t)(m0 + row) * p.lda + c);
            const int g = c / cpg;
            const float mean = p.mean_rstd[(b * 32 + g) * 2], rstd = p.mean_rstd[(b * 32 + g) * 2 + 1];
            const f32x4 ga = *reinterpret_cast<const f32x4 *>(p.gamma + c), be = *reinterpret_cast<const f32x4 *>(p.beta + c);
            unsigned hi[2], lo[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float a0 = (v[2 * e] - mean) * rstd * ga[2 * e] + be[2 * e];
                const float a1 = (v[2 * e + 1] - mean) * rstd * ga[2 * e + 1] + be[2 * e + 1];
                const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
                const _Float16 l0 = (_Float16)(a0 - (float)h0), l1 = (_Float16)(a1 - (float)h1);
                hi[e] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                lo[e] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
            }
            *reinterpret_cast<unsigned long long *>(&sA[0][row][c4]) = (unsigned long long)hi[0] | ((unsigned long long)hi[1] << 32);
            *reinterpret_cast<unsigned long long *>(&sA[1][row][c4]) = (unsigned long long)lo[0] | ((unsigned long long)lo[1] << 32);
        }
        __syncthreads();
        // ---- 16 k-steps of 16: A fragments from LDS (row = 32 i + lr, k = 16 t + 8 lh + 0..7), weights one step ahead
        u32x4 wh[2], wl[2];
        wfrag(k0 / 16, wh[0], wl[0]);
#pragma unroll
        for (int t = 0; t < KC / 16; ++t) {
            if (t + 1 < KC / 16) wfrag(k0 / 16 + t + 1, wh[(t + 1) & 1], wl[(t + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4 ah = *reinterpret_cast<const u32x4 *>(&sA[0][i * 32 + lr][t * 16 + lh * 8]);
                const u32x4 al = *reinterpret_cast<const u32x4 *>(&sA[1][i * 32 + lr][t * 16 + lh * 8]);
                acc[i] = mfma16(ah, wh[t & 1], acc[i]);
                acc[i] = mfma16(ah, wl[t & 1], acc[i]);
                acc[i] = mfma16(al, wh[t & 1], acc[i]);
            }
        }
    }
    // ---- epilogue: lane = column n0 + lr, rows 8 (e / 4) + 4 lh + e % 4 of each 32-row tile; a half-wave writes 128 B of a row
    const int n = n0 + lr;
    const float bias = p.bias ? p.bias[n] : 0.f;
    float chk = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = m0 + i * 32 + 8 * (e >> 2) + 4 * lh + (e & 3);
            const float v = acc[i][e] * p.inv_w_scale + bias;
            p.out[(int64_t)row * p.ldc + n] = v;
            chk += v;
        }
    if (p.range_flag && sgam_not_finite(chk)) atomicOr(p.range_flag, 1);          // an operand left fp16's range
}

}  // namespace

// 1 when (M, N, K, HW) fit the kernel: whole 64-row tiles inside one image, whole 128-column tiles, K a multiple of 256
extern "C" int32_t sgam_gemm_gn_f32x_fits(int32_t M, int32_t N, int32_t K, int32_t HW) {
    return (M > 0 && N > 0 && K > 0 && HW > 0 && M % GBM == 0 && HW % GBM == 0 && M % HW == 0 && N % GBN == 0 && K % KC == 0) ? 1 : 0;
}

extern "C" int sgam_gemm_gn_f32x(const float *x, int32_t lda, const float *mean_rstd, const float *gamma, const float *beta,
                                 const void *w_planes, float w_scale, const float *bias, float *out, int32_t ldc, int32_t M, int32_t N,
                                 int32_t K, int32_t HW, void *stream) {
    if (!x || !mean_rstd || !gamma || !beta || !w_planes || !out || sgam_gemm_gn_f32x_fits(M, N, K, HW) != 1 || lda < K || ldc < N ||
        !(w_scale > 0.f))
        return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(gamma) || !sgam_aligned16(beta) || !sgam_aligned16(w_planes) || lda % 4) return SGAM_EALIGN;
    GemmGnParams p;
    p.x = x; p.w = (const unsigned short *)w_planes; p.bias = bias; p.out = out; p.mean_rstd = mean_rstd; p.gamma = gamma; p.beta = beta;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.HW = HW; p.inv_w_scale = 1.0f / w_scale; p.range_flag = sgam_i_range_flag;
    if (sgam_i_prof_on) sgam_i_prof_shape(M, N, K, 1);
    if (sgam_i_prof_on) sgam_i_prof_work(2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    SGAM_KLAUNCH(gemm_gn_f32x_kernel, dim3(M / GBM, N / GBN), dim3(256), 0, sgam_stream(stream), p);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}
