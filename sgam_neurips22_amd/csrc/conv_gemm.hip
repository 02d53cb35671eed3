// conv_gemm.hip — NHWC convolution / GEMM as an implicit GEMM on the CDNA4 matrix cores.
//
//   C[m][n] = sum_k A[m][k] * Bw[n][k]
//     m  <-> output pixel (b, oy, ox)            (GEMM M = B*Ho*Wo)
//     n  <-> output channel                       (GEMM N = Cout)
//     k  <-> (tap = ky*KW+kx, input channel c)    (GEMM K = KH*KW*Cin, c fastest)
//
// A is never materialised (no im2col buffer): every 32-wide K slab of the A tile is one filter tap
// of BM pixels x 32 contiguous NHWC channels, fetched as 128-byte rows straight from the feature
// map (with the zero padding, the stride-2 / asymmetric-pad Downsample and the nearest-2x Upsample
// of diffusionmodules/model.py:38-75 folded into the address computation).
//
// Matrix core: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, exact fp32 == an fmaf chain;
// 64 cycles per instruction per SIMD, 157 TFLOP/s chip peak).  A wavefront (64 lanes) owns a
// (BM/2)x(BN/2) sub-tile as TM x TN accumulators of 32x32; 4 wavefronts per workgroup (2x2).
//
// LDS: double-buffered [BM][36] + [BN][36] fp32 slabs (row padded 32->36 floats: 144-byte rows keep
// ds_read_b128 16-byte aligned and conflict-free for the 16-lane groups).  Within each group of
// 8 k-values lane-half h = lane>>5 owns k = 4h..4h+3, so ONE ds_read_b128 per operand tile feeds
// four consecutive MFMAs (the K order inside a slab is a free permutation as long as A and B agree).
//
// Split-K (gridDim.z) is used when M*N is too small to fill 256 CUs (the 16x16 / 32x32 maps):
// partial sums go to a workspace [S][M][N] and a second kernel reduces them in a FIXED order
// (deterministic: no float atomics anywhere) and applies bias + residual.
#include "sgam_common.h"

namespace {

struct ConvKernelParams {
    const float *x, *w, *bias, *res;
    float *out;
    float *ws;  // split-K partials or nullptr
    int B, Hi, Wi, Cin, Ho, Wo, N, KH, KW, stride, pad_t, pad_l, ups;
    int lda, ldb, ldc, ldr, n_valid, bias_per_row;
    int M, ksplit, iters_total, iters_per_split;
};

constexpr int BK = 32;
constexpr int LDSLD = BK + 4;

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_gemm_f32_kernel(const ConvKernelParams p) {
    constexpr int TM = BM / 64, TN = BN / 64;   // 32x32 accumulator tiles per wavefront
    constexpr int AR = BM / 32, BR = BN / 32;   // rows of the A / B slab each thread stages
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDSLD];
    float *As = smem;                    // [2][BM][LDSLD]
    float *Bs = smem + 2 * BM * LDSLD;   // [2][BN][LDSLD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    const int chunks = (p.Cin + BK - 1) / BK;  // the last K slab of a tap may be partial (Cin % 4 == 0)
    const int it0 = blockIdx.z * p.iters_per_split;
    const int it1 = min(p.iters_total, it0 + p.iters_per_split);

    // ---- staging assignment: thread -> (float4 column, rows) ----
    const int col4 = tid & 7;
    const int row_in_pass = tid >> 3;  // 0..31
    const int Hl = p.ups ? 2 * p.Hi : p.Hi;
    const int Wl = p.ups ? 2 * p.Wi : p.Wi;

    int a_iy0[AR], a_ix0[AR];
    int64_t a_base[AR];
    bool a_ok[AR];
#pragma unroll
    for (int r = 0; r < AR; ++r) {
        const int m = m0 + row_in_pass + 32 * r;
        a_ok[r] = m < p.M;
        const int mm = a_ok[r] ? m : 0;
        const int hw = p.Ho * p.Wo;
        const int b = mm / hw;
        const int rem = mm - b * hw;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        a_iy0[r] = oy * p.stride - p.pad_t;
        a_ix0[r] = ox * p.stride - p.pad_l;
        a_base[r] = (int64_t)b * p.Hi * p.Wi;
    }
    int64_t b_off[BR];
    bool b_ok[BR];
#pragma unroll
    for (int r = 0; r < BR; ++r) {
        const int n = n0 + row_in_pass + 32 * r;
        b_ok[r] = n < p.N;
        b_off[r] = (int64_t)(b_ok[r] ? n : 0) * p.ldb + col4 * 4;
    }

    f32x4 areg[AR], breg[BR];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_global = [&](int it) {
        const int tap = it / chunks;
        const int ch = it - tap * chunks;
        const int ky = tap / p.KW;
        const int kx = tap - ky * p.KW;
        const int coff = ch * BK + col4 * 4;
        const bool k_ok = coff < p.Cin;  // K tail: columns past Cin contribute zeros on both operands
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const int iy = a_iy0[r] + ky, ix = a_ix0[r] + kx;
            const bool ok = k_ok && a_ok[r] && iy >= 0 && iy < Hl && ix >= 0 && ix < Wl;
            const int py = p.ups ? (iy >> 1) : iy, px = p.ups ? (ix >> 1) : ix;
            const float *src = p.x + (a_base[r] + (int64_t)py * p.Wi + px) * p.lda + coff;
            areg[r] = ok ? *reinterpret_cast<const f32x4 *>(src) : zero4;
        }
        const int64_t koff = (int64_t)tap * p.Cin + ch * BK;
#pragma unroll
        for (int r = 0; r < BR; ++r) {
            breg[r] = (k_ok && b_ok[r]) ? *reinterpret_cast<const f32x4 *>(p.w + b_off[r] + koff) : zero4;
        }
    };
    auto store_lds = [&](int buf) {
        float *a = As + buf * BM * LDSLD;
        float *b = Bs + buf * BN * LDSLD;
#pragma unroll
        for (int r = 0; r < AR; ++r)
            *reinterpret_cast<f32x4 *>(a + (row_in_pass + 32 * r) * LDSLD + col4 * 4) = areg[r];
#pragma unroll
        for (int r = 0; r < BR; ++r)
            *reinterpret_cast<f32x4 *>(b + (row_in_pass + 32 * r) * LDSLD + col4 * 4) = breg[r];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;

    if (it0 < it1) {
        load_global(it0);
        store_lds(0);
    }
    __syncthreads();

    for (int it = it0; it < it1; ++it) {
        const int buf = (it - it0) & 1;
        const bool has_next = (it + 1) < it1;
        if (has_next) load_global(it + 1);  // global loads in flight under the MFMAs below

        const float *a = As + buf * BM * LDSLD + (wm * (BM / 2) + frag_row) * LDSLD + frag_k;
        const float *b = Bs + buf * BN * LDSLD + (wn * (BN / 2) + frag_row) * LDSLD + frag_k;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4 *>(a + i * 32 * LDSLD + g * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4 *>(b + j * 32 * LDSLD + g * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        if (has_next) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const int col_l = lane & 31;
    const int row_h = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + col_l;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + row_h;
                if (m >= p.M) continue;
                float v = acc[i][j][e];
                if (p.ws) {
                    if (n < p.N) p.ws[((int64_t)blockIdx.z * p.M + m) * p.N + n] = v;
                } else if (n < p.n_valid) {
                    if (p.bias) v += p.bias_per_row ? p.bias[m] : p.bias[n];
                    if (p.res) v += p.res[(int64_t)m * p.ldr + n];
                    p.out[(int64_t)m * p.ldc + n] = v;
                }
            }
        }
    }
}

// Fixed-order split-K reduction + bias + residual.  One thread per 4 output columns.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvKernelParams p) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = p.N / 4;
    if (q >= (int64_t)p.M * nq) return;
    const int m = (int)(q / nq);
    const int n = (int)(q - (int64_t)m * nq) * 4;
    f32x4 s = *reinterpret_cast<const f32x4 *>(p.ws + (int64_t)m * p.N + n);
    for (int z = 1; z < p.ksplit; ++z) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(p.ws + ((int64_t)z * p.M + m) * p.N + n);
        s += t;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (n + e >= p.n_valid) continue;
        float v = s[e];
        if (p.bias) v += p.bias_per_row ? p.bias[m] : p.bias[n + e];
        if (p.res) v += p.res[(int64_t)m * p.ldr + n + e];
        p.out[(int64_t)m * p.ldc + n + e] = v;
    }
}

// [Cout][Cin][KH][KW] -> [Cout_pad][KH*KW][Cin_pad]
__global__ void pack_weight_kernel(const float *w, float *o, int Cout, int Cin, int KH, int KW, int Cout_pad,
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
    o[i] = v;
}

struct Plan {
    int bm, bn, ksplit, iters_total, iters_per_split;
};

Plan make_plan(const sgam_conv_desc *d) {
    const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
    Plan pl;
    pl.iters_total = d->KH * d->KW * ((d->Cin + BK - 1) / BK);
    auto blocks = [&](int bm, int bn) { return (int64_t)sgam_cdiv(M, bm) * sgam_cdiv(d->N, bn); };
    if (d->N % 128 == 0 && blocks(128, 128) >= 224) {
        pl.bm = 128; pl.bn = 128;
    } else if (d->N % 128 == 0 && blocks(64, 128) >= 224) {
        pl.bm = 64; pl.bn = 128;
    } else {
        pl.bm = 64; pl.bn = 64;
    }
    const int64_t nb = blocks(pl.bm, pl.bn);
    int ks = 1;
    if (nb < 192) {
        ks = (int)((384 + nb - 1) / nb);              // aim for ~1.5 workgroups per CU
        const int max_by_iters = pl.iters_total / 4;  // keep >= 4 K-slabs per split
        if (ks > max_by_iters) ks = max_by_iters;
        if (ks > 32) ks = 32;
        if (ks < 1) ks = 1;
    }
    pl.iters_per_split = (pl.iters_total + ks - 1) / ks;
    pl.ksplit = (pl.iters_total + pl.iters_per_split - 1) / pl.iters_per_split;
    return pl;
}

int validate(const sgam_conv_desc *d) {
    if (!d) return SGAM_EINVAL;
    if (d->B <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->N <= 0) return SGAM_EINVAL;
    if (d->Cin <= 0 || d->Cin % 4 != 0) return SGAM_EINVAL;
    if (d->N % 4 != 0) return SGAM_EINVAL;
    if (d->KH <= 0 || d->KW <= 0 || d->stride <= 0) return SGAM_EINVAL;
    if (d->lda < d->Cin || d->lda % 4 != 0) return SGAM_EALIGN;
    if (d->ldb < d->KH * d->KW * d->Cin || d->ldb % 4 != 0) return SGAM_EALIGN;
    if (d->n_valid <= 0 || d->n_valid > d->N || d->ldc < d->n_valid) return SGAM_EINVAL;
    return SGAM_OK;
}

}  // namespace

extern "C" int64_t sgam_conv2d_workspace_bytes(const sgam_conv_desc *d) {
    if (validate(d) != SGAM_OK) return -1;
    const Plan pl = make_plan(d);
    if (pl.ksplit <= 1) return 0;
    return (int64_t)pl.ksplit * d->B * d->Ho * d->Wo * d->N * (int64_t)sizeof(float);
}

extern "C" int sgam_conv2d_plan(const sgam_conv_desc *d, int32_t *bm, int32_t *bn, int32_t *ksplit) {
    const int rc = validate(d);
    if (rc != SGAM_OK) return rc;
    const Plan pl = make_plan(d);
    if (bm) *bm = pl.bm;
    if (bn) *bn = pl.bn;
    if (ksplit) *ksplit = pl.ksplit;
    return SGAM_OK;
}

extern "C" int sgam_conv2d_nhwc_f32(const sgam_conv_desc *d, const float *x, const float *w_packed,
                                    const float *bias, const float *residual, float *out, void *workspace,
                                    int64_t workspace_bytes, void *stream) {
    const int rc = validate(d);
    if (rc != SGAM_OK) return rc;
    if (!x || !w_packed || !out) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(w_packed)) return SGAM_EALIGN;
    const Plan pl = make_plan(d);
    ConvKernelParams p;
    p.x = x; p.w = w_packed; p.bias = bias; p.res = residual; p.out = out; p.ws = nullptr;
    p.B = d->B; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.N = d->N;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.ups = d->upsample2x ? 1 : 0;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr; p.n_valid = d->n_valid;
    p.bias_per_row = d->bias_per_row;
    p.M = d->B * d->Ho * d->Wo;
    p.ksplit = pl.ksplit; p.iters_total = pl.iters_total; p.iters_per_split = pl.iters_per_split;
    if (pl.ksplit > 1) {
        const int64_t need = (int64_t)pl.ksplit * p.M * p.N * (int64_t)sizeof(float);
        if (!workspace || workspace_bytes < need || !sgam_aligned16(workspace)) return SGAM_EWORKSPACE;
        p.ws = (float *)workspace;
    }
    const dim3 grid(sgam_cdiv(p.M, pl.bm), sgam_cdiv(p.N, pl.bn), pl.ksplit);
    const size_t lds = 0;  // LDS is static per instantiation (73,728 B for 128x128: above the 64 KiB dynamic default)
    hipStream_t s = sgam_stream(stream);
    if (pl.bm == 128 && pl.bn == 128) {
        hipLaunchKernelGGL((conv_gemm_f32_kernel<128, 128>), grid, dim3(256), lds, s, p);
    } else if (pl.bm == 64 && pl.bn == 128) {
        hipLaunchKernelGGL((conv_gemm_f32_kernel<64, 128>), grid, dim3(256), lds, s, p);
    } else {
        hipLaunchKernelGGL((conv_gemm_f32_kernel<64, 64>), grid, dim3(256), lds, s, p);
    }
    SGAM_LAUNCH_CHECK();
    if (pl.ksplit > 1) {
        const int64_t q = (int64_t)p.M * (p.N / 4);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(sgam_cdiv(q, 256)), dim3(256), 0, s, p);
        SGAM_LAUNCH_CHECK();
    }
    return SGAM_OK;
}

extern "C" int sgam_pack_conv_weight(const float *w_oihw, float *w_packed, int32_t Cout, int32_t Cin, int32_t KH,
                                     int32_t KW, int32_t Cout_pad, int32_t Cin_pad, void *stream) {
    if (!w_oihw || !w_packed || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || Cout_pad < Cout || Cin_pad < Cin)
        return SGAM_EINVAL;
    const int64_t total = (int64_t)Cout_pad * KH * KW * Cin_pad;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(sgam_cdiv(total, 256)), dim3(256), 0, sgam_stream(stream), w_oihw,
                       w_packed, Cout, Cin, KH, KW, Cout_pad, Cin_pad);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}
