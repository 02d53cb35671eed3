// h16.hip — the 16-bit throughput path of the VQGAN: bf16 (default) or fp16 activations and weights, fp32
// accumulation on v_mfma_f32_32x32x16_{bf16,f16} (16x the matrix rate of the fp32-in MFMA used by the parity path).
//
// Same implicit-GEMM design as conv_gemm.hip (NHWC, K = (tap, channel), bounds-checked buffer loads, double-
// buffered padded LDS slabs, split-K with a fixed-order reduction) with these differences:
//   * a K slab is 64 halfs = 128 bytes per row, so the LDS image ([rows][144 B]) and the 16-byte staging pattern
//     are byte-for-byte those of the fp32 kernel; one ds_read_b128 (8 halfs) is exactly one MFMA A/B operand:
//     lane l feeds row l&31, k = 8*(l>>5)..+7 of a 32x32x16 step;
//   * outputs are rounded once to the 16-bit type (RNE, v_cvt_pk_bf16_f32 / v_cvt_f16_f32) or kept in fp32
//     (attention scores, the latent fed to the fp32 quantiser, the final RGB-D);
//   * GroupNorm statistics, softmax, bias and residual arithmetic stay in fp32.
// `ht` selects the type at run time: 0 = bf16, 1 = fp16.
#include <stdlib.h>
#include <type_traits>

#include "sgam_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int HT> struct H;
template <> struct H<0> {
    __device__ static __forceinline__ float to_f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
    __device__ static __forceinline__ unsigned short from_f(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
    __device__ static __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct H<1> {
    __device__ static __forceinline__ float to_f(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
    __device__ static __forceinline__ unsigned short from_f(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
    __device__ static __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

struct H16Params {
    const unsigned short *x, *w, *res;
    const float *bias;
    void *out;      // 16-bit or fp32 (out_f32)
    float *ws;      // split-K partials or nullptr
    int B, Hi, Wi, Cin, Ho, Wo, N, KH, KW, stride, pad_t, pad_l, ups;
    int lda, ldb, ldc, ldr, n_valid, bias_per_row, out_f32;
    int M, ksplit, iters_total, iters_per_split;
    unsigned x_bytes, w_bytes;
    double *gn_partial;   // optional (direct epilogue, whole-K workgroups): per-(BM / 2 rows, group) {sum, sumsq} of the output
    int gn_cpg;           // channels per group of the output (N / 32)
};

constexpr int HBK = 64;          // halfs per K slab
constexpr int HLD = HBK + 8;     // LDS row stride in halfs (144 bytes)

__device__ __forceinline__ unsigned selu(bool c, unsigned a, unsigned b) {
    const unsigned m = 0u - (unsigned)c;
    return (a & m) | (b & ~m);
}

// DIR (the default whenever n_valid % 4 == 0 and the row strides are even): the product is computed TRANSPOSED (weights = MFMA
// rows, pixels = columns) with the weight rows of every 32-channel tile stored to LDS in the order that makes a lane's sixteen
// accumulator slots SIXTEEN CONSECUTIVE CHANNELS of one pixel (the h16_halo.hip epilogue): 16-byte stores, 8-byte residual
// loads, bias / residual / rounding in registers, and — optionally — the GroupNorm statistics of the output as per-chunk
// partial sums, so that a 1x1 / strided convolution or the attention block's proj_out feeds the next normalisation without a
// statistics pass.  !DIR: pixels = MFMA rows, one 2- or 4-byte store per accumulator slot (any n_valid, any stride).
template <int BM, int BN, int HT, bool DIR>
__global__ __launch_bounds__(256) void conv_gemm_h16_kernel(const H16Params p) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int AR = BM / 32, BR = BN / 32;
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * (BM + BN) * HLD];
    unsigned short *As = smem;
    unsigned short *Bs = smem + 2 * BM * HLD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)p.w_bytes, 0x00020000);

    const int it0 = blockIdx.z * p.iters_per_split;
    const int it1 = min(p.iters_total, it0 + p.iters_per_split);

    const int col8 = tid & 7;          // 16-byte chunk (8 halfs) within the 128-byte slab row
    const int row_in_pass = tid >> 3;  // 32 rows per pass
    const int Hl = p.ups ? 2 * p.Hi : p.Hi;
    const int Wl = p.ups ? 2 * p.Wi : p.Wi;

    int a_iy0[AR], a_ix0[AR], a_base[AR];
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
    }
    unsigned b_off[BR];
#pragma unroll
    for (int r = 0; r < BR; ++r) {
        const int n = n0 + row_in_pass + 32 * r;
        b_off[r] = n < p.N ? (unsigned)(n * p.ldb + col8 * 8) * 2u : 0xC0000000u;
    }

    const int taps = p.KH * p.KW;
    int ch = it0 / taps;
    int tap = it0 - ch * taps;
    int ky = tap / p.KW;
    int kx = tap - ky * p.KW;

    // two register stages: a slab requested in step t is written to LDS in step t+1 and consumed in step t+2,
    // so every buffer load has TWO K steps (not one) to come back — with 16-bit operands a step is only
    // 16 MFMAs (~0.5k cycles) per wavefront, far less than an L2 round trip.
    u32x4 areg[2][AR], breg[2][BR];
    auto issue_loads = [&](bool live, auto stage_tag) {
        constexpr int ST = decltype(stage_tag)::value;
        const int coff = ch * HBK + col8 * 8;
        const bool k_ok = live && coff < p.Cin;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const int iy = a_iy0[r] + ky, ix = a_ix0[r] + kx;
            const bool ok = k_ok && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;
            const int py = iy >> p.ups, px = ix >> p.ups;
            const unsigned off = (unsigned)((a_base[r] + py * p.Wi + px) * p.lda + coff) * 2u;
            areg[ST][r] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)selu(ok, off, p.x_bytes), 0, 0);
        }
        const unsigned koff = (unsigned)(tap * p.Cin + ch * HBK) * 2u;
#pragma unroll
        for (int r = 0; r < BR; ++r)
            breg[ST][r] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)selu(k_ok, b_off[r] + koff, p.w_bytes), 0, 0);
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
    auto store_lds = [&](int buf, auto stage_tag) {
        constexpr int ST = decltype(stage_tag)::value;
        unsigned short *a = As + buf * BM * HLD;
        unsigned short *b = Bs + buf * BN * HLD;
#pragma unroll
        for (int r = 0; r < AR; ++r) *reinterpret_cast<u32x4 *>(a + (row_in_pass + 32 * r) * HLD + col8 * 8) = areg[ST][r];
        // DIR: channel c = 16 half + e of a 32-channel tile sits in LDS row 8 (e / 4) + 4 half + e % 4 = the MFMA row whose
        // accumulator slot e belongs to lane half `half`
        const int brow = DIR ? 8 * ((row_in_pass & 15) >> 2) + 4 * (row_in_pass >> 4) + (row_in_pass & 3) : row_in_pass;
#pragma unroll
        for (int r = 0; r < BR; ++r) *reinterpret_cast<u32x4 *>(b + (brow + 32 * r) * HLD + col8 * 8) = breg[ST][r];
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 8;

    auto mfma_half = [&](int buf, int kk0) {
        const unsigned short *a = As + buf * BM * HLD + (wm * (BM / 2) + frag_row) * HLD + frag_k;
        const unsigned short *b = Bs + buf * BN * HLD + (wn * (BN / 2) + frag_row) * HLD + frag_k;
#pragma unroll
        for (int kk = kk0; kk < kk0 + HBK / 32; ++kk) {
            u32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const u32x4 *>(a + i * 32 * HLD + kk * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const u32x4 *>(b + j * 32 * HLD + kk * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = DIR ? H<HT>::mfma(bf[j], af[i], acc[i][j]) : H<HT>::mfma(af[i], bf[j], acc[i][j]);
        }
    };

    int nleft = it1 - it0;                 // slabs not yet requested
    issue_loads(nleft-- > 0, S0{});        // slab 0 -> stage 0
    store_lds(0, S0{});
    issue_loads(nleft-- > 0, S1{});        // slab 1 -> stage 1 (in flight across the barrier)
    __syncthreads();

    // slabs past the end are requested out of range (zeros) and contribute nothing: the loop can always run in pairs
    for (int it = it0; it < it1; it += 2) {
        // even step: LDS buffer 0 holds slab `it`; stage 1 holds slab it+1 (in flight); request slab it+2 into stage 0
        issue_loads(nleft-- > 0, S0{});
        mfma_half(0, 0);
        store_lds(1, S1{});
        mfma_half(0, HBK / 32);
        __syncthreads();
        // odd step: buffer 1 holds slab it+1; stage 0 holds slab it+2 (in flight); request slab it+3 into stage 1
        issue_loads(nleft-- > 0, S1{});
        mfma_half(1, 0);
        store_lds(0, S0{});
        mfma_half(1, HBK / 32);
        __syncthreads();
    }

    // ---- epilogue (branch-free: bounds-checked buffer loads / stores) ----
    const int col_l = lane & 31;
    const int row_h = 4 * (lane >> 5);
    const bool to_ws = p.ws != nullptr;
    const int n_lim = to_ws ? p.N : p.n_valid;
    const int ldo = to_ws ? p.N : p.ldc;
    const bool f32o = to_ws || p.out_f32;
    void *obase = to_ws ? (void *)(p.ws + (int64_t)blockIdx.z * p.M * p.N) : p.out;
    const unsigned osz = f32o ? 4u : 2u;
    const unsigned o_bytes = (unsigned)(((int64_t)(p.M - 1) * ldo + n_lim) * osz);
    const unsigned r_bytes = (p.res && !to_ws) ? (unsigned)(((int64_t)(p.M - 1) * p.ldr + p.n_valid) * 2) : 0u;
    const unsigned bias_bytes = (p.bias && !to_ws) ? (unsigned)((p.bias_per_row ? p.M : p.N) * 4) : 0u;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (int)o_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)p.res, 0, (int)r_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)p.bias, 0, (int)bias_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    if constexpr (DIR) {
        // lane (pixel = lane & 31, half = lane >> 5) holds channels 16 half + e, e = 0..15, of every (row tile, channel tile)
        const int pl = lane & 31, hh = lane >> 5;
        const int wn0 = n0 + wn * (BN / 2);
        float us[TN][4], uss[TN][4];          // per 4-channel unit: sum, sum of squares over this lane's pixels
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = wn0 + j * 32 + hh * 16;
            f32x4 bv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bv[k] = p.bias_per_row ? f32x4{0.f, 0.f, 0.f, 0.f}
                                       : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                       rb, (int)selu(nb + 4 * k < n_lim, (unsigned)(nb + 4 * k) * 4u, OOB), 0, 0));
                us[j][k] = uss[j][k] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm * (BM / 2) + i * 32 + pl;
                const bool m_ok = m < p.M;
                const float brow_v = p.bias_per_row ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                                   rb, (int)selu(m_ok, (unsigned)m * 4u, OOB), 0, 0))
                                                    : 0.f;
                u32x4 o16[2];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int n4 = nb + 4 * k;
                    const bool ok = m_ok && n4 < n_lim;
                    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
                    const u32x2_t rq = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(
                                                                       rr, (int)selu(ok, (unsigned)(m * p.ldr + n4) * 2u, OOB), 0, 0));
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (acc[i][j][4 * k + e] + (bv[k][e] + brow_v));
                    v[0] += H<HT>::to_f((unsigned short)(rq[0] & 0xFFFFu));
                    v[1] += H<HT>::to_f((unsigned short)(rq[0] >> 16));
                    v[2] += H<HT>::to_f((unsigned short)(rq[1] & 0xFFFFu));
                    v[3] += H<HT>::to_f((unsigned short)(rq[1] >> 16));
                    if (f32o) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro,
                                                               (int)selu(ok, (unsigned)(m * ldo + n4) * 4u, OOB), 0, 0);
                    } else {
                        const unsigned short h0 = H<HT>::from_f(v[0]), h1 = H<HT>::from_f(v[1]), h2 = H<HT>::from_f(v[2]),
                                             h3 = H<HT>::from_f(v[3]);
                        o16[k >> 1][(k & 1) * 2] = (unsigned)h0 | ((unsigned)h1 << 16);
                        o16[k >> 1][(k & 1) * 2 + 1] = (unsigned)h2 | ((unsigned)h3 << 16);
                        // the statistics describe the STORED (rounded) tensor: that is what the next GroupNorm normalises
                        v = f32x4{H<HT>::to_f(h0), H<HT>::to_f(h1), H<HT>::to_f(h2), H<HT>::to_f(h3)};
                    }
                    if (ok) {
                        us[j][k] += (v[0] + v[1]) + (v[2] + v[3]);
                        uss[j][k] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                    }
                }
                if (!f32o) {
                    if (nb + 16 <= n_lim) {               // the usual case: two 16-byte stores
#pragma unroll
                        for (int q2 = 0; q2 < 2; ++q2)
                            __builtin_amdgcn_raw_buffer_store_b128(o16[q2], ro, (int)selu(m_ok, (unsigned)(m * ldo + nb + 8 * q2) * 2u, OOB), 0,
                                                                   0);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
                            const u32x2_t o8 = {o16[k >> 1][(k & 1) * 2], o16[k >> 1][(k & 1) * 2 + 1]};
                            __builtin_amdgcn_raw_buffer_store_b64(
                                o8, ro, (int)selu(m_ok && nb + 4 * k < n_lim, (unsigned)(m * ldo + nb + 4 * k) * 2u, OOB), 0, 0);
                        }
                    }
                }
            }
        }
        if (p.gn_partial && !to_ws) {
            // over the 32 pixels of a lane half (xor shuffles stay inside it), then lane 0 of each half leaves its 4 x TN units;
            // chunk = BM / 2 consecutive rows (the host guarantees that a chunk lies inside one image and that N == n_valid)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int off = 16; off >= 1; off >>= 1) {
                        us[j][k] += __shfl_xor(us[j][k], off, 64);
                        uss[j][k] += __shfl_xor(uss[j][k], off, 64);
                    }
            __syncthreads();                   // (the operand buffers are dead: every wavefront is past its last MFMA)
            float *sl = reinterpret_cast<float *>(smem) + wave * (2 * TN * 8);      // wave-private: [unit = 8 j + 4 half + k][2]
            if (pl == 0) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        sl[(j * 8 + hh * 4 + k) * 2] = us[j][k];
                        sl[(j * 8 + hh * 4 + k) * 2 + 1] = uss[j][k];
                    }
            }
            const int c4_per_group = p.gn_cpg / 4;
            const int groups_here = (TN * 8) / c4_per_group;
            if (lane < groups_here) {
                double ds = 0.0, dss = 0.0;
                for (int k = 0; k < c4_per_group; ++k) {
                    ds += (double)sl[(lane * c4_per_group + k) * 2];
                    dss += (double)sl[(lane * c4_per_group + k) * 2 + 1];
                }
                const int g = (wn0 / p.gn_cpg) + lane;
                const int groups = p.N / p.gn_cpg;
                const int chunk = blockIdx.x * 2 + wm;          // image-major: chunks of an image are consecutive
                if (g < groups && (int64_t)chunk * (BM / 2) < p.M) {
                    double *o = p.gn_partial + ((int64_t)chunk * groups + g) * 2;
                    o[0] = ds;
                    o[1] = dss;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + col_l;
            const bool n_ok = n < n_lim;
            const float bias_n = p.bias_per_row ? 0.f
                                                : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                      rb, (int)selu(n_ok, (unsigned)n * 4u, OOB), 0, 0));
            float rv[16], bv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + row_h;
                const bool ok = n_ok && m < p.M;
                rv[e] = H<HT>::to_f(__builtin_amdgcn_raw_buffer_load_b16(rr, (int)selu(ok, (unsigned)(m * p.ldr + n) * 2u, OOB), 0, 0));
                bv[e] = p.bias_per_row ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                       rb, (int)selu(ok, (unsigned)m * 4u, OOB), 0, 0))
                                       : bias_n;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + row_h;
                const bool ok = n_ok && m < p.M;
                const float v = (acc[i][j][e] + bv[e]) + rv[e];
                if (f32o) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro,
                                                          (int)selu(ok, (unsigned)(m * ldo + n) * 4u, OOB), 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b16(H<HT>::from_f(v), ro, (int)selu(ok, (unsigned)(m * ldo + n) * 2u, OOB), 0, 0);
                }
            }
        }
    }
}

template <int HT>
__global__ __launch_bounds__(256) void splitk_reduce_h16_kernel(const H16Params p) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = p.N / 4;
    if (q >= (int64_t)p.M * nq) return;
    const int m = (int)(q / nq);
    const int n = (int)(q - (int64_t)m * nq) * 4;
    f32x4 s = *reinterpret_cast<const f32x4 *>(p.ws + (int64_t)m * p.N + n);
    for (int z = 1; z < p.ksplit; ++z) s += *reinterpret_cast<const f32x4 *>(p.ws + ((int64_t)z * p.M + m) * p.N + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (n + e >= p.n_valid) continue;
        float v = s[e];
        if (p.bias) v += p.bias_per_row ? p.bias[m] : p.bias[n + e];
        if (p.res) v += H<HT>::to_f(p.res[(int64_t)m * p.ldr + n + e]);
        if (p.out_f32) ((float *)p.out)[(int64_t)m * p.ldc + n + e] = v;
        else ((unsigned short *)p.out)[(int64_t)m * p.ldc + n + e] = H<HT>::from_f(v);
    }
}

template <int HT>
__global__ void pack_weight_h16_kernel(const float *w, unsigned short *o, int Cout, int Cin, int KH, int KW, int Cout_pad,
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
    o[i] = H<HT>::from_f(v);
}

template <int HT>
__global__ void cast_f32_to_h16_kernel(const float *x, unsigned short *y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = H<HT>::from_f(x[i]);
}
template <int HT>
__global__ void cast_h16_to_f32_kernel(const unsigned short *x, float *y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = H<HT>::to_f(x[i]);
}

// softmax over fp32 scores, probabilities written in the 16-bit type (the A operand of the P.V GEMM)
template <int HT, int MAXV>
__global__ __launch_bounds__(256) void softmax_rows_h16_kernel(const float *__restrict__ s, unsigned short *__restrict__ pout,
                                                               int cols, int lds, int ldp, float scale, int block) {
    const float *row = s + (int64_t)blockIdx.x * lds;
    unsigned short *orow = pout + (int64_t)blockIdx.x * ldp;
    const int c4 = cols >> 2;
    // block > 0: block-diagonal form (norm_softmax.hip): the columns outside the row's own block become exact zeros
    const int lo4 = block ? ((int)blockIdx.x / block) * (block >> 2) : 0, hi4 = block ? lo4 + (block >> 2) : c4;
    f32x4 v[MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = threadIdx.x + k * 256;
        if (i < c4) {
            if (i >= lo4 && i < hi4) {
                v[k] = reinterpret_cast<const f32x4 *>(row)[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[k][e] *= scale; mx = fmaxf(mx, v[k][e]); }
            } else {
                v[k] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
        }
    }
    __shared__ float red[4];
    __shared__ float bc;
    mx = sgam_wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    mx = bc;
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = threadIdx.x + k * 256;
        if (i < c4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[k][e] = __expf(v[k][e] - mx); sum += v[k][e]; }
        }
    }
    sum = sgam_wave_sum(sum);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) bc = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    const float inv = 1.0f / bc;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = threadIdx.x + k * 256;
        if (i < c4) {
            const unsigned lo = (unsigned)H<HT>::from_f(v[k][0] * inv) | ((unsigned)H<HT>::from_f(v[k][1] * inv) << 16);
            const unsigned hi = (unsigned)H<HT>::from_f(v[k][2] * inv) | ((unsigned)H<HT>::from_f(v[k][3] * inv) << 16);
            reinterpret_cast<uint2 *>(orow)[i] = make_uint2(lo, hi);
        }
    }
}

// VQModel.encode head with 16-bit NHWC output (pixel row padded to ldy halfs)
template <int HT>
__global__ __launch_bounds__(256) void encode_head_h16_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask,
                                                              const float *__restrict__ w, const float *__restrict__ bias,
                                                              unsigned short *__restrict__ y, int HW, int ldy) {
    const int b = blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= HW) return;
    const float *xb = x + (int64_t)b * 4 * HW;
    float in[5];
#pragma unroll
    for (int c = 0; c < 4; ++c) in[c] = xb[(int64_t)c * HW + pix];
    in[4] = mask ? (mask[(int64_t)b * HW + pix] ? 1.0f : 0.0f) : 0.0f;
    unsigned short o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        float acc = bias[n];
#pragma unroll
        for (int c = 0; c < 5; ++c) acc = fmaf(in[c], w[n * 5 + c], acc);
        o[n] = H<HT>::from_f(acc);
    }
    unsigned short *row = y + ((int64_t)b * HW + pix) * ldy;
    reinterpret_cast<uint2 *>(row)[0] = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
    for (int k = 1; k < (ldy >> 2); ++k) reinterpret_cast<uint2 *>(row)[k] = make_uint2(0u, 0u);
}

// [HW][ld] 16-bit -> [C][HW] 16-bit (v^T for the P.V GEMM)
template <int HT>
__global__ __launch_bounds__(256) void transpose_h16_kernel(const unsigned short *__restrict__ x, unsigned short *__restrict__ y,
                                                            int C, int HW, int ldx) {
    __shared__ unsigned short tile[32][34];
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pix = p0 + ty + 8 * r, c = c0 + tx;
        tile[ty + 8 * r][tx] = (pix < HW && c < C) ? x[(int64_t)pix * ldx + c] : (unsigned short)0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + ty + 8 * r, pix = p0 + tx;
        if (c < C && pix < HW) y[(int64_t)c * HW + pix] = tile[tx][ty + 8 * r];
    }
}

struct HPlan {
    int bm, bn, ksplit, iters_total, iters_per_split;
};

HPlan make_hplan(const sgam_conv_desc *d) {
    const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
    HPlan pl;
    pl.iters_total = d->KH * d->KW * ((d->Cin + HBK - 1) / HBK);
    auto blocks = [&](int bm, int bn) { return (int64_t)sgam_cdiv(M, bm) * sgam_cdiv(d->N, bn); };
    if (d->N % 128 == 0 && blocks(128, 128) >= 224) { pl.bm = 128; pl.bn = 128; }
    else if (d->N % 128 == 0 && blocks(64, 128) >= 224) { pl.bm = 64; pl.bn = 128; }
    else { pl.bm = 64; pl.bn = 64; }
    if (d->plan_bm > 0 && d->plan_bn > 0) { pl.bm = d->plan_bm; pl.bn = d->plan_bn; }   // autotuned override
    if (pl.bm == 256) pl.bm = 128;            // (256 rows: a tile of the halo-staged kernel only, h16_halo.hip)
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
    pl.ksplit = (pl.iters_total + pl.iters_per_split - 1) / pl.iters_per_split;
    return pl;
}

int hvalidate(const sgam_conv_desc *d) {
    if (!d) return SGAM_EINVAL;
    if (d->B <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->N <= 0) return SGAM_EINVAL;
    if (d->Cin <= 0 || d->Cin % 8 != 0 || d->N % 4 != 0) return SGAM_EINVAL;
    if (d->KH <= 0 || d->KW <= 0 || d->stride <= 0) return SGAM_EINVAL;
    if (d->lda < d->Cin || d->lda % 8 != 0) return SGAM_EALIGN;
    if (d->ldb < d->KH * d->KW * d->Cin || d->ldb % 8 != 0) return SGAM_EALIGN;
    if (d->n_valid <= 0 || d->n_valid > d->N || d->ldc < d->n_valid) return SGAM_EINVAL;
    if (d->plan_bm != 0 || d->plan_bn != 0) {
        const bool ok = (d->plan_bm == 256 && d->plan_bn == 128) || (d->plan_bm == 128 && d->plan_bn == 128) ||
                        (d->plan_bm == 64 && d->plan_bn == 128) || (d->plan_bm == 64 && d->plan_bn == 64);
        if (!ok || (d->plan_bn == 128 && d->N % 128 != 0)) return SGAM_EINVAL;
    }
    if (d->plan_ksplit < 0 || d->plan_ksplit > 64) return SGAM_EINVAL;
    return SGAM_OK;
}

// the direct (transposed-product) epilogue needs 4-channel units and dword-aligned 8- / 16-byte accesses
bool h16_direct_ok(const sgam_conv_desc *d) {
    return d->n_valid % 4 == 0 && d->ldc % 2 == 0 && d->ldr % 2 == 0;
}

// chunks of output statistics per image the generic kernel can leave (0 = none): whole-K workgroups, the direct epilogue, no
// padded channels, 32 groups of 4 ... 32 channels that do not straddle a wavefront's channel range, chunks of BM / 2 rows
// that do not straddle an image
int h16_generic_chunks(const sgam_conv_desc *d) {
    if (hvalidate(d) != SGAM_OK || !h16_direct_ok(d) || d->n_valid != d->N || d->N % 128 != 0 || d->bias_per_row) return 0;
    const HPlan pl = make_hplan(d);
    const int cpg = d->N / 32, hw = d->Ho * d->Wo;
    if (pl.ksplit != 1 || cpg % 4 != 0 || (pl.bn / 2) % cpg != 0 || hw % (pl.bm / 2) != 0) return 0;
    return hw / (pl.bm / 2);
}

template <int HT>
int conv_h16_launch(const sgam_conv_desc *d, const void *x, const void *w, const float *bias, const void *res, void *out,
                    int out_f32, void *workspace, int64_t workspace_bytes, hipStream_t s, double *gn_partial = nullptr) {
    const HPlan pl = make_hplan(d);
    H16Params p;
    p.x = (const unsigned short *)x; p.w = (const unsigned short *)w; p.res = (const unsigned short *)res; p.bias = bias;
    p.out = out; p.ws = nullptr;
    p.B = d->B; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.N = d->N;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.ups = d->upsample2x ? 1 : 0;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr; p.n_valid = d->n_valid; p.bias_per_row = d->bias_per_row;
    p.out_f32 = out_f32;
    p.M = d->B * d->Ho * d->Wo;
    p.ksplit = pl.ksplit; p.iters_total = pl.iters_total; p.iters_per_split = pl.iters_per_split;
    const int64_t xb = (((int64_t)d->B * d->Hi * d->Wi - 1) * d->lda + d->Cin) * 2;
    const int64_t wb = (((int64_t)d->N - 1) * d->ldb + (int64_t)d->KH * d->KW * d->Cin) * 2;
    if (xb >= (1ll << 32) - 64 || wb >= (1ll << 32) - 64) return SGAM_EINVAL;
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
    p.gn_partial = gn_partial; p.gn_cpg = d->N / 32;
    if (gn_partial && h16_generic_chunks(d) <= 0) return SGAM_EINVAL;
    const bool dir = h16_direct_ok(d);
    if (pl.ksplit > 1) {
        const int64_t need = (int64_t)pl.ksplit * p.M * p.N * (int64_t)sizeof(float);
        if (!workspace || workspace_bytes < need || !sgam_aligned16(workspace)) return SGAM_EWORKSPACE;
        p.ws = (float *)workspace;
    }
    const dim3 grid(sgam_cdiv(p.M, pl.bm), sgam_cdiv(p.N, pl.bn), pl.ksplit);
    if (sgam_i_prof_on) sgam_i_prof_shape(p.M, d->n_valid, d->KH * d->KW * d->Cin, pl.ksplit);
    if (sgam_i_prof_on)
        sgam_i_prof_work(2.0 * p.M * d->n_valid * (double)(d->KH * d->KW * d->Cin),
                         2.0 * ((double)d->B * d->Hi * d->Wi * d->Cin + (double)d->n_valid * d->KH * d->KW * d->Cin +
                                (double)p.M * d->n_valid));
#define H16_LAUNCH(DIR_)                                                                                              \
    do {                                                                                                              \
        if (pl.bm == 128 && pl.bn == 128) SGAM_KLAUNCH((conv_gemm_h16_kernel<128, 128, HT, DIR_>), grid, dim3(256), 0, s, p); \
        else if (pl.bm == 64 && pl.bn == 128) SGAM_KLAUNCH((conv_gemm_h16_kernel<64, 128, HT, DIR_>), grid, dim3(256), 0, s, p); \
        else SGAM_KLAUNCH((conv_gemm_h16_kernel<64, 64, HT, DIR_>), grid, dim3(256), 0, s, p);                          \
    } while (0)
    if (dir) H16_LAUNCH(true);
    else H16_LAUNCH(false);
#undef H16_LAUNCH
    SGAM_LAUNCH_CHECK();
    if (pl.ksplit > 1) {
        const int64_t q = (int64_t)p.M * (p.N / 4);
        SGAM_KLAUNCH(splitk_reduce_h16_kernel<HT>, dim3(sgam_cdiv(q, 256)), dim3(256), 0, s, p);
        SGAM_LAUNCH_CHECK();
    }
    return SGAM_OK;
}

}  // namespace

#define HT_DISPATCH(ht, CALL0, CALL1) \
    do {                              \
        if ((ht) == 0) { CALL0; }     \
        else if ((ht) == 1) { CALL1; } \
        else return SGAM_EINVAL;      \
    } while (0)

extern "C" int64_t sgam_conv2d_h16_workspace_bytes(const sgam_conv_desc *d) {
    if (hvalidate(d) != SGAM_OK) return -1;
    const HPlan pl = make_hplan(d);
    if (pl.ksplit <= 1) return 0;
    return (int64_t)pl.ksplit * d->B * d->Ho * d->Wo * d->N * (int64_t)sizeof(float);
}

extern "C" int sgam_conv2d_h16_plan(const sgam_conv_desc *d, int32_t *bm, int32_t *bn, int32_t *ksplit) {
    const int rc = hvalidate(d);
    if (rc != SGAM_OK) return rc;
    const HPlan pl = make_hplan(d);
    if (bm) *bm = pl.bm;
    if (bn) *bn = pl.bn;
    if (ksplit) *ksplit = pl.ksplit;
    return SGAM_OK;
}

extern "C" int sgam_conv2d_nhwc_h16(const sgam_conv_desc *d, int32_t ht, const void *x, const void *w_packed,
                                    const float *bias, const void *residual, void *out, int32_t out_f32, void *workspace,
                                    int64_t workspace_bytes, void *stream) {
    const int rc = hvalidate(d);
    if (rc != SGAM_OK) return rc;
    if (!x || !w_packed || !out) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(w_packed)) return SGAM_EALIGN;
    hipStream_t s = sgam_stream(stream);
    HT_DISPATCH(ht, return conv_h16_launch<0>(d, x, w_packed, bias, residual, out, out_f32, workspace, workspace_bytes, s),
                return conv_h16_launch<1>(d, x, w_packed, bias, residual, out, out_f32, workspace, workspace_bytes, s));
    return SGAM_OK;
}

extern "C" int32_t sgam_conv2d_h16_generic_stats_chunks(const sgam_conv_desc *d) { return d ? h16_generic_chunks(d) : 0; }

extern "C" int sgam_conv2d_stats_nhwc_h16(const sgam_conv_desc *d, int32_t ht, const void *x, const void *w_packed,
                                          const float *bias, const void *residual, void *out, int32_t out_f32, double *gn_partial,
                                          void *workspace, int64_t workspace_bytes, void *stream) {
    const int rc = hvalidate(d);
    if (rc != SGAM_OK) return rc;
    if (!x || !w_packed || !out || !gn_partial) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(w_packed) || !sgam_aligned16(gn_partial)) return SGAM_EALIGN;
    hipStream_t s = sgam_stream(stream);
    HT_DISPATCH(ht, return conv_h16_launch<0>(d, x, w_packed, bias, residual, out, out_f32, workspace, workspace_bytes, s, gn_partial),
                return conv_h16_launch<1>(d, x, w_packed, bias, residual, out, out_f32, workspace, workspace_bytes, s, gn_partial));
    return SGAM_OK;
}

extern "C" int sgam_pack_conv_weight_h16(const float *w_oihw, void *w_packed, int32_t ht, int32_t Cout, int32_t Cin,
                                         int32_t KH, int32_t KW, int32_t Cout_pad, int32_t Cin_pad, void *stream) {
    if (!w_oihw || !w_packed || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || Cout_pad < Cout || Cin_pad < Cin) return SGAM_EINVAL;
    const int64_t total = (int64_t)Cout_pad * KH * KW * Cin_pad;
    const dim3 g(sgam_cdiv(total, 256));
    hipStream_t s = sgam_stream(stream);
    HT_DISPATCH(ht, SGAM_KLAUNCH(pack_weight_h16_kernel<0>, g, dim3(256), 0, s, w_oihw, (unsigned short *)w_packed, Cout, Cin, KH, KW, Cout_pad, Cin_pad),
                SGAM_KLAUNCH(pack_weight_h16_kernel<1>, g, dim3(256), 0, s, w_oihw, (unsigned short *)w_packed, Cout, Cin, KH, KW, Cout_pad, Cin_pad));
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_cast_f32_h16(const float *x, void *y, int32_t ht, int64_t n, void *stream) {
    if (!x || !y || n <= 0) return SGAM_EINVAL;
    hipStream_t s = sgam_stream(stream);
    HT_DISPATCH(ht, SGAM_KLAUNCH(cast_f32_to_h16_kernel<0>, dim3(sgam_cdiv(n, 256)), dim3(256), 0, s, x, (unsigned short *)y, n),
                SGAM_KLAUNCH(cast_f32_to_h16_kernel<1>, dim3(sgam_cdiv(n, 256)), dim3(256), 0, s, x, (unsigned short *)y, n));
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_cast_h16_f32(const void *x, float *y, int32_t ht, int64_t n, void *stream) {
    if (!x || !y || n <= 0) return SGAM_EINVAL;
    hipStream_t s = sgam_stream(stream);
    HT_DISPATCH(ht, SGAM_KLAUNCH(cast_h16_to_f32_kernel<0>, dim3(sgam_cdiv(n, 256)), dim3(256), 0, s, (const unsigned short *)x, y, n),
                SGAM_KLAUNCH(cast_h16_to_f32_kernel<1>, dim3(sgam_cdiv(n, 256)), dim3(256), 0, s, (const unsigned short *)x, y, n));
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

static int softmax_rows_h16_impl(const float *s_in, void *p_out, int32_t ht, int32_t rows, int32_t cols, int32_t lds,
                                 int32_t ldp, float scale, int32_t block, void *stream) {
    if (!s_in || !p_out || rows <= 0 || cols <= 0 || cols % 4 != 0 || lds < cols || ldp < cols || lds % 4 != 0 || ldp % 4 != 0)
        return SGAM_EINVAL;
    hipStream_t s = sgam_stream(stream);
    unsigned short *po = (unsigned short *)p_out;
#define SM_LAUNCH(HTV, MV) SGAM_KLAUNCH((softmax_rows_h16_kernel<HTV, MV>), dim3(rows), dim3(256), 0, s, s_in, po, cols, lds, ldp, scale, block)
    if (cols <= 1024) HT_DISPATCH(ht, SM_LAUNCH(0, 1), SM_LAUNCH(1, 1));
    else if (cols <= 4096) HT_DISPATCH(ht, SM_LAUNCH(0, 4), SM_LAUNCH(1, 4));
    else if (cols <= 16384) HT_DISPATCH(ht, SM_LAUNCH(0, 16), SM_LAUNCH(1, 16));
    else return SGAM_EINVAL;
#undef SM_LAUNCH
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_softmax_rows_h16(const float *s_in, void *p_out, int32_t ht, int32_t rows, int32_t cols, int32_t lds,
                                     int32_t ldp, float scale, void *stream) {
    return softmax_rows_h16_impl(s_in, p_out, ht, rows, cols, lds, ldp, scale, 0, stream);
}

extern "C" int sgam_softmax_rows_blockdiag_h16(const float *s_in, void *p_out, int32_t ht, int32_t rows, int32_t cols, int32_t lds,
                                               int32_t ldp, float scale, int32_t block, void *stream) {
    if (block <= 0 || block % 4 != 0 || rows % block != 0 || cols % block != 0 || rows / block > cols / block) return SGAM_EINVAL;
    return softmax_rows_h16_impl(s_in, p_out, ht, rows, cols, lds, ldp, scale, block, stream);
}

extern "C" int sgam_encode_head_h16(const float *x, const uint8_t *mask, const float *w, const float *bias, void *y, int32_t ht,
                                    int32_t B, int32_t HW, int32_t ldy, void *stream) {
    if (!x || !w || !bias || !y || B <= 0 || HW <= 0 || ldy < 4 || ldy % 8 != 0) return SGAM_EINVAL;
    hipStream_t s = sgam_stream(stream);
    HT_DISPATCH(ht, SGAM_KLAUNCH(encode_head_h16_kernel<0>, dim3(sgam_cdiv(HW, 256), B), dim3(256), 0, s, x, mask, w, bias, (unsigned short *)y, HW, ldy),
                SGAM_KLAUNCH(encode_head_h16_kernel<1>, dim3(sgam_cdiv(HW, 256), B), dim3(256), 0, s, x, mask, w, bias, (unsigned short *)y, HW, ldy));
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_transpose_h16(const void *x, void *y, int32_t ht, int32_t C, int32_t HW, int32_t ldx, void *stream) {
    if (!x || !y || C <= 0 || HW <= 0 || ldx < C) return SGAM_EINVAL;
    hipStream_t s = sgam_stream(stream);
    const dim3 g(sgam_cdiv(HW, 32), sgam_cdiv(C, 32));
    HT_DISPATCH(ht, SGAM_KLAUNCH(transpose_h16_kernel<0>, g, dim3(256), 0, s, (const unsigned short *)x, (unsigned short *)y, C, HW, ldx),
                SGAM_KLAUNCH(transpose_h16_kernel<1>, g, dim3(256), 0, s, (const unsigned short *)x, (unsigned short *)y, C, HW, ldx));
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}
