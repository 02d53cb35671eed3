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
#include <stdlib.h>

#ifndef SGAM_SCHED
#define SGAM_SCHED 0
#endif

#include "sgam_common.h"

namespace {

struct ConvKernelParams {
    const float *x, *w, *bias, *res;
    float *out;
    float *ws;  // split-K partials or nullptr
    const float *gn;  // fused GroupNorm prologue: per-(batch, input channel) {scale, shift} pairs, or nullptr
    int gn_swish;
    unsigned x_bytes, w_bytes;  // extents of the A / B operands for the bounds-checked buffer loads
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


// ------------------------------------------------------------------------------------------------
// v2 mainloop: same tiling / LDS image / MFMA order as above, but
//   * operands are fetched with bounds-checked BUFFER loads: an out-of-image tap, an M/N tail row or a
//     K tail column gets an offset past the descriptor's extent and the hardware returns zeros — no
//     exec-masked branch regions in the instruction stream (v1 had ~30 branches per K slab);
//   * the slab for step t+1 is requested at the top of step t, and written to the OTHER LDS buffer in
//     the middle of step t's MFMA stream (after half of the MFMAs), so neither the address arithmetic
//     nor the ds_writes sit on the critical path between two barriers;
//   * optional fused GroupNorm(+swish) prologue: v = swish(v*scale[c] + shift[c]) applied to the A slab
//     between load and LDS store (zero padding stays zero) — removes the standalone normalise pass.
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

template <int BM, int BN, bool GN, int BKT>
__global__ __launch_bounds__(256) void conv_gemm_f32_v2_kernel(const ConvKernelParams p) {
    // BKT = K slab width per barrier (32 or 64 floats); LDS rows are BKT + 4 floats
    constexpr int BK = BKT;
    constexpr int LDSLD = BK + 4;
    constexpr int C4 = BK / 4;             // float4 columns per slab row
    constexpr int RPP = 256 / C4;          // rows staged per pass of the 256 threads
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int AR = BM / RPP, BR = BN / RPP;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDSLD];
    float *As = smem;
    float *Bs = smem + 2 * BM * LDSLD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)p.w_bytes, 0x00020000);
    const unsigned gn_bytes = GN ? (unsigned)(p.B * p.Cin * 2) * 4u : 0u;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)p.gn, 0, (int)gn_bytes, 0x00020000);

    const int it0 = blockIdx.z * p.iters_per_split;
    const int it1 = min(p.iters_total, it0 + p.iters_per_split);

    const int col4 = tid % C4;
    const int row_in_pass = tid / C4;
    const int Hl = p.ups ? 2 * p.Hi : p.Hi;
    const int Wl = p.ups ? 2 * p.Wi : p.Wi;

    int a_iy0[AR], a_ix0[AR], a_base[AR], a_gn[AR];
#pragma unroll
    for (int r = 0; r < AR; ++r) {
        const int m = m0 + row_in_pass + RPP * r;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int hw = p.Ho * p.Wo;
        const int b = mm / hw;
        const int rem = mm - b * hw;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        // rows past M get a coordinate that fails every bounds test
        a_iy0[r] = ok ? oy * p.stride - p.pad_t : -(1 << 28);
        a_ix0[r] = ox * p.stride - p.pad_l;
        a_base[r] = b * p.Hi * p.Wi;
        a_gn[r] = b * p.Cin * 2;
    }
    unsigned b_off[BR];
#pragma unroll
    for (int r = 0; r < BR; ++r) {
        const int n = n0 + row_in_pass + RPP * r;
        b_off[r] = n < p.N ? (unsigned)(n * p.ldb + col4 * 4) * 4u : 0xC0000000u;  // + koff stays out of range
    }

    // K walk of this split, channel-slab major: it -> (ch = it / taps, tap = it % taps).  All taps of one
    // 32-channel slab are visited back to back (the GroupNorm table of the slab is fetched once per slab and the
    // 9 shifted reads of the same channels hit L1/L2); no divisions inside the loop.
    const int taps = p.KH * p.KW;
    int ch = it0 / taps;
    int tap = it0 - ch * taps;
    int ky = tap / p.KW;
    int kx = tap - ky * p.KW;
    bool gn_reload = true;

    f32x4 areg[AR], breg[BR];
    f32x4 gsc[GN ? AR : 1][2];
    unsigned amask = 0;  // which staged A rows are real data (padding / tails must stay exactly zero)

    // select without control flow (the compiler otherwise builds exec-masked regions around the address math)
    auto sel = [](bool c, unsigned a, unsigned b) -> unsigned {
        const unsigned m = 0u - (unsigned)c;
        return (a & m) | (b & ~m);
    };
    auto issue_loads = [&](bool live) {
        const int coff = ch * BK + col4 * 4;
        const bool k_ok = live && coff < p.Cin;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const int iy = a_iy0[r] + ky, ix = a_ix0[r] + kx;
            const bool ok = k_ok && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;
            const int py = iy >> p.ups, px = ix >> p.ups;
            const unsigned off = (unsigned)((a_base[r] + py * p.Wi + px) * p.lda + coff) * 4u;
            areg[r] = buf_load4(rx, sel(ok, off, p.x_bytes));
            amask = (amask & ~(1u << r)) | ((ok ? 1u : 0u) << r);
        }
        if constexpr (GN) {
            if (gn_reload) {  // wave-uniform: first slab of the split or a new channel slab
#pragma unroll
                for (int r = 0; r < AR; ++r) {
                    // {sc0,sh0,sc1,sh1},{sc2,sh2,sc3,sh3}; past Cin the table reads as scale = shift = 0
                    const unsigned goff = (unsigned)(a_gn[r] + coff * 2) * 4u;
                    gsc[r][0] = buf_load4(rg, sel(k_ok, goff, gn_bytes));
                    gsc[r][1] = buf_load4(rg, sel(k_ok, goff + 16u, gn_bytes));
                }
            }
        }
        const unsigned koff = (unsigned)(tap * p.Cin + ch * BK) * 4u;
#pragma unroll
        for (int r = 0; r < BR; ++r) breg[r] = buf_load4(rw, sel(k_ok, b_off[r] + koff, p.w_bytes));
        // advance the walk
        gn_reload = false;
        ++tap;
        if (++kx == p.KW) {
            kx = 0;
            ++ky;
        }
        if (tap == taps) {
            tap = 0; ky = 0; kx = 0;
            ++ch;
            gn_reload = true;
        }
    };
    auto store_lds = [&](int buf) {
        float *a = As + buf * BM * LDSLD;
        float *b = Bs + buf * BN * LDSLD;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            f32x4 v = areg[r];
            if constexpr (GN) {
                const float live = (amask >> r) & 1u ? 1.0f : 0.0f;  // zero padding stays exactly zero
                v[0] = (v[0] * gsc[r][0][0] + gsc[r][0][1]) * live;
                v[1] = (v[1] * gsc[r][0][2] + gsc[r][0][3]) * live;
                v[2] = (v[2] * gsc[r][1][0] + gsc[r][1][1]) * live;
                v[3] = (v[3] * gsc[r][1][2] + gsc[r][1][3]) * live;
                if (p.gn_swish) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = sgam_swish(v[e]);
                }
            }
            *reinterpret_cast<f32x4 *>(a + (row_in_pass + RPP * r) * LDSLD + col4 * 4) = v;
        }
#pragma unroll
        for (int r = 0; r < BR; ++r)
            *reinterpret_cast<f32x4 *>(b + (row_in_pass + RPP * r) * LDSLD + col4 * 4) = breg[r];
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

    issue_loads(it0 < it1);
    store_lds(0);
    __syncthreads();

    for (int it = it0; it < it1; ++it) {
        const int buf = (it - it0) & 1;
        const float *a = As + buf * BM * LDSLD + (wm * (BM / 2) + frag_row) * LDSLD + frag_k;
        const float *b = Bs + buf * BN * LDSLD + (wn * (BN / 2) + frag_row) * LDSLD + frag_k;
        // straight-line body: past the last slab the loads are all out of range (zeros) and the LDS store is
        // a harmless write to the buffer nobody reads again
        issue_loads((it + 1) < it1);
#if SGAM_SCHED == 2
        // ask the scheduler for an MFMA-paced interleave: each 64-cycle MFMA shadows a few VALU/SALU/DS/VMEM issues
#pragma unroll
        for (int q = 0; q < TM * TN * 16; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);  // up to 3 VALU/SALU
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // up to 1 DS read
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // up to 1 VMEM read
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // up to 1 DS write
        }
#endif
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            if (g == BK / 16) {
#if SGAM_SCHED == 0
                if constexpr (!GN) __builtin_amdgcn_sched_barrier(0);
#endif
                store_lds(buf ^ 1);  // the other buffer: last read before the previous barrier
#if SGAM_SCHED == 0
                if constexpr (!GN) __builtin_amdgcn_sched_barrier(0);
#endif
            }
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
        __syncthreads();
    }

    // ---- epilogue: D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
    // Bias / residual come in through bounds-checked buffer loads and results leave through bounds-checked
    // buffer stores (out-of-range lanes are dropped by the hardware): no per-element branches.
    const int col_l = lane & 31;
    const int row_h = 4 * (lane >> 5);
    const bool to_ws = p.ws != nullptr;
    const int n_lim = to_ws ? p.N : p.n_valid;
    const int ldo = to_ws ? p.N : p.ldc;
    float *obase = to_ws ? p.ws + (int64_t)blockIdx.z * p.M * p.N : p.out;
    const unsigned o_bytes = (unsigned)(((int64_t)(p.M - 1) * ldo + n_lim) * 4);
    const unsigned r_bytes = (p.res && !to_ws) ? (unsigned)(((int64_t)(p.M - 1) * p.ldr + p.n_valid) * 4) : 0u;
    const unsigned bias_bytes = (p.bias && !to_ws) ? (unsigned)((p.bias_per_row ? p.M : p.N) * 4) : 0u;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)obase, 0, (int)o_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)p.res, 0, (int)r_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)p.bias, 0, (int)bias_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + col_l;
            const bool n_ok = n < n_lim;
            const float bias_n = p.bias_per_row ? 0.f
                                                : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                      rb, (int)sel(n_ok, (unsigned)n * 4u, OOB), 0, 0));
            float rv[16], bv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + row_h;
                const bool ok = n_ok && m < p.M;
                rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                      rr, (int)sel(ok, (unsigned)(m * p.ldr + n) * 4u, OOB), 0, 0));
                bv[e] = p.bias_per_row ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                       rb, (int)sel(ok, (unsigned)m * 4u, OOB), 0, 0))
                                       : bias_n;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + row_h;
                const bool ok = n_ok && m < p.M;
                const float v = (acc[i][j][e] + bv[e]) + rv[e];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro,
                                                      (int)sel(ok, (unsigned)(m * ldo + n) * 4u, OOB), 0, 0);
            }
        }
    }
}


template <int BM, int BN, bool GN>
__global__ __launch_bounds__(512) void conv_gemm_f32_v3_kernel(const ConvKernelParams p) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int AR = BM / 32, BR = BN / 32;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDSLD];
    float *As = smem;
    float *Bs = smem + 2 * BM * LDSLD;

    // 8 wavefronts: 0-3 feed the matrix cores (ds_read + MFMA only), 4-7 stream the next K slab HBM/L2 -> LDS.
    // Each SIMD hosts one wave of each role, so address arithmetic, buffer loads and ds_writes issue from a
    // different wave than the MFMAs and never sit in the MFMA wave's in-order instruction stream.
    const bool producer = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >= 256;
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)p.w_bytes, 0x00020000);
    const unsigned gn_bytes = GN ? (unsigned)(p.B * p.Cin * 2) * 4u : 0u;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)p.gn, 0, (int)gn_bytes, 0x00020000);

    const int it0 = blockIdx.z * p.iters_per_split;
    const int it1 = min(p.iters_total, it0 + p.iters_per_split);

    const int col4 = tid & 7;
    const int row_in_pass = tid >> 3;
    const int Hl = p.ups ? 2 * p.Hi : p.Hi;
    const int Wl = p.ups ? 2 * p.Wi : p.Wi;

    int a_iy0[AR], a_ix0[AR], a_base[AR], a_gn[AR];
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
        // rows past M get a coordinate that fails every bounds test
        a_iy0[r] = ok ? oy * p.stride - p.pad_t : -(1 << 28);
        a_ix0[r] = ox * p.stride - p.pad_l;
        a_base[r] = b * p.Hi * p.Wi;
        a_gn[r] = b * p.Cin * 2;
    }
    unsigned b_off[BR];
#pragma unroll
    for (int r = 0; r < BR; ++r) {
        const int n = n0 + row_in_pass + 32 * r;
        b_off[r] = n < p.N ? (unsigned)(n * p.ldb + col4 * 4) * 4u : 0xC0000000u;  // + koff stays out of range
    }

    // K walk of this split, channel-slab major: it -> (ch = it / taps, tap = it % taps).  All taps of one
    // 32-channel slab are visited back to back (the GroupNorm table of the slab is fetched once per slab and the
    // 9 shifted reads of the same channels hit L1/L2); no divisions inside the loop.
    const int taps = p.KH * p.KW;
    int ch = it0 / taps;
    int tap = it0 - ch * taps;
    int ky = tap / p.KW;
    int kx = tap - ky * p.KW;
    bool gn_reload = true;

    f32x4 areg[AR], breg[BR];
    f32x4 gsc[GN ? AR : 1][2];
    unsigned amask = 0;  // which staged A rows are real data (padding / tails must stay exactly zero)

    // select without control flow (the compiler otherwise builds exec-masked regions around the address math)
    auto sel = [](bool c, unsigned a, unsigned b) -> unsigned {
        const unsigned m = 0u - (unsigned)c;
        return (a & m) | (b & ~m);
    };
    auto issue_loads = [&](bool live) {
        const int coff = ch * BK + col4 * 4;
        const bool k_ok = live && coff < p.Cin;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const int iy = a_iy0[r] + ky, ix = a_ix0[r] + kx;
            const bool ok = k_ok && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;
            const int py = iy >> p.ups, px = ix >> p.ups;
            const unsigned off = (unsigned)((a_base[r] + py * p.Wi + px) * p.lda + coff) * 4u;
            areg[r] = buf_load4(rx, sel(ok, off, p.x_bytes));
            amask = (amask & ~(1u << r)) | ((ok ? 1u : 0u) << r);
        }
        if constexpr (GN) {
            if (gn_reload) {  // wave-uniform: first slab of the split or a new channel slab
#pragma unroll
                for (int r = 0; r < AR; ++r) {
                    // {sc0,sh0,sc1,sh1},{sc2,sh2,sc3,sh3}; past Cin the table reads as scale = shift = 0
                    const unsigned goff = (unsigned)(a_gn[r] + coff * 2) * 4u;
                    gsc[r][0] = buf_load4(rg, sel(k_ok, goff, gn_bytes));
                    gsc[r][1] = buf_load4(rg, sel(k_ok, goff + 16u, gn_bytes));
                }
            }
        }
        const unsigned koff = (unsigned)(tap * p.Cin + ch * BK) * 4u;
#pragma unroll
        for (int r = 0; r < BR; ++r) breg[r] = buf_load4(rw, sel(k_ok, b_off[r] + koff, p.w_bytes));
        // advance the walk
        gn_reload = false;
        ++tap;
        if (++kx == p.KW) {
            kx = 0;
            ++ky;
        }
        if (tap == taps) {
            tap = 0; ky = 0; kx = 0;
            ++ch;
            gn_reload = true;
        }
    };
    auto store_lds = [&](int buf) {
        float *a = As + buf * BM * LDSLD;
        float *b = Bs + buf * BN * LDSLD;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            f32x4 v = areg[r];
            if constexpr (GN) {
                const float live = (amask >> r) & 1u ? 1.0f : 0.0f;  // zero padding stays exactly zero
                v[0] = (v[0] * gsc[r][0][0] + gsc[r][0][1]) * live;
                v[1] = (v[1] * gsc[r][0][2] + gsc[r][0][3]) * live;
                v[2] = (v[2] * gsc[r][1][0] + gsc[r][1][1]) * live;
                v[3] = (v[3] * gsc[r][1][2] + gsc[r][1][3]) * live;
                if (p.gn_swish) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = sgam_swish(v[e]);
                }
            }
            *reinterpret_cast<f32x4 *>(a + (row_in_pass + 32 * r) * LDSLD + col4 * 4) = v;
        }
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

    if (producer) {
        issue_loads(it0 < it1);
        store_lds(0);
    }
    __syncthreads();

    for (int it = it0; it < it1; ++it) {
        const int buf = (it - it0) & 1;
        if (producer) {
            // slab it+1 -> the other LDS buffer (its last readers passed the previous barrier)
            if ((it + 1) < it1) {
                issue_loads(true);
                store_lds(buf ^ 1);
            }
        } else {
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
        }
        __syncthreads();
    }
    if (producer) return;

    // ---- epilogue: D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
    // Bias / residual come in through bounds-checked buffer loads and results leave through bounds-checked
    // buffer stores (out-of-range lanes are dropped by the hardware): no per-element branches.
    const int col_l = lane & 31;
    const int row_h = 4 * (lane >> 5);
    const bool to_ws = p.ws != nullptr;
    const int n_lim = to_ws ? p.N : p.n_valid;
    const int ldo = to_ws ? p.N : p.ldc;
    float *obase = to_ws ? p.ws + (int64_t)blockIdx.z * p.M * p.N : p.out;
    const unsigned o_bytes = (unsigned)(((int64_t)(p.M - 1) * ldo + n_lim) * 4);
    const unsigned r_bytes = (p.res && !to_ws) ? (unsigned)(((int64_t)(p.M - 1) * p.ldr + p.n_valid) * 4) : 0u;
    const unsigned bias_bytes = (p.bias && !to_ws) ? (unsigned)((p.bias_per_row ? p.M : p.N) * 4) : 0u;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)obase, 0, (int)o_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)p.res, 0, (int)r_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)p.bias, 0, (int)bias_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + col_l;
            const bool n_ok = n < n_lim;
            const float bias_n = p.bias_per_row ? 0.f
                                                : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                      rb, (int)sel(n_ok, (unsigned)n * 4u, OOB), 0, 0));
            float rv[16], bv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + row_h;
                const bool ok = n_ok && m < p.M;
                rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                      rr, (int)sel(ok, (unsigned)(m * p.ldr + n) * 4u, OOB), 0, 0));
                bv[e] = p.bias_per_row ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                       rb, (int)sel(ok, (unsigned)m * 4u, OOB), 0, 0))
                                       : bias_n;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + row_h;
                const bool ok = n_ok && m < p.M;
                const float v = (acc[i][j][e] + bv[e]) + rv[e];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro,
                                                      (int)sel(ok, (unsigned)(m * ldo + n) * 4u, OOB), 0, 0);
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
    int bm, bn, bk, ksplit, iters_total, iters_per_split;
};

Plan make_plan(const sgam_conv_desc *d) {
    const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
    Plan pl;
    pl.bk = BK;
    auto blocks = [&](int bm, int bn) { return (int64_t)sgam_cdiv(M, bm) * sgam_cdiv(d->N, bn); };
    if (d->N % 128 == 0 && blocks(128, 128) >= 224) {
        pl.bm = 128; pl.bn = 128;
        static const int bk128 = [] { const char *e = getenv("SGAM_BK128"); return e ? atoi(e) : 32; }();
        if (d->Cin % 64 == 0) pl.bk = bk128;   // wider K slab: half the barriers per MFMA (one workgroup per CU)
    } else if (d->N % 128 == 0 && blocks(64, 128) >= 224) {
        pl.bm = 64; pl.bn = 128;
    } else {
        pl.bm = 64; pl.bn = 64;
    }
    if (d->plan_bm > 0 && d->plan_bn > 0) {   // autotuned override
        pl.bm = d->plan_bm; pl.bn = d->plan_bn; pl.bk = BK;
    }
    pl.iters_total = d->KH * d->KW * ((d->Cin + pl.bk - 1) / pl.bk);
    const int64_t nb = blocks(pl.bm, pl.bn);
    int ks = 1;
    if (d->plan_ksplit > 0) {
        ks = d->plan_ksplit;
        if (ks > pl.iters_total) ks = pl.iters_total;
    } else if (nb < 192) {
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
    if (d->plan_bm != 0 || d->plan_bn != 0) {
        const bool ok = (d->plan_bm == 128 && d->plan_bn == 128) || (d->plan_bm == 64 && d->plan_bn == 128) ||
                        (d->plan_bm == 64 && d->plan_bn == 64);
        if (!ok || (d->plan_bn == 128 && d->N % 128 != 0)) return SGAM_EINVAL;
    }
    if (d->plan_ksplit < 0 || d->plan_ksplit > 64) return SGAM_EINVAL;
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

static bool use_v1() {
    static const int v = [] { const char *e = getenv("SGAM_CONV_V1"); return (e && e[0] == '1') ? 1 : 0; }();
    return v != 0;
}

static int conv_variant() {  // 2 = single-role v2 (default), 3 = producer/consumer wave specialisation (measured slower)
    static const int v = [] { const char *e = getenv("SGAM_CONV_VARIANT"); return e ? atoi(e) : 2; }();
    return v;
}

template <bool GN>
static void launch_v3(const Plan &pl, const dim3 &grid, hipStream_t s, const ConvKernelParams &p) {
    if (pl.bm == 128 && pl.bn == 128) {
        SGAM_KLAUNCH((conv_gemm_f32_v3_kernel<128, 128, GN>), grid, dim3(512), 0, s, p);
    } else if (pl.bm == 64 && pl.bn == 128) {
        SGAM_KLAUNCH((conv_gemm_f32_v3_kernel<64, 128, GN>), grid, dim3(512), 0, s, p);
    } else {
        SGAM_KLAUNCH((conv_gemm_f32_v3_kernel<64, 64, GN>), grid, dim3(512), 0, s, p);
    }
}

template <bool GN>
static void launch_v2(const Plan &pl, const dim3 &grid, hipStream_t s, const ConvKernelParams &p) {
    if (pl.bm == 128 && pl.bn == 128 && pl.bk == 64) {
        SGAM_KLAUNCH((conv_gemm_f32_v2_kernel<128, 128, GN, 64>), grid, dim3(256), 0, s, p);
    } else if (pl.bm == 128 && pl.bn == 128) {
        SGAM_KLAUNCH((conv_gemm_f32_v2_kernel<128, 128, GN, 32>), grid, dim3(256), 0, s, p);
    } else if (pl.bm == 64 && pl.bn == 128) {
        SGAM_KLAUNCH((conv_gemm_f32_v2_kernel<64, 128, GN, 32>), grid, dim3(256), 0, s, p);
    } else {
        SGAM_KLAUNCH((conv_gemm_f32_v2_kernel<64, 64, GN, 32>), grid, dim3(256), 0, s, p);
    }
}

extern "C" int sgam_conv2d_gn_nhwc_f32(const sgam_conv_desc *d, const float *x, const float *gn_scale_shift,
                                       int32_t gn_swish, const float *w_packed, const float *bias,
                                       const float *residual, float *out, void *workspace, int64_t workspace_bytes,
                                       void *stream) {
    const int rc = validate(d);
    if (rc != SGAM_OK) return rc;
    if (!x || !w_packed || !out) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(w_packed)) return SGAM_EALIGN;
    if (gn_scale_shift && !sgam_aligned16(gn_scale_shift)) return SGAM_EALIGN;
    const Plan pl = make_plan(d);
    ConvKernelParams p;
    p.x = x; p.w = w_packed; p.bias = bias; p.res = residual; p.out = out; p.ws = nullptr;
    p.gn = gn_scale_shift; p.gn_swish = gn_swish;
    p.B = d->B; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.N = d->N;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.ups = d->upsample2x ? 1 : 0;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr; p.n_valid = d->n_valid;
    p.bias_per_row = d->bias_per_row;
    p.M = d->B * d->Ho * d->Wo;
    p.ksplit = pl.ksplit; p.iters_total = pl.iters_total; p.iters_per_split = pl.iters_per_split;
    const int64_t xb = (((int64_t)d->B * d->Hi * d->Wi - 1) * d->lda + d->Cin) * 4;
    const int64_t wb = (((int64_t)d->N - 1) * d->ldb + (int64_t)d->KH * d->KW * d->Cin) * 4;
    if (xb >= (1ll << 32) - 64 || wb >= (1ll << 32) - 64) return SGAM_EINVAL;  // 32-bit buffer offsets
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
    if (pl.ksplit > 1) {
        const int64_t need = (int64_t)pl.ksplit * p.M * p.N * (int64_t)sizeof(float);
        if (!workspace || workspace_bytes < need || !sgam_aligned16(workspace)) return SGAM_EWORKSPACE;
        p.ws = (float *)workspace;
    }
    const dim3 grid(sgam_cdiv(p.M, pl.bm), sgam_cdiv(p.N, pl.bn), pl.ksplit);
    hipStream_t s = sgam_stream(stream);
    if (sgam_i_prof_on) sgam_i_prof_shape(p.M, d->n_valid, d->KH * d->KW * d->Cin, pl.ksplit);
    if (sgam_i_prof_on)
        sgam_i_prof_work(2.0 * p.M * d->n_valid * (double)(d->KH * d->KW * d->Cin),
                         4.0 * ((double)d->B * d->Hi * d->Wi * d->Cin + (double)d->n_valid * d->KH * d->KW * d->Cin +
                                (double)p.M * d->n_valid));
    if (use_v1() && !gn_scale_shift) {
        if (pl.bm == 128 && pl.bn == 128) {
            SGAM_KLAUNCH((conv_gemm_f32_kernel<128, 128>), grid, dim3(256), 0, s, p);
        } else if (pl.bm == 64 && pl.bn == 128) {
            SGAM_KLAUNCH((conv_gemm_f32_kernel<64, 128>), grid, dim3(256), 0, s, p);
        } else {
            SGAM_KLAUNCH((conv_gemm_f32_kernel<64, 64>), grid, dim3(256), 0, s, p);
        }
    } else if (gn_scale_shift) {
        if (conv_variant() == 3) launch_v3<true>(pl, grid, s, p); else launch_v2<true>(pl, grid, s, p);
    } else {
        if (conv_variant() == 3) launch_v3<false>(pl, grid, s, p); else launch_v2<false>(pl, grid, s, p);
    }
    SGAM_LAUNCH_CHECK();
    if (pl.ksplit > 1) {
        const int64_t q = (int64_t)p.M * (p.N / 4);
        SGAM_KLAUNCH(splitk_reduce_kernel, dim3(sgam_cdiv(q, 256)), dim3(256), 0, s, p);
        SGAM_LAUNCH_CHECK();
    }
    return SGAM_OK;
}

extern "C" int sgam_conv2d_nhwc_f32(const sgam_conv_desc *d, const float *x, const float *w_packed,
                                    const float *bias, const float *residual, float *out, void *workspace,
                                    int64_t workspace_bytes, void *stream) {
    return sgam_conv2d_gn_nhwc_f32(d, x, nullptr, 0, w_packed, bias, residual, out, workspace, workspace_bytes, stream);
}

extern "C" int sgam_pack_conv_weight(const float *w_oihw, float *w_packed, int32_t Cout, int32_t Cin, int32_t KH,
                                     int32_t KW, int32_t Cout_pad, int32_t Cin_pad, void *stream) {
    if (!w_oihw || !w_packed || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || Cout_pad < Cout || Cin_pad < Cin)
        return SGAM_EINVAL;
    const int64_t total = (int64_t)Cout_pad * KH * KW * Cin_pad;
    SGAM_KLAUNCH(pack_weight_kernel, dim3(sgam_cdiv(total, 256)), dim3(256), 0, sgam_stream(stream), w_oihw,
                       w_packed, Cout, Cin, KH, KW, Cout_pad, Cin_pad);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}
