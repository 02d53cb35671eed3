// groupnorm.hip — GroupNorm(32 groups, eps, affine) with optional fused swish on NHWC maps, for fp32 (parity path)
// and bf16 / fp16 (throughput path) storage.  Statistics are always fp32 per lane -> fp64 across lanes, in a fixed
// order (no atomics): results are bit-reproducible run to run.  Replaces Normalize + nonlinearity
// (diffusionmodules/model.py:29-35, 119-127, 170, 429-430, 536-537).
//
// Two schedules, chosen by map size (both HBM/L2-bound: one read for statistics, one read + one write to apply):
//   small maps (HW <= 1024: the 16^2 and 32^2 levels, 32 of the 67 GroupNorms of a frame)
//       ONE launch, grid (groups, B): a workgroup owns one (batch, group): pass 1 sums its HW x cpg elements,
//       pass 2 re-reads them (L2 hits), normalises, applies swish and writes.
//   larger maps
//       gn_partial  (grid (nchunk <= 256, B)): per-chunk {sum, sumsq} of every group -> partial[b][chunk][g]
//       gn_finalize (grid B): folds the partials (256 lanes, fixed order) -> per-(b, channel) {scale, shift} table
//       gn_apply    : streams y = swish(x * scale + shift)
//   (measured on MI355X: folding the partials inside gn_apply with <= 64 chunks starves the statistics pass of
//    parallelism — 26.7 us vs ~8 us on the 33 MB maps — and a one-launch kernel only pays below ~1k pixels.)
// The statistics-only entry (sgam_groupnorm_stats_nhwc_f32) feeds the fused conv prologue with the same table.
#include "sgam_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- element traits: T = 2 fp32, 0 bf16, 1 fp16 ----
template <int T> struct E;
template <> struct E<2> {
    typedef float S;
    static constexpr int VEC = 4;  // elements per 16-byte access
    __device__ static __forceinline__ float ld(const S *p) { return *p; }
    __device__ static __forceinline__ void st(S *p, float v) { *p = v; }
};
template <> struct E<0> {
    typedef unsigned short S;
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float ld(const S *p) { return __builtin_bit_cast(float, (unsigned)*p << 16); }
    __device__ static __forceinline__ void st(S *p, float v) { *p = __builtin_bit_cast(unsigned short, (__bf16)v); }
};
template <> struct E<1> {
    typedef unsigned short S;
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float ld(const S *p) { return (float)__builtin_bit_cast(_Float16, *p); }
    __device__ static __forceinline__ void st(S *p, float v) { *p = __builtin_bit_cast(unsigned short, (_Float16)v); }
};

// unpack one 16-byte vector into VEC floats / pack back
template <int T> __device__ __forceinline__ void unpack(const u32x4 &v, float *f) {
    if constexpr (T == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned u = v[e];  // (bit_cast straight from the vector-element lvalue reads element 0 every time)
            f[e] = __builtin_bit_cast(float, u);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned short lo = (unsigned short)(v[e] & 0xffffu), hi = (unsigned short)(v[e] >> 16);
            f[2 * e] = E<T>::ld(&lo);
            f[2 * e + 1] = E<T>::ld(&hi);
        }
    }
}
template <int T> __device__ __forceinline__ u32x4 pack(const float *f) {
    u32x4 v;
    if constexpr (T == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_bit_cast(unsigned, f[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned short lo, hi;
            E<T>::st(&lo, f[2 * e]);
            E<T>::st(&hi, f[2 * e + 1]);
            v[e] = (unsigned)lo | ((unsigned)hi << 16);
        }
    }
    return v;
}

__device__ __forceinline__ double block_sum_f64(double v, double *sh /*[4]*/) {
    v = sgam_wave_sum_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---------------- small maps: one workgroup per (batch, group) ----------------
template <int T>
__global__ __launch_bounds__(256) void gn_small_kernel(const typename E<T>::S *__restrict__ x, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, typename E<T>::S *__restrict__ y,
                                                       int HW, int C, int groups, float eps, int swish) {
    typedef typename E<T>::S S;
    constexpr int VEC = E<T>::VEC;
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = C / groups;
    const int nv = cpg / VEC > 0 ? cpg / VEC : 1;          // 16-byte vectors per pixel of this group
    const int vlen = cpg < VEC ? cpg : VEC;                // valid elements per vector (cpg = 4 with 16-bit: half a vector)
    const S *xb = x + (int64_t)b * HW * C + g * cpg;
    S *yb = y + (int64_t)b * HW * C + g * cpg;
    const int items = HW * nv;
    float s = 0.f, ss = 0.f;
    // full 16-byte vectors and at most KEEP of them per lane (every fp32 map this kernel is used for): the group's
    // slice is read ONCE and stays in registers between the statistics and the normalisation
    constexpr int KEEP = 16;
    const bool keep = vlen == VEC && items <= KEEP * 256;
    u32x4 held[KEEP];
    if (keep) {
#pragma unroll
        for (int j = 0; j < KEEP; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < items) {
                const int pix = i / nv, k = i - pix * nv;
                held[j] = *reinterpret_cast<const u32x4 *>(xb + (int64_t)pix * C + k * VEC);
                float f[VEC];
                unpack<T>(held[j], f);
#pragma unroll
                for (int e = 0; e < VEC; ++e) { s += f[e]; ss += f[e] * f[e]; }
            }
        }
    } else {
        for (int i = threadIdx.x; i < items; i += 256) {
            const int pix = i / nv, k = i - pix * nv;
            const S *p = xb + (int64_t)pix * C + k * VEC;
            if (vlen == VEC) {
                float f[VEC];
                unpack<T>(*reinterpret_cast<const u32x4 *>(p), f);
#pragma unroll
                for (int e = 0; e < VEC; ++e) { s += f[e]; ss += f[e] * f[e]; }
            } else {
                for (int e = 0; e < vlen; ++e) { const float v = E<T>::ld(p + e); s += v; ss += v * v; }
            }
        }
    }
    __shared__ double sh[4];
    const double ts = block_sum_f64((double)s, sh);
    const double tss = block_sum_f64((double)ss, sh);
    const double n = (double)HW * (double)cpg;
    const double mean = ts / n;
    double var = tss / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float fmean = (float)mean, frstd = (float)(1.0 / sqrt(var + (double)eps));
    if (keep) {
#pragma unroll
        for (int j = 0; j < KEEP; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < items) {
                const int pix = i / nv, k = i - pix * nv;
                const int c0 = g * cpg + k * VEC;
                float f[VEC];
                unpack<T>(held[j], f);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float sc = frstd * gamma[c0 + e];
                    float v = f[e] * sc + (beta[c0 + e] - fmean * sc);
                    f[e] = swish ? sgam_swish(v) : v;
                }
                *reinterpret_cast<u32x4 *>(yb + (int64_t)pix * C + k * VEC) = pack<T>(f);
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < items; i += 256) {
        const int pix = i / nv, k = i - pix * nv;
        const S *p = xb + (int64_t)pix * C + k * VEC;
        S *q = yb + (int64_t)pix * C + k * VEC;
        const int c0 = g * cpg + k * VEC;
        if (vlen == VEC) {
            float f[VEC];
            unpack<T>(*reinterpret_cast<const u32x4 *>(p), f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float sc = frstd * gamma[c0 + e];
                float v = f[e] * sc + (beta[c0 + e] - fmean * sc);
                f[e] = swish ? sgam_swish(v) : v;
            }
            *reinterpret_cast<u32x4 *>(q) = pack<T>(f);
        } else {
            for (int e = 0; e < vlen; ++e) {
                const float sc = frstd * gamma[c0 + e];
                float v = E<T>::ld(p + e) * sc + (beta[c0 + e] - fmean * sc);
                E<T>::st(q + e, swish ? sgam_swish(v) : v);
            }
        }
    }
}

// ---------------- large maps: partial sums, then apply (with the fold of the partials inside) ----------------
constexpr int GT = 256;

template <int T>
__global__ __launch_bounds__(GT) void gn_partial_kernel(const typename E<T>::S *__restrict__ x, double *__restrict__ partial,
                                                        int HW, int C, int groups, int pix_per_chunk) {
    constexpr int VEC = E<T>::VEC;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int cv = C / VEC;            // 16-byte columns per pixel
    const int rows = GT / cv;          // pixels per pass
    const int col = threadIdx.x % cv, row = threadIdx.x / cv;
    const int p0 = chunk * pix_per_chunk, p1 = min(HW, p0 + pix_per_chunk);
    const u32x4 *xb = reinterpret_cast<const u32x4 *>(x + (int64_t)b * HW * C);
    // two 4-element halves per 16-byte vector when VEC = 8 (a 4-channel group never straddles a half)
    float s0 = 0.f, ss0 = 0.f, s1 = 0.f, ss1 = 0.f;
    for (int pix = p0 + row; pix < p1; pix += rows) {
        float f[VEC];
        unpack<T>(xb[(int64_t)pix * cv + col], f);
        s0 += (f[0] + f[1]) + (f[2] + f[3]);
        ss0 += (f[0] * f[0] + f[1] * f[1]) + (f[2] * f[2] + f[3] * f[3]);
        if constexpr (VEC == 8) {
            s1 += (f[4] + f[5]) + (f[6] + f[7]);
            ss1 += (f[4] * f[4] + f[5] * f[5]) + (f[6] * f[6] + f[7] * f[7]);
        }
    }
    __shared__ float sh[4][GT];
    sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = ss0; sh[2][threadIdx.x] = s1; sh[3][threadIdx.x] = ss1;
    __syncthreads();
    if ((int)threadIdx.x < groups) {
        const int g = threadIdx.x;
        const int cpg = C / groups;
        double ds = 0.0, dss = 0.0;
        for (int r = 0; r < rows; ++r)
            for (int h = g * cpg / 4; h < (g + 1) * cpg / 4; ++h) {   // 4-channel halves of this group
                const int t = r * cv + (VEC == 8 ? (h >> 1) : h);
                const int k = VEC == 8 ? (h & 1) * 2 : 0;
                ds += (double)sh[k][t];
                dss += (double)sh[k + 1][t];
            }
        double *o = partial + (((int64_t)b * nchunk + chunk) * groups + g) * 2;
        o[0] = ds;
        o[1] = dss;
    }
}

// fold partial[b][0..nchunk)[g] into mean / rstd for every group (256 lanes, fixed order) — LDS result
template <int T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const typename E<T>::S *__restrict__ x, const float *__restrict__ scale_shift,
                                                       typename E<T>::S *__restrict__ y, int HW, int C, int swish,
                                                       int blocks_per_batch) {
    constexpr int VEC = E<T>::VEC;
    const int b = blockIdx.x / blocks_per_batch, blk = blockIdx.x - b * blocks_per_batch;
    const int cv = C / VEC;
    const int64_t total = (int64_t)HW * cv;
    const u32x4 *xb = reinterpret_cast<const u32x4 *>(x + (int64_t)b * HW * C);
    u32x4 *yb = reinterpret_cast<u32x4 *>(y + (int64_t)b * HW * C);
    // this lane's column is fixed (the stride is a multiple of cv): hoist its scale / shift
    const int col = (int)(((int64_t)blk * 256 + threadIdx.x) % cv);
    float sc[VEC], sf[VEC];
    const float *tab = scale_shift + ((int64_t)b * C + col * VEC) * 2;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        sc[e] = tab[2 * e];
        sf[e] = tab[2 * e + 1];
    }
    for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < total; i += (int64_t)blocks_per_batch * 256) {
        float f[VEC];
        unpack<T>(xb[i], f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float v = f[e] * sc[e] + sf[e];
            f[e] = swish ? sgam_swish(v) : v;
        }
        yb[i] = pack<T>(f);
    }
}

// lane-strided sum of the chunk partials {sum, sumsq}: four loads are in flight before the first add (a plain loop is
// compiled to load -> wait -> add, one memory round trip per chunk), the adds keep the order k, k + 256, k + 512, ...
__device__ __forceinline__ void gn_fold_chunks(const double *base, int groups, int nchunk, double &s, double &ss) {
    const int64_t st = (int64_t)groups * 2;
    int k = threadIdx.x;
    for (; k + 768 < nchunk; k += 1024) {
        const double2 a = *reinterpret_cast<const double2 *>(base + k * st);
        const double2 b = *reinterpret_cast<const double2 *>(base + (k + 256) * st);
        const double2 c = *reinterpret_cast<const double2 *>(base + (k + 512) * st);
        const double2 d = *reinterpret_cast<const double2 *>(base + (k + 768) * st);
        s += a.x; ss += a.y;
        s += b.x; ss += b.y;
        s += c.x; ss += c.y;
        s += d.x; ss += d.y;
    }
    for (; k < nchunk; k += 256) {
        const double2 v = *reinterpret_cast<const double2 *>(base + k * st);
        s += v.x;
        ss += v.y;
    }
}

// finalize: one workgroup per (group, image) folds the chunk partials (lane-strided, then a fixed shuffle / LDS tree:
// deterministic) and writes the per-channel scale / shift of its channels
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double *__restrict__ partial, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *__restrict__ scale_shift,
                                                          int HW, int C, int groups, int nchunk, float eps) {
    __shared__ double sh_s[4], sh_ss[4];
    const int g = blockIdx.x, b = blockIdx.y;
    const double *base = partial + ((int64_t)b * nchunk * groups + g) * 2;
    double s = 0.0, ss = 0.0;
    gn_fold_chunks(base, groups, nchunk, s, ss);
    s = sgam_wave_sum_f64(s);
    ss = sgam_wave_sum_f64(ss);
    if ((threadIdx.x & 63) == 0) {
        sh_s[threadIdx.x >> 6] = s;
        sh_ss[threadIdx.x >> 6] = ss;
    }
    __syncthreads();
    const int cpg = C / groups;
    if ((int)threadIdx.x < cpg) {
        const double ts = (sh_s[0] + sh_s[1]) + (sh_s[2] + sh_s[3]);
        const double tss = (sh_ss[0] + sh_ss[1]) + (sh_ss[2] + sh_ss[3]);
        const double n = (double)HW * (double)cpg;
        const double mean = ts / n;
        double var = tss / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const int c = g * cpg + threadIdx.x;
        const float sc = rstd * gamma[c];
        scale_shift[((int64_t)b * C + c) * 2 + 0] = sc;
        scale_shift[((int64_t)b * C + c) * 2 + 1] = beta[c] - (float)mean * sc;
    }
}

// the same fold, delivering {mean, rstd} per (image, group) for consumers that normalise on the fly
__global__ __launch_bounds__(256) void gn_finalize_stats_kernel(const double *__restrict__ partial, float *__restrict__ mean_rstd,
                                                                int HW, int C, int groups, int nchunk, float eps) {
    __shared__ double sh_s[4], sh_ss[4];
    const int g = blockIdx.x, b = blockIdx.y;
    const double *base = partial + ((int64_t)b * nchunk * groups + g) * 2;
    double s = 0.0, ss = 0.0;
    gn_fold_chunks(base, groups, nchunk, s, ss);
    s = sgam_wave_sum_f64(s);
    ss = sgam_wave_sum_f64(ss);
    if ((threadIdx.x & 63) == 0) {
        sh_s[threadIdx.x >> 6] = s;
        sh_ss[threadIdx.x >> 6] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ts = (sh_s[0] + sh_s[1]) + (sh_s[2] + sh_s[3]);
        const double tss = (sh_ss[0] + sh_ss[1]) + (sh_ss[2] + sh_ss[3]);
        const double n = (double)HW * (double)(C / groups);
        const double mean = ts / n;
        double var = tss / n - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_rstd[(b * groups + g) * 2] = (float)mean;
        mean_rstd[(b * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__global__ void gn_finalize_kernel(const double *, const float *, const float *, float *, int, int, int, int, float);

int gn_nchunk(int HW, int C, int vec) {
    const int rows = GT / (C / vec);
    int n = HW / (rows * 16);
    if (n > 256) n = 256;
    if (n < 1) n = 1;
    return n;
}

bool gn_shape_ok(int B, int HW, int C, int groups) {
    return B > 0 && HW > 0 && C > 0 && C % 128 == 0 && C <= 1024 && groups == 32;
}

constexpr int GN_SMALL_HW = 1024;   // 16^2 and 32^2 levels: 32 workgroups of <= 64 KB each, one launch

template <int T>
int gn_launch(const void *x, const float *gamma, const float *beta, void *y, int B, int HW, int C, int groups, float eps,
              int swish, void *workspace, hipStream_t s) {
    typedef typename E<T>::S S;
    if (HW <= GN_SMALL_HW) {
        SGAM_KLAUNCH(gn_small_kernel<T>, dim3(groups, B), dim3(256), 0, s, (const S *)x, gamma, beta, (S *)y, HW, C,
                           groups, eps, swish);
        SGAM_LAUNCH_CHECK();
        return SGAM_OK;
    }
    const int nchunk = gn_nchunk(HW, C, E<T>::VEC);
    const int ppc = sgam_cdiv(HW, nchunk);
    double *partial = (double *)workspace;
    SGAM_KLAUNCH(gn_partial_kernel<T>, dim3(nchunk, B), dim3(GT), 0, s, (const S *)x, partial, HW, C, groups, ppc);
    SGAM_LAUNCH_CHECK();
    float *table = (float *)((char *)workspace + (int64_t)B * 256 * 64 * 2 * sizeof(double));
    SGAM_KLAUNCH(gn_finalize_kernel, dim3(groups, B), dim3(256), 0, s, partial, gamma, beta, table, HW, C, groups, nchunk, eps);
    SGAM_LAUNCH_CHECK();
    const int cv = C / E<T>::VEC;
    int bpb = sgam_cdiv((int64_t)HW * cv, 256 * 4);   // ~4 vectors per lane
    if (bpb > 4096) bpb = 4096;
    if (bpb < 1) bpb = 1;
    SGAM_KLAUNCH(gn_apply_kernel<T>, dim3(bpb * B), dim3(256), 0, s, (const S *)x, table, (S *)y, HW, C, swish, bpb);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

}  // namespace

extern "C" int64_t sgam_groupnorm_workspace_bytes(int32_t B, int32_t HW, int32_t C) {
    if (!gn_shape_ok(B, HW, C, 32)) return -1;
    // partial sums (double2 per (b, chunk <= 256, group <= 64)) + scale/shift table (float2 per (b, c))
    return (int64_t)B * 256 * 64 * 2 * (int64_t)sizeof(double) + (int64_t)B * C * 2 * (int64_t)sizeof(float);
}

extern "C" int64_t sgam_groupnorm_h16_workspace_bytes(int32_t B, int32_t HW, int32_t C) {
    return sgam_groupnorm_workspace_bytes(B, HW, C);
}

extern "C" int sgam_groupnorm_nhwc_f32(const float *x, const float *gamma, const float *beta, float *y, int32_t B,
                                       int32_t HW, int32_t C, int32_t groups, float eps, int32_t fuse_swish,
                                       void *workspace, int64_t workspace_bytes, void *stream) {
    if (!x || !y || !gamma || !beta || !gn_shape_ok(B, HW, C, groups)) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(y) || !sgam_aligned16(workspace)) return SGAM_EALIGN;
    if (!workspace || workspace_bytes < sgam_groupnorm_workspace_bytes(B, HW, C)) return SGAM_EWORKSPACE;
    return gn_launch<2>(x, gamma, beta, y, B, HW, C, groups, eps, fuse_swish, workspace, sgam_stream(stream));
}

extern "C" int sgam_groupnorm_nhwc_h16(const void *x, const float *gamma, const float *beta, void *y, int32_t ht, int32_t B,
                                       int32_t HW, int32_t C, int32_t groups, float eps, int32_t fuse_swish, void *workspace,
                                       int64_t workspace_bytes, void *stream) {
    if (!x || !y || !gamma || !beta || !gn_shape_ok(B, HW, C, groups)) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(y) || !sgam_aligned16(workspace)) return SGAM_EALIGN;
    if (!workspace || workspace_bytes < sgam_groupnorm_workspace_bytes(B, HW, C)) return SGAM_EWORKSPACE;
    if (ht == 0) return gn_launch<0>(x, gamma, beta, y, B, HW, C, groups, eps, fuse_swish, workspace, sgam_stream(stream));
    if (ht == 1) return gn_launch<1>(x, gamma, beta, y, B, HW, C, groups, eps, fuse_swish, workspace, sgam_stream(stream));
    return SGAM_EINVAL;
}

// GroupNorm whose per-chunk statistics were already produced by the epilogue of the convolution that wrote x
// (sgam_conv2d_stats_nhwc_f32x): finalize + apply only — the statistics pass over x disappears.
extern "C" int sgam_groupnorm_from_partials_f32(const float *x, const double *partial, int32_t nchunk, const float *gamma,
                                                const float *beta, float *y, int32_t B, int32_t HW, int32_t C,
                                                int32_t groups, float eps, int32_t fuse_swish, void *workspace,
                                                int64_t workspace_bytes, void *stream) {
    if (!x || !y || !partial || nchunk <= 0 || !gamma || !beta || !gn_shape_ok(B, HW, C, groups)) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(y) || !sgam_aligned16(workspace) || !sgam_aligned16(partial)) return SGAM_EALIGN;
    if (!workspace || workspace_bytes < (int64_t)B * C * 2 * (int64_t)sizeof(float)) return SGAM_EWORKSPACE;
    hipStream_t s = sgam_stream(stream);
    float *table = (float *)workspace;
    SGAM_KLAUNCH(gn_finalize_kernel, dim3(groups, B), dim3(256), 0, s, partial, gamma, beta, table, HW, C, groups, nchunk, eps);
    SGAM_LAUNCH_CHECK();
    const int cv = C / 4;
    int bpb = sgam_cdiv((int64_t)HW * cv, 256 * 4);
    if (bpb > 4096) bpb = 4096;
    if (bpb < 1) bpb = 1;
    SGAM_KLAUNCH(gn_apply_kernel<2>, dim3(bpb * B), dim3(256), 0, s, x, table, y, HW, C, fuse_swish, bpb);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// {mean, rstd} per (image, group) straight from conv-epilogue partials (no pass over the tensor at all): the consumer is
// a convolution that normalises while staging its input (sgam_conv2d_gn_nhwc_f32x)
extern "C" int sgam_groupnorm_stats_from_partials_f32(const double *partial, int32_t nchunk, float *mean_rstd, int32_t B,
                                                      int32_t HW, int32_t C, int32_t groups, float eps, void *stream) {
    if (!partial || nchunk <= 0 || !mean_rstd || !gn_shape_ok(B, HW, C, groups)) return SGAM_EINVAL;
    if (!sgam_aligned16(partial)) return SGAM_EALIGN;
    SGAM_KLAUNCH(gn_finalize_stats_kernel, dim3(groups, B), dim3(256), 0, sgam_stream(stream), partial, mean_rstd, HW, C,
                       groups, nchunk, eps);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// {mean, rstd} of any NHWC fp32 tensor: the partial-sum pass + the fold
extern "C" int sgam_groupnorm_meanrstd_nhwc_f32(const float *x, float *mean_rstd, int32_t B, int32_t HW, int32_t C,
                                                int32_t groups, float eps, void *workspace, int64_t workspace_bytes,
                                                void *stream) {
    if (!x || !mean_rstd || !gn_shape_ok(B, HW, C, groups)) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(workspace)) return SGAM_EALIGN;
    if (!workspace || workspace_bytes < sgam_groupnorm_workspace_bytes(B, HW, C)) return SGAM_EWORKSPACE;
    hipStream_t s = sgam_stream(stream);
    const int nchunk = gn_nchunk(HW, C, 4);
    double *partial = (double *)workspace;
    SGAM_KLAUNCH(gn_partial_kernel<2>, dim3(nchunk, B), dim3(GT), 0, s, x, partial, HW, C, groups, sgam_cdiv(HW, nchunk));
    SGAM_LAUNCH_CHECK();
    SGAM_KLAUNCH(gn_finalize_stats_kernel, dim3(groups, B), dim3(256), 0, s, partial, mean_rstd, HW, C, groups, nchunk, eps);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// 16-bit tensors (bf16 ht = 0 / fp16 ht = 1) whose producer — the 16-bit halo conv (h16_halo.hip) — left partial statistics:
// finalize + apply only, and {mean, rstd} for consumers that normalise while staging
extern "C" int sgam_groupnorm_from_partials_h16(const void *x, const double *partial, int32_t nchunk, const float *gamma,
                                                const float *beta, void *y, int32_t ht, int32_t B, int32_t HW, int32_t C,
                                                int32_t groups, float eps, int32_t fuse_swish, void *workspace,
                                                int64_t workspace_bytes, void *stream) {
    if (!x || !y || !partial || nchunk <= 0 || !gamma || !beta || !gn_shape_ok(B, HW, C, groups) || (ht != 0 && ht != 1))
        return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(y) || !sgam_aligned16(workspace) || !sgam_aligned16(partial)) return SGAM_EALIGN;
    if (!workspace || workspace_bytes < (int64_t)B * C * 2 * (int64_t)sizeof(float)) return SGAM_EWORKSPACE;
    hipStream_t s = sgam_stream(stream);
    float *table = (float *)workspace;
    SGAM_KLAUNCH(gn_finalize_kernel, dim3(groups, B), dim3(256), 0, s, partial, gamma, beta, table, HW, C, groups, nchunk, eps);
    SGAM_LAUNCH_CHECK();
    const int cv = C / 8;
    int bpb = sgam_cdiv((int64_t)HW * cv, 256 * 4);
    if (bpb > 4096) bpb = 4096;
    if (bpb < 1) bpb = 1;
    if (ht == 0)
        SGAM_KLAUNCH(gn_apply_kernel<0>, dim3(bpb * B), dim3(256), 0, s, (const unsigned short *)x, table, (unsigned short *)y, HW, C,
                     fuse_swish, bpb);
    else
        SGAM_KLAUNCH(gn_apply_kernel<1>, dim3(bpb * B), dim3(256), 0, s, (const unsigned short *)x, table, (unsigned short *)y, HW, C,
                     fuse_swish, bpb);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// the per-(image, channel) {scale, shift} table alone, from the producer's chunk records:
// the finalize half of sgam_groupnorm_from_partials_*, for consumers that apply y = x scale + shift themselves while they stage x
// (the fused AttnBlock front end of the 16-bit mode, attention.hip: sgam_attn_block_h16)
extern "C" int sgam_groupnorm_table_from_partials(const double *partial, int32_t nchunk, const float *gamma, const float *beta,
                                                  float *scale_shift, int32_t B, int32_t HW, int32_t C, int32_t groups, float eps,
                                                  void *stream) {
    if (!partial || nchunk <= 0 || !gamma || !beta || !scale_shift || !gn_shape_ok(B, HW, C, groups))
        return SGAM_EINVAL;
    if (!sgam_aligned16(partial) || !sgam_aligned16(scale_shift)) return SGAM_EALIGN;
    hipStream_t s = sgam_stream(stream);
    SGAM_KLAUNCH(gn_finalize_kernel, dim3(groups, B), dim3(256), 0, s, partial, gamma, beta, scale_shift, HW, C, groups, nchunk, eps);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_groupnorm_meanrstd_nhwc_h16(const void *x, float *mean_rstd, int32_t ht, int32_t B, int32_t HW, int32_t C,
                                                int32_t groups, float eps, void *workspace, int64_t workspace_bytes,
                                                void *stream) {
    if (!x || !mean_rstd || !gn_shape_ok(B, HW, C, groups) || (ht != 0 && ht != 1)) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(workspace)) return SGAM_EALIGN;
    if (!workspace || workspace_bytes < sgam_groupnorm_workspace_bytes(B, HW, C)) return SGAM_EWORKSPACE;
    hipStream_t s = sgam_stream(stream);
    const int nchunk = gn_nchunk(HW, C, 8);
    double *partial = (double *)workspace;
    if (ht == 0)
        SGAM_KLAUNCH(gn_partial_kernel<0>, dim3(nchunk, B), dim3(GT), 0, s, (const unsigned short *)x, partial, HW, C, groups,
                     sgam_cdiv(HW, nchunk));
    else
        SGAM_KLAUNCH(gn_partial_kernel<1>, dim3(nchunk, B), dim3(GT), 0, s, (const unsigned short *)x, partial, HW, C, groups,
                     sgam_cdiv(HW, nchunk));
    SGAM_LAUNCH_CHECK();
    SGAM_KLAUNCH(gn_finalize_stats_kernel, dim3(groups, B), dim3(256), 0, s, partial, mean_rstd, HW, C, groups, nchunk, eps);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_groupnorm_stats_nhwc_f32(const float *x, const float *gamma, const float *beta,
                                             float *scale_shift, int32_t B, int32_t HW, int32_t C, int32_t groups,
                                             float eps, void *workspace, int64_t workspace_bytes, void *stream) {
    if (!x || !scale_shift || !gamma || !beta || !gn_shape_ok(B, HW, C, groups)) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(scale_shift) || !sgam_aligned16(workspace)) return SGAM_EALIGN;
    if (!workspace || workspace_bytes < sgam_groupnorm_workspace_bytes(B, HW, C)) return SGAM_EWORKSPACE;
    hipStream_t s = sgam_stream(stream);
    const int nchunk = gn_nchunk(HW, C, 4);
    double *partial = (double *)workspace;
    SGAM_KLAUNCH(gn_partial_kernel<2>, dim3(nchunk, B), dim3(GT), 0, s, x, partial, HW, C, groups, sgam_cdiv(HW, nchunk));
    SGAM_LAUNCH_CHECK();
    SGAM_KLAUNCH(gn_finalize_kernel, dim3(groups, B), dim3(256), 0, s, partial, gamma, beta, scale_shift, HW, C, groups, nchunk, eps);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}
