// norm_softmax.hip — the reduction kernels of the VQGAN: GroupNorm(+swish) on NHWC maps and the
// row softmax of the attention scores.  Both are HBM-bound (one read + one write per element);
// reductions run on wavefront shuffles (64 lanes) and a small LDS hand-off, never on atomics, so
// results are bit-reproducible run to run.
#include "sgam_common.h"

namespace {

// --------------------------------------------------------------------------------------------
// GroupNorm, NHWC [B][HW][C], C = 32 groups * cpg, C % 128 == 0 so that the float4 a lane loads
// (4 consecutive channels) always falls inside one group.
//
// pass 1  gn_partial_kernel : grid (nchunk, B).  A workgroup of 256 lanes walks a contiguous chunk
//         of pixels; lane t owns float4 column t % (C/4) of every (256/(C/4))-th pixel.  Per-lane fp32
//         sum / sum-of-squares (<= a few hundred terms), then 8 lanes per group are combined in fp64
//         in a fixed order -> partial[b][chunk][g] = {sum, sumsq} (double2).
// pass 2  gn_finalize_kernel: grid (B).  Fixed-order fp64 reduction over chunks -> per-(b, channel)
//         scale = rstd*gamma, shift = beta - mean*rstd*gamma  (biased variance, eps inside sqrt).
// pass 3  gn_apply_kernel   : y = x*scale + shift, optional swish y*sigmoid(y).
// --------------------------------------------------------------------------------------------
constexpr int GN_THREADS = 256;

__global__ __launch_bounds__(GN_THREADS) void gn_partial_kernel(const float *__restrict__ x, double *__restrict__ partial,
                                                                int HW, int C, int groups, int pix_per_chunk) {
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int c4 = C >> 2;                   // float4 columns per pixel: 32, 64 or 128
    const int rows = GN_THREADS / c4;        // pixels covered per pass: 8, 4 or 2
    const int col = threadIdx.x % c4;
    const int row = threadIdx.x / c4;
    const int p0 = chunk * pix_per_chunk;
    const int p1 = min(HW, p0 + pix_per_chunk);
    const f32x4 *xb = reinterpret_cast<const f32x4 *>(x + (int64_t)b * HW * C);
    float s = 0.f, ss = 0.f;
    for (int p = p0 + row; p < p1; p += rows) {
        const f32x4 v = xb[(int64_t)p * c4 + col];
        s += (v[0] + v[1]) + (v[2] + v[3]);
        ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    __shared__ float sh_s[GN_THREADS], sh_ss[GN_THREADS];
    sh_s[threadIdx.x] = s;
    sh_ss[threadIdx.x] = ss;
    __syncthreads();
    if (threadIdx.x < groups) {
        const int g = threadIdx.x;
        const int cpg4 = c4 / groups;  // float4 columns per group: 1, 2 or 4
        double ds = 0.0, dss = 0.0;
        for (int r = 0; r < rows; ++r)
            for (int k = 0; k < cpg4; ++k) {
                const int t = r * c4 + g * cpg4 + k;
                ds += (double)sh_s[t];
                dss += (double)sh_ss[t];
            }
        double *o = partial + (((int64_t)b * nchunk + chunk) * groups + g) * 2;
        o[0] = ds;
        o[1] = dss;
    }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const double *__restrict__ partial,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta,
                                                          float *__restrict__ scale_shift, int HW, int C, int groups,
                                                          int nchunk, float eps) {
    // 256 lanes: lane -> (group g = t % groups, part = t / groups); each lane sums chunks part, part+np, ...
    // (independent loads, unrolled), then `np` partials per group are combined in a fixed order.
    const int b = blockIdx.x;
    __shared__ double sh_s[256], sh_ss[256];
    __shared__ float sh_mean[64], sh_rstd[64];
    const int np = 256 / groups;
    const int g = threadIdx.x % groups, part = threadIdx.x / groups;
    double s = 0.0, ss = 0.0;
    if (part < np) {
        const double *base = partial + ((int64_t)b * nchunk * groups + g) * 2;
#pragma unroll 4
        for (int k = part; k < nchunk; k += np) {
            const double2 v = *reinterpret_cast<const double2 *>(base + (int64_t)k * groups * 2);
            s += v.x;
            ss += v.y;
        }
    }
    sh_s[threadIdx.x] = s;
    sh_ss[threadIdx.x] = ss;
    __syncthreads();
    if ((int)threadIdx.x < groups) {
        double ts = 0.0, tss = 0.0;
        for (int q = 0; q < np; ++q) {
            ts += sh_s[q * groups + g];
            tss += sh_ss[q * groups + g];
        }
        const double n = (double)HW * (double)(C / groups);
        const double mean = ts / n;
        double var = tss / n - mean * mean;
        if (var < 0.0) var = 0.0;
        sh_mean[g] = (float)mean;
        sh_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cpg = C / groups;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int gg = c / cpg;
        const float sc = sh_rstd[gg] * gamma[c];
        scale_shift[((int64_t)b * C + c) * 2 + 0] = sc;
        scale_shift[((int64_t)b * C + c) * 2 + 1] = beta[c] - sh_mean[gg] * sc;
    }
}

// x * sigmoid(x) with sigmoid = 1/(1+exp(-x)), the expression order of diffusionmodules/model.py:29-31
__device__ __forceinline__ float swish_f(float v) { return sgam_swish(v); }  // same function as the fused conv prologue

__global__ __launch_bounds__(256) void gn_apply_kernel(const float *__restrict__ x, const float *__restrict__ scale_shift,
                                                       float *__restrict__ y, int64_t total4, int HW, int C,
                                                       int fuse_swish) {
    const int c4 = C >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % c4);
        const int b = (int)(i / ((int64_t)HW * c4));
        const f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
        const float *ssp = scale_shift + ((int64_t)b * C + col * 4) * 2;
        const f32x4 s01 = *reinterpret_cast<const f32x4 *>(ssp);      // sc0 sh0 sc1 sh1
        const f32x4 s23 = *reinterpret_cast<const f32x4 *>(ssp + 4);  // sc2 sh2 sc3 sh3
        f32x4 o;
        o[0] = v[0] * s01[0] + s01[1];
        o[1] = v[1] * s01[2] + s01[3];
        o[2] = v[2] * s23[0] + s23[1];
        o[3] = v[3] * s23[2] + s23[3];
        if (fuse_swish) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = swish_f(o[e]);
        }
        reinterpret_cast<f32x4 *>(y)[i] = o;
    }
}

int gn_nchunk(int HW, int C) {
    const int rows = GN_THREADS / (C / 4);
    int n = HW / (rows * 16);  // >= 16 passes per workgroup
    if (n > 256) n = 256;
    if (n < 1) n = 1;
    return n;
}

// --------------------------------------------------------------------------------------------
// Row softmax in place, one workgroup (256 lanes) per row, the row kept in registers between the
// max / exp-sum / normalise passes (cols <= 256*4*MAXV).  s = softmax(scale * s).
// --------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void softmax_rows_kernel(float *__restrict__ s, int cols, int ld, float scale) {
    float *row = s + (int64_t)blockIdx.x * ld;
    const int c4 = cols >> 2;
    f32x4 v[MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = threadIdx.x + k * 256;
        if (i < c4) {
            v[k] = reinterpret_cast<const f32x4 *>(row)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[k][e] *= scale;
                mx = fmaxf(mx, v[k][e]);
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
            for (int e = 0; e < 4; ++e) {
                v[k][e] = expf(v[k][e] - mx);
                sum += v[k][e];
            }
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
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k][e] *= inv;
            reinterpret_cast<f32x4 *>(row)[i] = v[k];
        }
    }
}

}  // namespace

// shared with the 16-bit path (h16.hip): statistics are finalised in fp64 -> fp32 table for both
extern "C" int sgam_gn_finalize_launch(const double *partial, const float *gamma, const float *beta, float *scale_shift,
                                       int B, int HW, int C, int groups, int nchunk, float eps, hipStream_t s) {
    if (256 % groups != 0) return SGAM_EINVAL;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, partial, gamma, beta, scale_shift, HW, C, groups,
                       nchunk, eps);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int64_t sgam_groupnorm_workspace_bytes(int32_t B, int32_t HW, int32_t C) {
    if (B <= 0 || HW <= 0 || C <= 0 || C % 128 != 0 || C > 1024) return -1;
    const int nchunk = gn_nchunk(HW, C);
    // partial sums (double2 per (b, chunk, group<=64)) + scale/shift table (float2 per (b, c))
    return (int64_t)B * nchunk * 64 * 2 * (int64_t)sizeof(double) + (int64_t)B * C * 2 * (int64_t)sizeof(float);
}

extern "C" int sgam_groupnorm_nhwc_f32(const float *x, const float *gamma, const float *beta, float *y, int32_t B,
                                       int32_t HW, int32_t C, int32_t groups, float eps, int32_t fuse_swish,
                                       void *workspace, int64_t workspace_bytes, void *stream) {
    if (!x || !y || !gamma || !beta || B <= 0 || HW <= 0) return SGAM_EINVAL;
    if (C <= 0 || C % 128 != 0 || C > 1024 || groups <= 0 || groups > 64 || C % groups != 0) return SGAM_EINVAL;
    if ((C / groups) % 4 != 0 || (C / 4) % groups != 0 || GN_THREADS % (C / 4) != 0) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(y) || !sgam_aligned16(workspace)) return SGAM_EALIGN;
    if (!workspace || workspace_bytes < sgam_groupnorm_workspace_bytes(B, HW, C)) return SGAM_EWORKSPACE;
    hipStream_t s = sgam_stream(stream);
    const int nchunk = gn_nchunk(HW, C);
    const int pix_per_chunk = sgam_cdiv(HW, nchunk);
    double *partial = (double *)workspace;
    float *scale_shift = (float *)((char *)workspace + (int64_t)B * nchunk * 64 * 2 * sizeof(double));
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, B), dim3(GN_THREADS), 0, s, x, partial, HW, C, groups,
                       pix_per_chunk);
    SGAM_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, partial, gamma, beta, scale_shift, HW, C, groups,
                       nchunk, eps);
    SGAM_LAUNCH_CHECK();
    const int64_t total4 = (int64_t)B * HW * (C / 4);
    int blocks = sgam_cdiv(total4, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks), dim3(256), 0, s, x, scale_shift, y, total4, HW, C, fuse_swish);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_groupnorm_stats_nhwc_f32(const float *x, const float *gamma, const float *beta,
                                             float *scale_shift, int32_t B, int32_t HW, int32_t C, int32_t groups,
                                             float eps, void *workspace, int64_t workspace_bytes, void *stream) {
    if (!x || !scale_shift || !gamma || !beta || B <= 0 || HW <= 0) return SGAM_EINVAL;
    if (C <= 0 || C % 128 != 0 || C > 1024 || groups <= 0 || groups > 64 || C % groups != 0) return SGAM_EINVAL;
    if ((C / groups) % 4 != 0 || (C / 4) % groups != 0 || GN_THREADS % (C / 4) != 0 || 256 % groups != 0) return SGAM_EINVAL;
    if (!sgam_aligned16(x) || !sgam_aligned16(scale_shift) || !sgam_aligned16(workspace)) return SGAM_EALIGN;
    if (!workspace || workspace_bytes < sgam_groupnorm_workspace_bytes(B, HW, C)) return SGAM_EWORKSPACE;
    hipStream_t s = sgam_stream(stream);
    const int nchunk = gn_nchunk(HW, C);
    const int pix_per_chunk = sgam_cdiv(HW, nchunk);
    double *partial = (double *)workspace;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, B), dim3(GN_THREADS), 0, s, x, partial, HW, C, groups,
                       pix_per_chunk);
    SGAM_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, partial, gamma, beta, scale_shift, HW, C, groups,
                       nchunk, eps);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_softmax_rows_f32(float *sp, int32_t rows, int32_t cols, int32_t ld, float scale, void *stream) {
    if (!sp || rows <= 0 || cols <= 0 || cols % 4 != 0 || ld < cols || ld % 4 != 0) return SGAM_EINVAL;
    if (!sgam_aligned16(sp)) return SGAM_EALIGN;
    hipStream_t s = sgam_stream(stream);
    if (cols <= 1024) {
        hipLaunchKernelGGL((softmax_rows_kernel<1>), dim3(rows), dim3(256), 0, s, sp, cols, ld, scale);
    } else if (cols <= 4096) {
        hipLaunchKernelGGL((softmax_rows_kernel<4>), dim3(rows), dim3(256), 0, s, sp, cols, ld, scale);
    } else if (cols <= 16384) {
        hipLaunchKernelGGL((softmax_rows_kernel<16>), dim3(rows), dim3(256), 0, s, sp, cols, ld, scale);
    } else {
        return SGAM_EINVAL;
    }
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}
