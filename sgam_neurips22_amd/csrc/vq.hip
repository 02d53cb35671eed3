// vq.hip — nearest-codeword vector quantiser (VectorQuantizer2, modules/vqvae/quantize.py:275-381).
//
// The z.e^T contraction ([T][D] x [n_e][D]^T) runs on the matrix cores through the same implicit-GEMM
// kernel as the convolutions (a 1x1 "conv" over T pixels with the codebook as the weight matrix);
// the distance expression, the arg-min (first index among exact ties, like torch.argmin on CPU) and
// the embedding gather are exact-semantics kernels: explicit round-to-nearest intrinsics, no
// contraction, so that   d = (|z|^2 + |e|^2) - 2*(z.e)   is evaluated in the reference's order.
#include "sgam_common.h"

namespace {

__device__ __forceinline__ void argmin_combine(float &d, int &j, float od, int oj) {
    if (od < d || (od == d && oj < j)) {
        d = od;
        j = oj;
    }
}

// one workgroup per token row
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float *__restrict__ z, const float *__restrict__ codebook,
                                                        const float *__restrict__ e_sq, const float *__restrict__ dots,
                                                        int64_t *__restrict__ idx_out, float *__restrict__ zq_out,
                                                        float *__restrict__ dist_out, int D, int n_e,
                                                        int straight_through) {
    const int t = blockIdx.x;
    const float *zr = z + (int64_t)t * D;
    __shared__ float red_f[4];
    __shared__ int red_i[4];
    __shared__ float bc_f;
    __shared__ int bc_i;

    // |z|^2
    float zz = 0.f;
    for (int c = threadIdx.x; c < D; c += 256) zz = __fmaf_rn(zr[c], zr[c], zz);
    zz = sgam_wave_sum(zz);
    if ((threadIdx.x & 63) == 0) red_f[threadIdx.x >> 6] = zz;
    __syncthreads();
    if (threadIdx.x == 0) bc_f = (red_f[0] + red_f[1]) + (red_f[2] + red_f[3]);
    __syncthreads();
    zz = bc_f;
    __syncthreads();

    const float *dr = dots + (int64_t)t * n_e;
    float best = INFINITY;
    int bj = 0x7fffffff;
    for (int j = threadIdx.x; j < n_e; j += 256) {
        const float d = __fsub_rn(__fadd_rn(zz, e_sq[j]), __fmul_rn(2.0f, dr[j]));
        if (dist_out) dist_out[(int64_t)t * n_e + j] = d;
        if (d < best) {  // strict: the lowest j among equal distances survives within a lane
            best = d;
            bj = j;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float od = __shfl_xor(best, o, 64);
        const int oj = __shfl_xor(bj, o, 64);
        argmin_combine(best, bj, od, oj);
    }
    if ((threadIdx.x & 63) == 0) {
        red_f[threadIdx.x >> 6] = best;
        red_i[threadIdx.x >> 6] = bj;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float d = red_f[0];
        int j = red_i[0];
        for (int w = 1; w < 4; ++w) argmin_combine(d, j, red_f[w], red_i[w]);
        if (j == 0x7fffffff) j = 0;  // all-NaN row
        bc_i = j;
        idx_out[t] = (int64_t)j;
    }
    __syncthreads();
    if (zq_out) {
        const float *er = codebook + (int64_t)bc_i * D;
        for (int c = threadIdx.x; c < D; c += 256) {
            const float e = er[c];
            // quantize.py:304  z + (z_q - z).detach(): NOT a pure copy in fp32
            zq_out[(int64_t)t * D + c] = straight_through ? __fadd_rn(zr[c], __fsub_rn(e, zr[c])) : e;
        }
    }
}

__global__ void vq_gather_kernel(const float *__restrict__ codebook, const int64_t *__restrict__ idx,
                                 float *__restrict__ out, int T, int D, int n_e) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d4 = D >> 2;
    if (i >= (int64_t)T * d4) return;
    const int t = (int)(i / d4), c = (int)(i % d4);
    int64_t j = idx[t];
    if (j < 0) j = 0;
    if (j >= n_e) j = n_e - 1;
    reinterpret_cast<f32x4 *>(out)[i] = reinterpret_cast<const f32x4 *>(codebook + j * D)[c];
}

__global__ __launch_bounds__(256) void row_sumsq_kernel(const float *__restrict__ x, float *__restrict__ out, int rows,
                                                        int cols) {
    // one wavefront per row
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const float v = x[(int64_t)row * cols + c];
        s = __fmaf_rn(v, v, s);
    }
    s = sgam_wave_sum(s);
    if (lane == 0) out[row] = s;
}

// k rounds of "smallest (d, j) strictly greater than the previous pick" — ascending values,
// ties by lower index.  One workgroup per token row.
__global__ __launch_bounds__(256) void vq_topk_kernel(const float *__restrict__ dist, float *__restrict__ vals,
                                                      int64_t *__restrict__ inds, int n_e, int k) {
    const int t = blockIdx.x;
    const float *dr = dist + (int64_t)t * n_e;
    __shared__ float red_f[4];
    __shared__ int red_i[4];
    __shared__ float bc_f;
    __shared__ int bc_i;
    float last_d = -INFINITY;
    int last_j = -1;
    for (int r = 0; r < k; ++r) {
        float best = INFINITY;
        int bj = 0x7fffffff;
        for (int j = threadIdx.x; j < n_e; j += 256) {
            const float d = dr[j];
            const bool after = (d > last_d) || (d == last_d && j > last_j);
            if (after && d < best) {
                best = d;
                bj = j;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float od = __shfl_xor(best, o, 64);
            const int oj = __shfl_xor(bj, o, 64);
            argmin_combine(best, bj, od, oj);
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            red_f[threadIdx.x >> 6] = best;
            red_i[threadIdx.x >> 6] = bj;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float d = red_f[0];
            int j = red_i[0];
            for (int w = 1; w < 4; ++w) argmin_combine(d, j, red_f[w], red_i[w]);
            bc_f = d;
            bc_i = j;
            vals[(int64_t)t * k + r] = d;
            inds[(int64_t)t * k + r] = (j == 0x7fffffff) ? 0 : (int64_t)j;
        }
        __syncthreads();
        last_d = bc_f;
        last_j = bc_i;
    }
}

// commitment loss (quantize.py:296-301, legacy=True): mean((z_q - z)^2) + beta * mean((z_q - z)^2).  One wavefront per
// token sums its D squared differences in fp64 (fixed lane order); a single workgroup then folds the T token sums in a
// fixed order, so the scalar is run-to-run reproducible.
__global__ __launch_bounds__(256) void vq_commit_partial_kernel(const float *__restrict__ z, const float *__restrict__ codebook,
                                                                const int64_t *__restrict__ idx, double *__restrict__ partial,
                                                                int T, int D, int n_e) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const int lane = threadIdx.x & 63;
    int64_t j = idx[t];
    j = j < 0 ? 0 : (j >= n_e ? n_e - 1 : j);
    double s = 0.0;
    for (int c = lane; c < D; c += 64) {
        const float d = __fsub_rn(codebook[j * D + c], z[(int64_t)t * D + c]);
        s += (double)d * (double)d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) partial[t] = s;
}

__global__ __launch_bounds__(256) void vq_commit_fold_kernel(const double *__restrict__ partial, float *__restrict__ loss,
                                                             int T, int D, float beta) {
    __shared__ double red[256];
    double s = 0.0;
    for (int t = threadIdx.x; t < T; t += 256) s += partial[t];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float m = (float)(red[0] / ((double)T * (double)D));
        loss[0] = __fadd_rn(m, __fmul_rn(beta, m));
    }
}

}  // namespace

extern "C" int sgam_vq_commit_loss_f32(const float *z, const float *codebook, const int64_t *idx, double *partial,
                                       float *loss, int32_t T, int32_t D, int32_t n_e, float beta, void *stream) {
    if (!z || !codebook || !idx || !partial || !loss || T <= 0 || D <= 0 || n_e <= 0) return SGAM_EINVAL;
    SGAM_KLAUNCH(vq_commit_partial_kernel, dim3(sgam_cdiv(T, 4)), dim3(256), 0, sgam_stream(stream), z, codebook, idx,
                       partial, T, D, n_e);
    SGAM_LAUNCH_CHECK();
    SGAM_KLAUNCH(vq_commit_fold_kernel, dim3(1), dim3(256), 0, sgam_stream(stream), partial, loss, T, D, beta);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_row_sumsq_f32(const float *x, float *out, int32_t rows, int32_t cols, void *stream) {
    if (!x || !out || rows <= 0 || cols <= 0) return SGAM_EINVAL;
    SGAM_KLAUNCH(row_sumsq_kernel, dim3(sgam_cdiv(rows, 4)), dim3(256), 0, sgam_stream(stream), x, out, rows, cols);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

static sgam_conv_desc vq_dot_desc(int32_t T, int32_t D, int32_t n_e) {
    // dots[t][j] = z[t] . e[j] on the MFMA path: a 1x1 "conv" over T pixels, weight matrix = codebook.
    sgam_conv_desc d = {};
    d.B = 1; d.Hi = 1; d.Wi = T; d.Cin = D; d.Ho = 1; d.Wo = T; d.N = n_e;
    d.KH = 1; d.KW = 1; d.stride = 1; d.pad_t = 0; d.pad_l = 0; d.upsample2x = 0;
    d.lda = D; d.ldb = D; d.ldc = n_e; d.ldr = 0; d.n_valid = n_e; d.bias_per_row = 0;
    return d;
}

extern "C" int64_t sgam_vq_workspace_bytes(int32_t T, int32_t D, int32_t n_e) {
    if (T <= 0 || D <= 0 || D % 32 != 0 || n_e <= 0 || n_e % 64 != 0) return -1;
    const sgam_conv_desc d = vq_dot_desc(T, D, n_e);
    return sgam_conv2d_workspace_bytes(&d);
}

extern "C" int sgam_vq_nearest_f32(const float *z, const float *codebook, const float *e_sq, float *dots,
                                   int64_t *idx_out, float *zq_out, float *dist_out, int32_t T, int32_t D,
                                   int32_t n_e, int32_t straight_through, void *workspace,
                                   int64_t workspace_bytes, void *stream) {
    if (!z || !codebook || !e_sq || !dots || !idx_out || T <= 0) return SGAM_EINVAL;
    if (D <= 0 || D % 32 != 0 || n_e <= 0 || n_e % 64 != 0) return SGAM_EINVAL;
    const sgam_conv_desc d = vq_dot_desc(T, D, n_e);
    int rc = sgam_conv2d_nhwc_f32(&d, z, codebook, nullptr, nullptr, dots, workspace, workspace_bytes, stream);
    if (rc != SGAM_OK) return rc;
    SGAM_KLAUNCH(vq_argmin_kernel, dim3(T), dim3(256), 0, sgam_stream(stream), z, codebook, e_sq, dots, idx_out,
                       zq_out, dist_out, D, n_e, straight_through);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_vq_gather_f32(const float *codebook, const int64_t *idx, float *out, int32_t T, int32_t D,
                                  int32_t n_e, void *stream) {
    if (!codebook || !idx || !out || T <= 0 || D <= 0 || D % 4 != 0 || n_e <= 0) return SGAM_EINVAL;
    const int64_t total = (int64_t)T * (D / 4);
    SGAM_KLAUNCH(vq_gather_kernel, dim3(sgam_cdiv(total, 256)), dim3(256), 0, sgam_stream(stream), codebook, idx,
                       out, T, D, n_e);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_vq_topk_f32(const float *dist, float *vals, int64_t *inds, int32_t T, int32_t n_e, int32_t k,
                                void *stream) {
    if (!dist || !vals || !inds || T <= 0 || n_e <= 0 || k <= 0 || k > 64 || k > n_e) return SGAM_EINVAL;
    SGAM_KLAUNCH(vq_topk_kernel, dim3(T), dim3(256), 0, sgam_stream(stream), dist, vals, inds, n_e, k);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}
