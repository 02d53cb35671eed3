// train.hip — backward / optimiser kernels of the reference's training step: VQModel.training_step,
// sgam/generative_sensing_module/model.py:271-345, with VQLPIPSWithDiscriminator (modules/losses/vqperceptual.py:34-137),
// the PatchGAN discriminator (modules/discriminator/model.py) and LPIPS (modules/losses/lpips.py).  gfx950 only.
//
// Every product of the backward pass is a GEMM on the existing MFMA kernels (fp32-in MFMA mode: gradients are far below
// fp16's normal range, so the split-fp32 trick of the inference path does not apply):
//   data gradient    dcol[M][taps*Cin] = dy[M][Cout] . W[Cout][taps*Cin]          then col2im_gather (below)
//   weight gradient  dW[Cout][taps*Cin] = dy^T[Cout][M] . col^T[taps*Cin][M]^T     with col^T from im2col_t (below)
// which makes one pair of index kernels serve every convolution: 3x3 / 1x1 / 4x4, stride 1 / 2, the (0,1,0,1) padding of
// Downsample (model.py:62-75) and the nearest-2x upsampling folded into Upsample.conv (:43-53).  This file holds those index
// kernels and the element-wise / reduction kernels around the GEMMs: GroupNorm(+swish) and BatchNorm(+LeakyReLU) backward,
// soft-max backward, max-pool, the LPIPS feature-level kernel, the L1 reconstruction loss and the hinge terms with their
// gradients, the quantiser's straight-through + commitment gradient, bias gradients, sums of squares, Adam.  Not on the
// inference hot path; written for correctness, determinism (no atomics) and coalesced access, not tuned.
#include "sgam_common.h"

namespace {

struct ConvGeo {
    int B, Hi, Wi, Cin, Cin_pad, Ho, Wo, KH, KW, stride, pad_t, pad_l, ups;
};

// source pixel of (output pixel, tap) in the conv's input; false = zero padding.  `ups`: the conv runs on the nearest-2x
// upsampling of the stored tensor (virtual size 2 Hi x 2 Wi).
__device__ __forceinline__ bool src_of(const ConvGeo &g, int oy, int ox, int ky, int kx, int &iy, int &ix) {
    const int vy = oy * g.stride + ky - g.pad_t, vx = ox * g.stride + kx - g.pad_l;
    const int Hv = g.ups ? 2 * g.Hi : g.Hi, Wv = g.ups ? 2 * g.Wi : g.Wi;
    if ((unsigned)vy >= (unsigned)Hv || (unsigned)vx >= (unsigned)Wv) return false;
    iy = g.ups ? vy >> 1 : vy;
    ix = g.ups ? vx >> 1 : vx;
    return true;
}

// col^T[k = tap * Cin_pad + c][m = (b, oy, ox)] of the NHWC input x (row stride ld); rows of col^T are ldm apart (>= M: the
// caller zero-fills the tail when it rounds the GEMM's K = M up)
__global__ __launch_bounds__(256) void im2col_t_kernel(const float *__restrict__ x, int ld, float *__restrict__ out, int64_t ldm, ConvGeo g) {
    const int64_t M = (int64_t)g.B * g.Ho * g.Wo;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)g.KH * g.KW * g.Cin_pad * M;
    if (i >= total) return;
    const int64_t m = i % M;
    const int k = (int)(i / M);
    const int c = k % g.Cin_pad, tap = k / g.Cin_pad;
    const int ky = tap / g.KW, kx = tap - ky * g.KW;
    const int ox = (int)(m % g.Wo), oy = (int)((m / g.Wo) % g.Ho), b = (int)(m / ((int64_t)g.Wo * g.Ho));
    int iy, ix;
    float v = 0.f;
    if (c < g.Cin && src_of(g, oy, ox, ky, kx, iy, ix)) v = x[((int64_t)(b * g.Hi + iy) * g.Wi + ix) * ld + c];
    out[(int64_t)k * ldm + m] = v;
}

// dx[b][iy][ix][c] = sum over (output pixel, tap) pairs that read this input pixel of dcol[m][tap * Cin_pad + c]; a gather
// in a fixed order (taps ascending, then the 2 x 2 virtual pixels of an upsampled source), so it is deterministic
__global__ __launch_bounds__(256) void col2im_gather_kernel(const float *__restrict__ dcol, float *__restrict__ dx, int ldx, ConvGeo g) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)g.B * g.Hi * g.Wi * g.Cin;
    if (i >= total) return;
    const int c = (int)(i % g.Cin);
    const int64_t p = i / g.Cin;
    const int ix = (int)(p % g.Wi), iy = (int)((p / g.Wi) % g.Hi), b = (int)(p / ((int64_t)g.Wi * g.Hi));
    const int K = g.KH * g.KW * g.Cin_pad;
    const int nv = g.ups ? 2 : 1;
    float acc = 0.f;
    for (int ky = 0; ky < g.KH; ++ky)
        for (int kx = 0; kx < g.KW; ++kx)
            for (int a = 0; a < nv; ++a)
                for (int e = 0; e < nv; ++e) {
                    // virtual input pixel (vy, vx) = output position * stride + tap - pad
                    const int vy = (g.ups ? 2 * iy + a : iy) + g.pad_t - ky, vx = (g.ups ? 2 * ix + e : ix) + g.pad_l - kx;
                    if (vy < 0 || vx < 0 || vy % g.stride || vx % g.stride) continue;
                    const int oy = vy / g.stride, ox = vx / g.stride;
                    if (oy >= g.Ho || ox >= g.Wo) continue;
                    const int64_t m = ((int64_t)b * g.Ho + oy) * g.Wo + ox;
                    acc += dcol[m * K + (ky * g.KW + kx) * g.Cin_pad + c];
                }
    dx[p * ldx + c] = acc;
}

// packed weight gradient [Cout_pad][taps][Cin_pad] -> torch layout [Cout][Cin][KH][KW]
__global__ void unpack_weight_grad_kernel(const float *__restrict__ gp, int ldg, float *__restrict__ g, int Cout, int Cin, int taps,
                                          int Cin_pad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Cout * Cin * taps) return;
    const int t = (int)(i % taps), c = (int)((i / taps) % Cin), n = (int)(i / ((int64_t)taps * Cin));
    g[i] = gp[(int64_t)n * ldg + t * Cin_pad + c];
}

// ---- column sums of a [M][N] matrix (bias gradients): partial sums over row chunks, then the same kernel over the partials.
// Thread layout of every column reduction in this file: 64 adjacent columns x 4 row lanes per workgroup (a row segment of
// 256 bytes is one coalesced access), the row lanes joined through LDS in a fixed order.
constexpr int CT = 64, RL = 4;
template <typename T>
__device__ __forceinline__ T join_row_lanes(T v, T (*sh)[CT]) {          // sum over the RL row lanes of a column; valid for rl == 0
    const int cl = threadIdx.x % CT, rl = threadIdx.x / CT;
    sh[rl][cl] = v;
    __syncthreads();
    T t = sh[0][cl];
#pragma unroll
    for (int k = 1; k < RL; ++k) t += sh[k][cl];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ a, int lda, float *__restrict__ out, int M, int N,
                                                     int rows_per_chunk) {
    __shared__ float sh[RL][CT];
    const int cl = threadIdx.x % CT, rl = threadIdx.x / CT;
    const int n = blockIdx.x * CT + cl;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float s = 0.f;
    if (n < N)
        for (int r = r0 + rl; r < r1; r += RL) s += a[(int64_t)r * lda + n];
    s = join_row_lanes(s, sh);
    if (rl == 0 && n < N) out[(int64_t)blockIdx.y * N + n] = s;
}

// ---- GroupNorm(+swish) backward.  Forward: xh = (x - mean) rstd, n = gamma xh + beta, y = swish ? n sigmoid(n) : n.
// With g = dL/dn:  dgamma_c = sum g xh,  dbeta_c = sum g,
//                  dx = rstd (g gamma - mean_grp(g gamma) - xh mean_grp(g gamma xh))       (means over the group's elements)
__device__ __forceinline__ float dswish(float n) {
    const float s = 1.0f / (1.0f + __expf(-n));
    return s * (1.0f + n * (1.0f - s));
}

// Pass 1a: per (image, row chunk, channel) partial sums of g and g xh in fp64 (64 channels x 4 row lanes per workgroup).
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                             const float *__restrict__ mean_rstd, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, int swish, double *__restrict__ part,
                                                             int HW, int C, int groups, int rows_per_chunk) {
    __shared__ double sh[RL][CT];
    const int cl = threadIdx.x % CT, rl = threadIdx.x / CT;
    const int c = blockIdx.x * CT + cl, chunk = blockIdx.y, b = blockIdx.z, nchunk = gridDim.y;
    const int r0 = chunk * rows_per_chunk, r1 = min(HW, r0 + rows_per_chunk);
    double s_g = 0.0, s_gx = 0.0;
    if (c < C) {
        const int grp = c / (C / groups);
        const float mean = mean_rstd[(b * groups + grp) * 2], rstd = mean_rstd[(b * groups + grp) * 2 + 1];
        const float ga = gamma[c], be = beta[c];
        for (int p = r0 + rl; p < r1; p += RL) {
            const int64_t o = ((int64_t)b * HW + p) * C + c;
            const float xh = (x[o] - mean) * rstd;
            float g = dy[o];
            if (swish) g *= dswish(ga * xh + be);
            s_g += (double)g;
            s_gx += (double)g * (double)xh;
        }
    }
    s_g = join_row_lanes(s_g, sh);
    s_gx = join_row_lanes(s_gx, sh);
    if (rl == 0 && c < C) {
        double *o = part + (((int64_t)b * nchunk + chunk) * C + c) * 2;
        o[0] = s_g;
        o[1] = s_gx;
    }
}

// Pass 1b: one workgroup per (image, group): the chunks of its channels added in order -> dbeta_b, dgamma_b per channel and
// the two group means
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const double *__restrict__ part, int nchunk, const float *__restrict__ gamma,
                                                            float *__restrict__ dgamma_b, float *__restrict__ dbeta_b,
                                                            float *__restrict__ gmeans, int HW, int C, int groups) {
    const int b = blockIdx.x / groups, grp = blockIdx.x % groups, cpg = C / groups;
    __shared__ double sh[256][2];
    __shared__ double tot[2];
    if (threadIdx.x < 2) tot[threadIdx.x] = 0.0;
    for (int cc = 0; cc < cpg; ++cc) {
        const int c = grp * cpg + cc;
        double s_g = 0.0, s_gx = 0.0;
        for (int k = threadIdx.x; k < nchunk; k += 256) {
            const double *q = part + (((int64_t)b * nchunk + k) * C + c) * 2;
            s_g += q[0];
            s_gx += q[1];
        }
        sh[threadIdx.x][0] = s_g;
        sh[threadIdx.x][1] = s_gx;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) {
                sh[threadIdx.x][0] += sh[threadIdx.x + o][0];
                sh[threadIdx.x][1] += sh[threadIdx.x + o][1];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            dbeta_b[b * C + c] = (float)sh[0][0];
            dgamma_b[b * C + c] = (float)sh[0][1];
            tot[0] += (double)gamma[c] * sh[0][0];
            tot[1] += (double)gamma[c] * sh[0][1];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)cpg * HW;
        gmeans[(b * groups + grp) * 2] = (float)(tot[0] / n);
        gmeans[(b * groups + grp) * 2 + 1] = (float)(tot[1] / n);
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                           const float *__restrict__ mean_rstd, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, int swish, const float *__restrict__ gmeans,
                                                           float *__restrict__ dx, int64_t total, int HW, int C, int groups) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int b = (int)(i / ((int64_t)HW * C));
    const int grp = c / (C / groups);
    const float mean = mean_rstd[(b * groups + grp) * 2], rstd = mean_rstd[(b * groups + grp) * 2 + 1];
    const float xh = (x[i] - mean) * rstd;
    float g = dy[i];
    if (swish) g *= dswish(gamma[c] * xh + beta[c]);
    dx[i] = rstd * (g * gamma[c] - gmeans[(b * groups + grp) * 2] - xh * gmeans[(b * groups + grp) * 2 + 1]);
}

// ---- soft-max backward over rows: p = softmax(scale * s) -> ds = scale * p (dp - sum_j dp_j p_j); one workgroup per row
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float *__restrict__ p, const float *__restrict__ dp,
                                                               float *__restrict__ ds, int cols, int ld, float scale) {
    const int64_t r = blockIdx.x;
    __shared__ double sh[256];
    double s = 0.0;
    for (int j = threadIdx.x; j < cols; j += 256) s += (double)dp[r * ld + j] * (double)p[r * ld + j];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    const float dot = (float)sh[0];
    for (int j = threadIdx.x; j < cols; j += 256) ds[r * ld + j] = scale * p[r * ld + j] * (dp[r * ld + j] - dot);
}

// ---- L1 reconstruction loss over n_valid of ld columns: grad = sign(rec - target) * gscale (0 on the padding columns) and
// per-workgroup partial sums of |rec - target| (doubles; the host adds the few hundred partials)
__global__ __launch_bounds__(256) void l1_loss_grad_kernel(const float *__restrict__ rec, const float *__restrict__ target,
                                                           float *__restrict__ grad, double *__restrict__ partial, int64_t rows, int C,
                                                           int ld_rec, int ld_grad, float gscale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double a = 0.0;
    if (i < rows * ld_grad) {
        const int64_t r = i / ld_grad;
        const int c = (int)(i - r * ld_grad);
        float gv = 0.f;
        if (c < C) {
            const float d = rec[r * ld_rec + c] - target[r * C + c];
            a = (double)fabsf(d);
            gv = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
        }
        grad[i] = gv;
    }
    __shared__ double sh[256];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// ---- quantiser backward (quantize.py:296-304, legacy form): loss_q = mean((zq.detach - z)^2) + beta mean((zq - z.detach)^2),
// z_q = z + (zq - z).detach  =>  dL/dz = dzq + 2 c_commit (z - zq),   c_commit = codebook_weight / numel
__global__ void vq_bwd_kernel(const float *__restrict__ dzq, const float *__restrict__ z, const float *__restrict__ zq,
                              float *__restrict__ dz, int64_t n, float c2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dz[i] = dzq[i] + c2 * (z[i] - zq[i]);
}

// codebook gradient of the same loss: dE[k] = 2 beta c sum_{t: idx_t = k} (zq_t - z_t); thread = (code, 4 channels), scanning
// the tokens in order (T is a few hundred on this path; deterministic, no atomics)
__global__ void vq_codebook_grad_kernel(const int64_t *__restrict__ idx, const float *__restrict__ z, const float *__restrict__ zq,
                                        float *__restrict__ dE, int T, int n_e, int D, float c2b) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_e * D) return;
    const int k = (int)(i / D), d = (int)(i - (int64_t)k * D);
    float s = 0.f;
    for (int t = 0; t < T; ++t)
        if (idx[t] == k) s += zq[(int64_t)t * D + d] - z[(int64_t)t * D + d];
    dE[i] = c2b * s;
}

__global__ void axpby_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, int64_t n,
                             float alpha, float beta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}

// torch.optim.Adam (no weight decay, no amsgrad), one launch per parameter tensor:
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                            int64_t n, float b1, float b2, float step_size, float inv_sqrt_bc2, float eps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
}

// the same update for a whole parameter SET in one launch (opt.step() of the reference walks ~160-350 tensors): workgroup b takes
// elements [block_off[b], block_off[b] + 4096) of tensor block_tensor[b]; the four pointer tables are device arrays
constexpr int ADAM_CHUNK = 4096;
__global__ __launch_bounds__(256) void adam_multi_kernel(float *const *__restrict__ ps, const float *const *__restrict__ gs,
                                                         float *const *__restrict__ ms, float *const *__restrict__ vs,
                                                         const int64_t *__restrict__ ns, const int32_t *__restrict__ block_tensor,
                                                         const int64_t *__restrict__ block_off, float b1, float b2, float step_size,
                                                         float inv_sqrt_bc2, float eps) {
    const int t = block_tensor[blockIdx.x];
    const int64_t n = ns[t], i0 = block_off[blockIdx.x];
    float *__restrict__ p = ps[t], *__restrict__ m = ms[t], *__restrict__ v = vs[t];
    const float *__restrict__ g = gs[t];
#pragma unroll 4
    for (int k = 0; k < ADAM_CHUNK / 256; ++k) {
        const int64_t i = i0 + k * 256 + threadIdx.x;
        if (i < n) {
            const float gi = g[i];
            const float mi = b1 * m[i] + (1.0f - b1) * gi;
            const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
            m[i] = mi;
            v[i] = vi;
            p[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
        }
    }
}

// ---- PatchGAN discriminator pieces (modules/discriminator/model.py: Conv 4x4 -> [BatchNorm2d] -> LeakyReLU(0.2)) ----
// BatchNorm2d in training mode over the rows (= batch x pixels) of an NHWC matrix [rows][C]: per-channel partial sums of x and
// x^2 over row chunks (fp64), folded by bn_fold_kernel into {mean, rstd} (biased variance, as F.batch_norm normalises) and
// the running statistics (momentum update with the UNBIASED variance, like nn.BatchNorm2d)
__global__ __launch_bounds__(256) void bn_partial_kernel(const float *__restrict__ x, int ld, double *__restrict__ part, int rows, int C,
                                                         int rows_per_chunk) {
    __shared__ double sh[RL][CT];
    const int cl = threadIdx.x % CT, rl = threadIdx.x / CT;
    const int c = blockIdx.x * CT + cl;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    double s = 0.0, ss = 0.0;
    if (c < C)
        for (int r = r0 + rl; r < r1; r += RL) {
            const double v = (double)x[(int64_t)r * ld + c];
            s += v;
            ss += v * v;
        }
    s = join_row_lanes(s, sh);
    ss = join_row_lanes(ss, sh);
    if (rl == 0 && c < C) {
        part[((int64_t)blockIdx.y * C + c) * 2] = s;
        part[((int64_t)blockIdx.y * C + c) * 2 + 1] = ss;
    }
}

__global__ void bn_fold_kernel(const double *__restrict__ part, int chunks, float *__restrict__ mean_rstd, float *__restrict__ run_mean,
                               float *__restrict__ run_var, int rows, int C, float eps, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < chunks; ++k) {
        s += part[((int64_t)k * C + c) * 2];
        ss += part[((int64_t)k * C + c) * 2 + 1];
    }
    const double mean = s / rows;
    double var = ss / rows - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_rstd[c * 2] = (float)mean;
    mean_rstd[c * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        const double unb = rows > 1 ? var * rows / (rows - 1.0) : var;
        run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * (float)mean;
        run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)unb;
    }
}

// y = lrelu(n), n = has_bn ? gamma (x - mean) rstd + beta : x
__global__ void bn_lrelu_fwd_kernel(const float *__restrict__ x, const float *__restrict__ mean_rstd, const float *__restrict__ gamma,
                                    const float *__restrict__ beta, float *__restrict__ y, int64_t total, int C, float slope) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    float n = x[i];
    if (mean_rstd) n = gamma[c] * ((n - mean_rstd[c * 2]) * mean_rstd[c * 2 + 1]) + beta[c];
    y[i] = n > 0.f ? n : slope * n;
}

// g = dy * lrelu'(n); partial column sums of g and g * xh over row chunks (xh = 1 without BatchNorm); g is written to `gbuf`
__global__ __launch_bounds__(256) void bn_lrelu_bwd_partial_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                                   const float *__restrict__ mean_rstd, const float *__restrict__ gamma,
                                                                   const float *__restrict__ beta, float *__restrict__ gbuf,
                                                                   double *__restrict__ part, int rows, int C, int rows_per_chunk,
                                                                   float slope) {
    __shared__ double sh[RL][CT];
    const int cl = threadIdx.x % CT, rl = threadIdx.x / CT;
    const int c = blockIdx.x * CT + cl;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    double s = 0.0, sx = 0.0;
    if (c < C) {
        const float mean = mean_rstd ? mean_rstd[c * 2] : 0.f, rstd = mean_rstd ? mean_rstd[c * 2 + 1] : 1.f;
        const float ga = mean_rstd ? gamma[c] : 1.f, be = mean_rstd ? beta[c] : 0.f;
        for (int r = r0 + rl; r < r1; r += RL) {
            const int64_t o = (int64_t)r * C + c;
            const float xh = mean_rstd ? (x[o] - mean) * rstd : x[o];
            const float n = mean_rstd ? ga * xh + be : xh;
            const float g = dy[o] * (n > 0.f ? 1.f : slope);
            gbuf[o] = g;
            s += (double)g;
            sx += (double)g * (double)xh;
        }
    }
    s = join_row_lanes(s, sh);
    sx = join_row_lanes(sx, sh);
    if (rl == 0 && c < C) {
        part[((int64_t)blockIdx.y * C + c) * 2] = s;
        part[((int64_t)blockIdx.y * C + c) * 2 + 1] = sx;
    }
}

__global__ void bn_bwd_fold_kernel(const double *__restrict__ part, int chunks, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                   float *__restrict__ means, int rows, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, sx = 0.0;
    for (int k = 0; k < chunks; ++k) {
        s += part[((int64_t)k * C + c) * 2];
        sx += part[((int64_t)k * C + c) * 2 + 1];
    }
    dbeta[c] = (float)s;
    dgamma[c] = (float)sx;
    means[c * 2] = (float)(s / rows);
    means[c * 2 + 1] = (float)(sx / rows);
}

// dx = gamma rstd (g - mean(g) - xh mean(g xh)) with BatchNorm, else dx = g (already in gbuf)
__global__ void bn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ gbuf, const float *__restrict__ mean_rstd,
                                    const float *__restrict__ gamma, const float *__restrict__ means, float *__restrict__ dx,
                                    int64_t total, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const float xh = (x[i] - mean_rstd[c * 2]) * mean_rstd[c * 2 + 1];
    dx[i] = gamma[c] * mean_rstd[c * 2 + 1] * (gbuf[i] - means[c * 2] - xh * means[c * 2 + 1]);
}

// hinge terms over a logit map (vqperceptual.py:17-21, 98): mode +1: relu(1 + l) (fake, discriminator step), -1: relu(1 - l)
// (real), 0: l itself (generator step: g_loss = -mean(l), grad = gscale everywhere).  grad[i] = d(term)/dl * gscale,
// partial[workgroup] = sum of the terms
__global__ __launch_bounds__(256) void hinge_kernel(const float *__restrict__ l, float *__restrict__ grad, double *__restrict__ partial,
                                                    int64_t n, int mode, float gscale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double a = 0.0;
    if (i < n) {
        const float v = l[i];
        if (mode == 0) {
            a = (double)v;
            grad[i] = gscale;
        } else if (mode == 2 || mode == -2) {
            // vanilla GAN loss (vqperceptual.py:24-28): softplus(s l), s = +1 (fake) / -1 (real); d/dl = s sigmoid(s l).
            // softplus as torch evaluates it: x for x > 20 (its threshold), log1p(exp(x)) otherwise
            const float sgn = mode > 0 ? 1.0f : -1.0f, x = sgn * v;
            a = (double)(x > 20.f ? x : log1pf(expf(x)));
            grad[i] = sgn * gscale / (1.0f + expf(-x));
        } else {
            const float t = 1.0f + (float)mode * v;
            a = t > 0.f ? (double)t : 0.0;
            grad[i] = t > 0.f ? (float)mode * gscale : 0.f;
        }
    }
    __shared__ double sh[256];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// partial sums of squares (the two gradient norms of calculate_adaptive_weight, vqperceptual.py:63-75)
__global__ __launch_bounds__(256) void sumsq_kernel(const float *__restrict__ a, double *__restrict__ partial, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    __shared__ double sh[256];
    sh[threadIdx.x] = i < n ? (double)a[i] * (double)a[i] : 0.0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// ---- LPIPS pieces (modules/losses/lpips.py): MaxPool2d(2, 2), the input ScalingLayer, and per feature level
//      val_b = mean_p sum_c w_c (f0 / (|f0| + eps) - f1 / (|f1| + eps))^2     (normalize_tensor, NetLinLayer, spatial_average)
__global__ void maxpool2x2_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * Ho * Wo * C) return;
    const int c = (int)(i % C);
    const int ox = (int)((i / C) % Wo), oy = (int)((i / ((int64_t)C * Wo)) % Ho), b = (int)(i / ((int64_t)C * Wo * Ho));
    const float *p = x + (((int64_t)b * H + 2 * oy) * W + 2 * ox) * C + c;
    y[i] = fmaxf(fmaxf(p[0], p[C]), fmaxf(p[(int64_t)W * C], p[(int64_t)W * C + C]));
}

// the gradient goes to the first maximum of the window in row-major order (what torch's max_pool2d backward does)
__global__ void maxpool2x2_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dx, int B, int H,
                                      int W, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * H * W * C) return;
    const int c = (int)(i % C);
    const int ix = (int)((i / C) % W), iy = (int)((i / ((int64_t)C * W)) % H), b = (int)(i / ((int64_t)C * W * H));
    const int Ho = H / 2, Wo = W / 2, oy = iy >> 1, ox = ix >> 1;
    float g = 0.f;
    if (oy < Ho && ox < Wo) {
        const float *p = x + (((int64_t)b * H + 2 * oy) * W + 2 * ox) * C + c;
        const float v[4] = {p[0], p[C], p[(int64_t)W * C], p[(int64_t)W * C + C]};
        int best = 0;
        for (int k = 1; k < 4; ++k)
            if (v[k] > v[best]) best = k;
        if (best == ((iy & 1) * 2 + (ix & 1))) g = dy[(((int64_t)b * Ho + oy) * Wo + ox) * C + c];
    }
    dx[i] = g;
}

// y[..., c] = (x[..., c] - shift[c]) * inv_scale[c] for c < Cv, 0 for the padding channels; backward: dx = dy * inv_scale
__global__ void channel_affine_kernel(const float *__restrict__ x, int ldx, float *__restrict__ y, int ldy, int64_t rows, int Cv,
                                      f32x4 shift, f32x4 inv_scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ldy) return;
    const int64_t r = i / ldy;
    const int c = (int)(i - r * ldy);
    y[i] = c < Cv ? (x[r * ldx + c] - shift[c]) * inv_scale[c] : 0.f;
}

// one thread per pixel: partial[workgroup] = sum over its pixels of sum_c w_c (a_c - b_c)^2
__global__ __launch_bounds__(256) void lpips_level_fwd_kernel(const float *__restrict__ f0, const float *__restrict__ f1,
                                                              const float *__restrict__ w, double *__restrict__ partial, int HW, int C,
                                                              float eps) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (p < HW) {
        const float *a = f0 + ((int64_t)b * HW + p) * C, *q = f1 + ((int64_t)b * HW + p) * C;
        float s0 = 0.f, s1 = 0.f;
        for (int c = 0; c < C; ++c) {
            s0 += a[c] * a[c];
            s1 += q[c] * q[c];
        }
        const float i0 = 1.0f / (sqrtf(s0) + eps), i1 = 1.0f / (sqrtf(s1) + eps);
        float acc = 0.f;
        for (int c = 0; c < C; ++c) {
            const float d = a[c] * i0 - q[c] * i1;
            acc += w[c] * d * d;
        }
        v = (double)acc;
    }
    __shared__ double sh[256];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(int64_t)b * gridDim.x + blockIdx.x] = sh[0];
}

// d val_b / d f0 * gscale: with n = |f0|, a = f0 / (n + eps), u_c = 2 w_c (a_c - b_c):
//   df0_k = gscale / HW * (u_k / (n + eps) - f0_k (sum_c u_c f0_c) / (n (n + eps)^2))
__global__ __launch_bounds__(256) void lpips_level_bwd_kernel(const float *__restrict__ f0, const float *__restrict__ f1,
                                                              const float *__restrict__ w, float *__restrict__ df0, int HW, int C,
                                                              float eps, float gscale) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float *a = f0 + ((int64_t)b * HW + p) * C, *q = f1 + ((int64_t)b * HW + p) * C;
    float *o = df0 + ((int64_t)b * HW + p) * C;
    float s0 = 0.f, s1 = 0.f;
    for (int c = 0; c < C; ++c) {
        s0 += a[c] * a[c];
        s1 += q[c] * q[c];
    }
    const float n0 = sqrtf(s0), i0 = 1.0f / (n0 + eps), i1 = 1.0f / (sqrtf(s1) + eps);
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot += 2.0f * w[c] * (a[c] * i0 - q[c] * i1) * a[c];
    const float k2 = n0 > 0.f ? dot * i0 * i0 / n0 : 0.f;
    const float g = gscale / (float)HW;
    for (int c = 0; c < C; ++c) o[c] = g * (2.0f * w[c] * (a[c] * i0 - q[c] * i1) * i0 - a[c] * k2);
}

ConvGeo geo_of(const sgam_conv_desc *d, int cin_pad) {
    ConvGeo g;
    g.B = d->B; g.Hi = d->Hi; g.Wi = d->Wi; g.Cin = d->Cin; g.Cin_pad = cin_pad; g.Ho = d->Ho; g.Wo = d->Wo; g.KH = d->KH; g.KW = d->KW;
    g.stride = d->stride; g.pad_t = d->pad_t; g.pad_l = d->pad_l; g.ups = d->upsample2x ? 1 : 0;
    return g;
}

bool geo_ok(const sgam_conv_desc *d, int cin_pad) {
    return d && d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Cin > 0 && cin_pad >= d->Cin && d->Ho > 0 && d->Wo > 0 && d->KH > 0 && d->KW > 0 &&
           d->stride > 0 && d->pad_t >= 0 && d->pad_l >= 0;
}

}  // namespace

extern "C" int sgam_im2col_t_f32(const sgam_conv_desc *d, const float *x, float *col_t, int32_t cin_pad, int64_t ld_m, void *stream) {
    if (!geo_ok(d, cin_pad) || !x || !col_t || d->lda < d->Cin || ld_m < (int64_t)d->B * d->Ho * d->Wo) return SGAM_EINVAL;
    const ConvGeo g = geo_of(d, cin_pad);
    const int64_t total = (int64_t)g.KH * g.KW * g.Cin_pad * g.B * g.Ho * g.Wo;
    if (total >= ((int64_t)1 << 31) * 256) return SGAM_EINVAL;
    SGAM_KLAUNCH(im2col_t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sgam_stream(stream), x, d->lda, col_t, ld_m, g);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_col2im_gather_f32(const sgam_conv_desc *d, const float *dcol, float *dx, int32_t cin_pad, void *stream) {
    if (!geo_ok(d, cin_pad) || !dcol || !dx || d->lda < d->Cin) return SGAM_EINVAL;
    const ConvGeo g = geo_of(d, cin_pad);
    const int64_t total = (int64_t)g.B * g.Hi * g.Wi * g.Cin;
    SGAM_KLAUNCH(col2im_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sgam_stream(stream), dcol, dx, d->lda, g);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_unpack_conv_weight_grad_f32(const float *grad_packed, int32_t ld, float *grad_oihw, int32_t Cout, int32_t Cin,
                                                int32_t KH, int32_t KW, int32_t Cin_pad, void *stream) {
    if (!grad_packed || !grad_oihw || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || Cin_pad < Cin || ld < KH * KW * Cin_pad)
        return SGAM_EINVAL;
    const int64_t total = (int64_t)Cout * Cin * KH * KW;
    SGAM_KLAUNCH(unpack_weight_grad_kernel, dim3(sgam_cdiv(total, 256)), dim3(256), 0, sgam_stream(stream), grad_packed, ld, grad_oihw,
                 Cout, Cin, KH * KW, Cin_pad);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// out[N] = column sums of a[M][N]; workspace >= sgam_colsum_workspace_bytes(M, N)
extern "C" int64_t sgam_colsum_workspace_bytes(int32_t M, int32_t N) {
    if (M <= 0 || N <= 0) return -1;
    return (int64_t)sgam_cdiv(M, 256) * N * 4;
}

extern "C" int sgam_colsum_f32(const float *a, int32_t lda, float *out, int32_t M, int32_t N, void *workspace, int64_t workspace_bytes,
                               void *stream) {
    if (!a || !out || M <= 0 || N <= 0 || lda < N || !workspace || workspace_bytes < sgam_colsum_workspace_bytes(M, N)) return SGAM_EINVAL;
    const int chunks = sgam_cdiv(M, 256);
    hipStream_t s = sgam_stream(stream);
    SGAM_KLAUNCH(colsum_kernel, dim3(sgam_cdiv(N, CT), chunks), dim3(256), 0, s, a, lda, (float *)workspace, M, N, 256);
    SGAM_LAUNCH_CHECK();
    SGAM_KLAUNCH(colsum_kernel, dim3(sgam_cdiv(N, CT), 1), dim3(256), 0, s, (const float *)workspace, N, out, chunks, N, chunks);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int64_t sgam_groupnorm_bwd_workspace_bytes(int32_t B, int32_t HW, int32_t C) {
    if (B <= 0 || HW <= 0 || C <= 0) return -1;
    return (int64_t)B * sgam_cdiv(HW, 256) * C * 2 * 8;
}

extern "C" int sgam_groupnorm_bwd_nhwc_f32(const float *x, const float *dy, const float *mean_rstd, const float *gamma, const float *beta,
                                           int32_t swish, float *dx, float *dgamma_b, float *dbeta_b, float *group_means, int32_t B,
                                           int32_t HW, int32_t C, int32_t groups, void *workspace, int64_t workspace_bytes,
                                           void *stream) {
    if (!x || !dy || !mean_rstd || !gamma || !beta || !dx || !dgamma_b || !dbeta_b || !group_means || B <= 0 || HW <= 0 || C <= 0 ||
        groups <= 0 || C % groups)
        return SGAM_EINVAL;
    hipStream_t s = sgam_stream(stream);
    const int nchunk = sgam_cdiv(HW, 256);
    if (!workspace || workspace_bytes < sgam_groupnorm_bwd_workspace_bytes(B, HW, C)) return SGAM_EWORKSPACE;
    SGAM_KLAUNCH(gn_bwd_partial_kernel, dim3(sgam_cdiv(C, CT), nchunk, B), dim3(256), 0, s, x, dy, mean_rstd, gamma, beta, swish ? 1 : 0,
                 (double *)workspace, HW, C, groups, 256);
    SGAM_LAUNCH_CHECK();
    SGAM_KLAUNCH(gn_bwd_reduce_kernel, dim3(B * groups), dim3(256), 0, s, (const double *)workspace, nchunk, gamma, dgamma_b, dbeta_b,
                 group_means, HW, C, groups);
    SGAM_LAUNCH_CHECK();
    const int64_t total = (int64_t)B * HW * C;
    SGAM_KLAUNCH(gn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, dy, mean_rstd, gamma, beta, swish ? 1 : 0,
                 group_means, dx, total, HW, C, groups);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_softmax_bwd_rows_f32(const float *p, const float *dp, float *ds, int32_t rows, int32_t cols, int32_t ld, float scale,
                                         void *stream) {
    if (!p || !dp || !ds || rows <= 0 || cols <= 0 || ld < cols) return SGAM_EINVAL;
    SGAM_KLAUNCH(softmax_bwd_rows_kernel, dim3(rows), dim3(256), 0, sgam_stream(stream), p, dp, ds, cols, ld, scale);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// partial: sgam_cdiv(rows * ld_grad, 256) doubles
extern "C" int sgam_l1_loss_grad_f32(const float *rec, const float *target, float *grad, double *partial, int64_t rows, int32_t C,
                                     int32_t ld_rec, int32_t ld_grad, float grad_scale, void *stream) {
    if (!rec || !target || !grad || !partial || rows <= 0 || C <= 0 || ld_rec < C || ld_grad < C) return SGAM_EINVAL;
    const int64_t total = rows * ld_grad;
    SGAM_KLAUNCH(l1_loss_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sgam_stream(stream), rec, target, grad, partial,
                 rows, C, ld_rec, ld_grad, grad_scale);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_vq_bwd_f32(const float *dzq, const float *z, const float *zq, float *dz, int64_t n, float two_c, void *stream) {
    if (!dzq || !z || !zq || !dz || n <= 0) return SGAM_EINVAL;
    SGAM_KLAUNCH(vq_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sgam_stream(stream), dzq, z, zq, dz, n, two_c);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_vq_codebook_grad_f32(const int64_t *indices, const float *z, const float *zq, float *d_codebook, int32_t T,
                                         int32_t n_e, int32_t D, float two_c_beta, void *stream) {
    if (!indices || !z || !zq || !d_codebook || T <= 0 || n_e <= 0 || D <= 0) return SGAM_EINVAL;
    const int64_t total = (int64_t)n_e * D;
    SGAM_KLAUNCH(vq_codebook_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sgam_stream(stream), indices, z, zq,
                 d_codebook, T, n_e, D, two_c_beta);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// out = alpha a + beta b (b may be NULL)
extern "C" int sgam_axpby_f32(const float *a, const float *b, float *out, int64_t n, float alpha, float beta, void *stream) {
    if (!a || !out || n <= 0) return SGAM_EINVAL;
    SGAM_KLAUNCH(axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sgam_stream(stream), a, b, out, n, alpha, beta);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                                  float beta2, float eps, int32_t step, void *stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return SGAM_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    SGAM_KLAUNCH(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sgam_stream(stream), param, grad, exp_avg, exp_avg_sq, n,
                 beta1, beta2, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), eps);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_adam_multi_step_f32(float *const *params, const float *const *grads, float *const *exp_avgs, float *const *exp_avg_sqs,
                                        const int64_t *numels, const int32_t *block_tensor, const int64_t *block_off, int32_t n_blocks,
                                        float lr, float beta1, float beta2, float eps, int32_t step, void *stream) {
    if (!params || !grads || !exp_avgs || !exp_avg_sqs || !numels || !block_tensor || !block_off || n_blocks <= 0 || step <= 0)
        return SGAM_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    SGAM_KLAUNCH(adam_multi_kernel, dim3((unsigned)n_blocks), dim3(256), 0, sgam_stream(stream), params, grads, exp_avgs, exp_avg_sqs,
                 numels, block_tensor, block_off, beta1, beta2, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), eps);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// ---- BatchNorm2d (training mode) + LeakyReLU of the PatchGAN discriminator, on [rows = B*H*W][C] NHWC matrices ----
// workspace: sgam_cdiv(rows, 256) * C * 2 doubles
extern "C" int64_t sgam_batchnorm_workspace_bytes(int32_t rows, int32_t C) {
    if (rows <= 0 || C <= 0) return -1;
    return (int64_t)sgam_cdiv(rows, 256) * C * 2 * 8;
}

// mean_rstd [C][2] out; running_mean / running_var updated in place when given (momentum 0.1 in the reference)
extern "C" int sgam_batchnorm_stats_f32(const float *x, float *mean_rstd, float *running_mean, float *running_var, int32_t rows, int32_t C,
                                        float eps, float momentum, void *workspace, int64_t workspace_bytes, void *stream) {
    if (!x || !mean_rstd || rows <= 0 || C <= 0 || (running_mean == nullptr) != (running_var == nullptr) || !workspace ||
        workspace_bytes < sgam_batchnorm_workspace_bytes(rows, C))
        return SGAM_EINVAL;
    const int chunks = sgam_cdiv(rows, 256);
    hipStream_t s = sgam_stream(stream);
    SGAM_KLAUNCH(bn_partial_kernel, dim3(sgam_cdiv(C, CT), chunks), dim3(256), 0, s, x, C, (double *)workspace, rows, C, 256);
    SGAM_LAUNCH_CHECK();
    SGAM_KLAUNCH(bn_fold_kernel, dim3(sgam_cdiv(C, 256)), dim3(256), 0, s, (const double *)workspace, chunks, mean_rstd, running_mean,
                 running_var, rows, C, eps, momentum);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// y = leaky_relu(mean_rstd ? gamma * (x - mean) * rstd + beta : x, slope)
extern "C" int sgam_bn_lrelu_fwd_f32(const float *x, const float *mean_rstd, const float *gamma, const float *beta, float *y, int32_t rows,
                                     int32_t C, float slope, void *stream) {
    if (!x || !y || rows <= 0 || C <= 0 || (mean_rstd && (!gamma || !beta))) return SGAM_EINVAL;
    const int64_t total = (int64_t)rows * C;
    SGAM_KLAUNCH(bn_lrelu_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sgam_stream(stream), x, mean_rstd, gamma, beta, y,
                 total, C, slope);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// backward of the same: dx [rows][C]; dgamma / dbeta [C] (BatchNorm only); scratch: gbuf [rows][C] floats (may alias dx),
// means [C][2] floats, workspace as for the statistics
extern "C" int sgam_bn_lrelu_bwd_f32(const float *x, const float *dy, const float *mean_rstd, const float *gamma, const float *beta,
                                     float *dx, float *dgamma, float *dbeta, float *gbuf, float *means, int32_t rows, int32_t C,
                                     float slope, void *workspace, int64_t workspace_bytes, void *stream) {
    if (!x || !dy || !dx || !gbuf || rows <= 0 || C <= 0 || !workspace || workspace_bytes < sgam_batchnorm_workspace_bytes(rows, C))
        return SGAM_EINVAL;
    if (mean_rstd && (!gamma || !beta || !dgamma || !dbeta || !means || gbuf == dx)) return SGAM_EINVAL;
    if (!mean_rstd && gbuf != dx) return SGAM_EINVAL;          // without BatchNorm the masked gradient IS dx
    const int chunks = sgam_cdiv(rows, 256);
    hipStream_t s = sgam_stream(stream);
    SGAM_KLAUNCH(bn_lrelu_bwd_partial_kernel, dim3(sgam_cdiv(C, CT), chunks), dim3(256), 0, s, x, dy, mean_rstd, gamma, beta, gbuf,
                 (double *)workspace, rows, C, 256, slope);
    SGAM_LAUNCH_CHECK();
    if (!mean_rstd) return SGAM_OK;
    SGAM_KLAUNCH(bn_bwd_fold_kernel, dim3(sgam_cdiv(C, 256)), dim3(256), 0, s, (const double *)workspace, chunks, dgamma, dbeta, means, rows,
                 C);
    SGAM_LAUNCH_CHECK();
    const int64_t total = (int64_t)rows * C;
    SGAM_KLAUNCH(bn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, gbuf, mean_rstd, gamma, means, dx, total, C);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// partial: sgam_cdiv(n, 256) doubles
extern "C" int sgam_hinge_terms_f32(const float *logits, float *grad, double *partial, int64_t n, int32_t mode, float grad_scale,
                                    void *stream) {
    if (!logits || !grad || !partial || n <= 0 || mode < -2 || mode > 2) return SGAM_EINVAL;
    SGAM_KLAUNCH(hinge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sgam_stream(stream), logits, grad, partial, n, mode, grad_scale);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_sumsq_partial_f32(const float *a, double *partial, int64_t n, void *stream) {
    if (!a || !partial || n <= 0) return SGAM_EINVAL;
    SGAM_KLAUNCH(sumsq_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sgam_stream(stream), a, partial, n);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// ---- LPIPS (modules/losses/lpips.py) ----
extern "C" int sgam_maxpool2x2_f32(const float *x, float *y, int32_t B, int32_t H, int32_t W, int32_t C, void *stream) {
    if (!x || !y || B <= 0 || H < 2 || W < 2 || C <= 0) return SGAM_EINVAL;
    const int64_t total = (int64_t)B * (H / 2) * (W / 2) * C;
    SGAM_KLAUNCH(maxpool2x2_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sgam_stream(stream), x, y, B, H, W, C);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_maxpool2x2_bwd_f32(const float *x, const float *dy, float *dx, int32_t B, int32_t H, int32_t W, int32_t C, void *stream) {
    if (!x || !dy || !dx || B <= 0 || H < 2 || W < 2 || C <= 0) return SGAM_EINVAL;
    const int64_t total = (int64_t)B * H * W * C;
    SGAM_KLAUNCH(maxpool2x2_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sgam_stream(stream), x, dy, dx, B, H, W, C);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// y[rows][ldy] = (x[rows][ldx][:n] - shift) * inv_scale on the first n <= 4 channels, 0 on the rest (ScalingLayer; with shift 0
// also its backward)
extern "C" int sgam_channel_affine_f32(const float *x, int32_t ldx, float *y, int32_t ldy, int64_t rows, int32_t n, const float *shift4,
                                       const float *inv_scale4, void *stream) {
    if (!x || !y || rows <= 0 || n <= 0 || n > 4 || ldx < n || ldy < n || !shift4 || !inv_scale4) return SGAM_EINVAL;
    const f32x4 sh = {shift4[0], shift4[1], shift4[2], shift4[3]}, is = {inv_scale4[0], inv_scale4[1], inv_scale4[2], inv_scale4[3]};
    const int64_t total = rows * ldy;
    SGAM_KLAUNCH(channel_affine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sgam_stream(stream), x, ldx, y, ldy, rows, n, sh,
                 is);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// one feature level: partial [B][ceil(HW / 256)] doubles (val_b = sum of row b / HW); df0 (optional) = gscale * d val_b / d f0
extern "C" int sgam_lpips_level_f32(const float *f0, const float *f1, const float *lin_w, double *partial, float *df0, int32_t B, int32_t HW,
                                    int32_t C, float eps, float grad_scale, void *stream) {
    if (!f0 || !f1 || !lin_w || !partial || B <= 0 || HW <= 0 || C <= 0) return SGAM_EINVAL;
    const dim3 grid(sgam_cdiv(HW, 256), B);
    hipStream_t s = sgam_stream(stream);
    SGAM_KLAUNCH(lpips_level_fwd_kernel, grid, dim3(256), 0, s, f0, f1, lin_w, partial, HW, C, eps);
    SGAM_LAUNCH_CHECK();
    if (df0) {
        SGAM_KLAUNCH(lpips_level_bwd_kernel, grid, dim3(256), 0, s, f0, f1, lin_w, df0, HW, C, eps, grad_scale);
        SGAM_LAUNCH_CHECK();
    }
    return SGAM_OK;
}
