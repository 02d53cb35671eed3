// norm_softmax.hip — row softmax of the attention scores (GroupNorm lives in groupnorm.hip).  HBM-bound (one
// read + one write per element); reductions run on wavefront shuffles (64 lanes) and a small LDS hand-off,
// never on atomics, so results are bit-reproducible run to run.
#include "sgam_common.h"

namespace {

// --------------------------------------------------------------------------------------------
// Row softmax in place, one workgroup (256 lanes) per row, the row kept in registers between the
// max / exp-sum / normalise passes (cols <= 256*4*MAXV).  s = softmax(scale * s).
// --------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void softmax_rows_kernel(float *__restrict__ s, int cols, int ld, float scale, int block) {
    float *row = s + (int64_t)blockIdx.x * ld;
    const int c4 = cols >> 2;
    // block > 0: block-diagonal form — row r keeps the columns of its own block [r / block * block, + block), the others
    // become exact zeros (B images' attention as ONE score matrix: a query never sees another image's keys)
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
                for (int e = 0; e < 4; ++e) {
                    v[k][e] *= scale;
                    mx = fmaxf(mx, v[k][e]);
                }
            } else {
                v[k] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};      // exp(-inf - max) = 0
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

static int softmax_rows_impl(float *sp, int32_t rows, int32_t cols, int32_t ld, float scale, int32_t block, void *stream) {
    if (!sp || rows <= 0 || cols <= 0 || cols % 4 != 0 || ld < cols || ld % 4 != 0) return SGAM_EINVAL;
    if (!sgam_aligned16(sp)) return SGAM_EALIGN;
    hipStream_t s = sgam_stream(stream);
    if (cols <= 1024) {
        SGAM_KLAUNCH((softmax_rows_kernel<1>), dim3(rows), dim3(256), 0, s, sp, cols, ld, scale, block);
    } else if (cols <= 4096) {
        SGAM_KLAUNCH((softmax_rows_kernel<4>), dim3(rows), dim3(256), 0, s, sp, cols, ld, scale, block);
    } else if (cols <= 16384) {
        SGAM_KLAUNCH((softmax_rows_kernel<16>), dim3(rows), dim3(256), 0, s, sp, cols, ld, scale, block);
    } else {
        return SGAM_EINVAL;
    }
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_softmax_rows_f32(float *sp, int32_t rows, int32_t cols, int32_t ld, float scale, void *stream) {
    return softmax_rows_impl(sp, rows, cols, ld, scale, 0, stream);
}

extern "C" int sgam_softmax_rows_blockdiag_f32(float *sp, int32_t rows, int32_t cols, int32_t ld, float scale, int32_t block,
                                               void *stream) {
    if (block <= 0 || block % 4 != 0 || rows % block != 0 || cols % block != 0 || rows / block > cols / block) return SGAM_EINVAL;
    return softmax_rows_impl(sp, rows, cols, ld, scale, block, stream);
}
