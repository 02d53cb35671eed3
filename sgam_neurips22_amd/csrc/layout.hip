// layout.hip — module-boundary layout hops (NCHW <-> NHWC), the VQModel.encode head
// (cat(x, mask) -> 1x1 conv 5->4, model.py:107-113) and the frame-feedback codec
// (inference_pipeline.py:898-911 / :534-537).  All HBM-bound element-wise / transpose kernels.
#include "sgam_common.h"

namespace {

// y[b][p][c] = x[b][c][p] through a 32x33 LDS tile: both sides coalesced.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *__restrict__ x, float *__restrict__ y, int C,
                                                           int HW, int ldy) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float *xb = x + (int64_t)b * C * HW;
    float *yb = y + (int64_t)b * HW * ldy;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + ty + 8 * r, p = p0 + tx;
        tile[ty + 8 * r][tx] = (c < C && p < HW) ? xb[(int64_t)c * HW + p] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int p = p0 + ty + 8 * r, c = c0 + tx;
        if (p < HW && c < C) yb[(int64_t)p * ldy + c] = tile[tx][ty + 8 * r];
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float *__restrict__ x, float *__restrict__ y, int C,
                                                           int HW, int ldx) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float *xb = x + (int64_t)b * HW * ldx;
    float *yb = y + (int64_t)b * C * HW;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int p = p0 + ty + 8 * r, c = c0 + tx;
        tile[ty + 8 * r][tx] = (p < HW && c < C) ? xb[(int64_t)p * ldx + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + ty + 8 * r, p = p0 + tx;
        if (c < C && p < HW) yb[(int64_t)c * HW + p] = tile[tx][ty + 8 * r];
    }
}

// One lane per pixel: 5 inputs (4 planes + mask), 4 outputs, pixel row padded with zeros to ldy floats.
__global__ __launch_bounds__(256) void encode_head_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask,
                                                          const float *__restrict__ w, const float *__restrict__ bias,
                                                          float *__restrict__ y, int HW, int ldy) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float *xb = x + (int64_t)b * 4 * HW;
    float in[5];
#pragma unroll
    for (int c = 0; c < 4; ++c) in[c] = xb[(int64_t)c * HW + p];
    in[4] = mask ? (mask[(int64_t)b * HW + p] ? 1.0f : 0.0f) : 0.0f;
    f32x4 o;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        float acc = bias[n];
#pragma unroll
        for (int c = 0; c < 5; ++c) acc = fmaf(in[c], w[n * 5 + c], acc);
        o[n] = acc;
    }
    f32x4 *row = reinterpret_cast<f32x4 *>(y + ((int64_t)b * HW + p) * ldy);
    row[0] = o;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int k = 1; k < (ldy >> 2); ++k) row[k] = z;
}

// dec (B,4,HW) -> uint8 RGB by truncation, its float re-expansion through the host LUT, metric depth.
// Exact-semantics kernel (explicit rn intrinsics).  dataset_norm: 1 google_earth, 2 clevr-infinite.
__global__ __launch_bounds__(256) void frame_feedback_kernel(const float *__restrict__ dec, const float *__restrict__ lut,
                                                             int dataset_norm, uint8_t *__restrict__ rgb_u8,
                                                             float *__restrict__ rgb_f, float *__restrict__ depth,
                                                             int HW) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float *db = dec + (int64_t)b * 4 * HW;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // np.clip((x + 1) / 2 * 255., 0, 255).astype(np.uint8)
        float v = __fmul_rn(__fdiv_rn(__fadd_rn(db[(int64_t)c * HW + p], 1.0f), 2.0f), 255.0f);
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        const unsigned u = (v != v) ? 0u : (unsigned)v;  // truncation
        if (rgb_u8) rgb_u8[((int64_t)b * HW + p) * 3 + c] = (uint8_t)u;
        if (rgb_f) rgb_f[((int64_t)b * HW + p) * 3 + c] = lut[u];
    }
    if (depth) {
        const float x3 = db[(int64_t)3 * HW + p];
        const float h = __fdiv_rn(__fadd_rn(x3, 1.0f), 2.0f);
        float d;
        if (dataset_norm == 1) {
            const float span = (float)(1.0 / 10.099975586 - 1.0 / 14.765625);
            const float lo = (float)(1.0 / 14.765625);
            d = __fsub_rn(__fdiv_rn(1.0f, __fadd_rn(__fmul_rn(h, span), lo)), 10.0f);
        } else {
            const float span = (float)(1.0 / 7 - 1.0 / 16);
            const float lo = (float)(1.0 / 16);
            d = __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(h, span), lo));
        }
        depth[(int64_t)b * HW + p] = d;
    }
}

// the PNG re-read of prepare_batch_data (inference_pipeline.py:534): uint8 -> lut[u] = float32(u / 127.5 - 1.0)
__global__ __launch_bounds__(256) void rgb_lut_kernel(const uint8_t *__restrict__ u8, const float *__restrict__ lut,
                                                      float *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = lut[u8[i]];
}

}  // namespace

extern "C" int sgam_rgb_u8_to_f32(const uint8_t *rgb_u8, const float *lut256, float *rgb_f, int64_t n, void *stream) {
    if (!rgb_u8 || !lut256 || !rgb_f || n <= 0) return SGAM_EINVAL;
    SGAM_KLAUNCH(rgb_lut_kernel, dim3(sgam_cdiv(n, 256)), dim3(256), 0, sgam_stream(stream), rgb_u8, lut256, rgb_f, n);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// ---- split-fp32 range guard ----
extern "C" int32_t *sgam_i_range_flag = nullptr;
extern "C" int sgam_f32x_set_range_flag(int32_t *device_flag) {
    sgam_i_range_flag = device_flag;
    return SGAM_OK;
}

// ---- kernel timeline (see SGAM_KLAUNCH in sgam_common.h) ----
namespace {
struct ProfRec {
    const char *kernel, *where;
    hipEvent_t e0, e1;
    double flops, bytes;
    int shape[4];
};
constexpr int PROF_MAX = 16384;
ProfRec *g_prof = nullptr;
int g_prof_n = 0, g_prof_events = 0;
double g_work_flops = 0.0, g_work_bytes = 0.0;
int g_shape[4] = {0, 0, 0, 0};
}  // namespace
extern "C" void sgam_i_prof_shape(int m, int n, int k, int ksplit) {
    g_shape[0] = m; g_shape[1] = n; g_shape[2] = k; g_shape[3] = ksplit;
}
extern "C" int sgam_i_prof_on = 0;
extern "C" void sgam_i_prof_work(double flops, double bytes) {
    g_work_flops = flops;
    g_work_bytes = bytes;
}
extern "C" void sgam_i_prof_begin(const char *kernel, const char *where, hipStream_t s) {
    if (!g_prof || g_prof_n >= PROF_MAX) return;
    ProfRec &r = g_prof[g_prof_n];
    if (g_prof_n >= g_prof_events) {
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
        g_prof_events = g_prof_n + 1;
    }
    r.kernel = kernel;
    r.where = where;
    r.flops = g_work_flops;
    r.bytes = g_work_bytes;
    g_work_flops = g_work_bytes = 0.0;
    for (int i = 0; i < 4; ++i) { r.shape[i] = g_shape[i]; g_shape[i] = 0; }
    (void)hipEventRecord(r.e0, s);
}
extern "C" void sgam_i_prof_end(hipStream_t s) {
    if (!g_prof || g_prof_n >= PROF_MAX || g_prof_n >= g_prof_events) return;
    (void)hipEventRecord(g_prof[g_prof_n].e1, s);
    ++g_prof_n;
}
extern "C" int sgam_prof_enable(int32_t on) {
    if (on && !g_prof) g_prof = new ProfRec[PROF_MAX]();
    if (on) g_prof_n = 0;
    sgam_i_prof_on = on ? 1 : 0;
    return SGAM_OK;
}
extern "C" int sgam_prof_mark_empty(void *stream) {   // an event pair around nothing: what a bracket costs by itself
    if (!sgam_i_prof_on) return SGAM_EINVAL;
    sgam_i_prof_begin("(empty)", "", sgam_stream(stream));
    sgam_i_prof_end(sgam_stream(stream));
    return SGAM_OK;
}
extern "C" int32_t sgam_prof_count(void) { return g_prof_n; }
extern "C" int sgam_prof_get_shape(int32_t i, int32_t *mnks) {   // GEMM view of the launch: M, N, K, split-K factor (0: n/a)
    if (i < 0 || i >= g_prof_n || !mnks) return SGAM_EINVAL;
    for (int j = 0; j < 4; ++j) mnks[j] = g_prof[i].shape[j];
    return SGAM_OK;
}
extern "C" int sgam_prof_get(int32_t i, const char **kernel, const char **where, float *ms, double *flops, double *bytes) {
    if (i < 0 || i >= g_prof_n || !kernel || !where || !ms || !flops || !bytes) return SGAM_EINVAL;
    const ProfRec &r = g_prof[i];
    hipError_t e = hipEventElapsedTime(ms, r.e0, r.e1);
    if (e != hipSuccess) return (int)e;
    *kernel = r.kernel;
    *where = r.where;
    *flops = r.flops;
    *bytes = r.bytes;
    return SGAM_OK;
}

// (sgam_abi_version / sgam_build_info: build_info.hip)

extern "C" int sgam_nchw_to_nhwc_f32(const float *x, float *y, int32_t B, int32_t C, int32_t HW, int32_t ldy,
                                     void *stream) {
    if (!x || !y || B <= 0 || C <= 0 || HW <= 0 || ldy < C) return SGAM_EINVAL;
    SGAM_KLAUNCH(nchw_to_nhwc_kernel, dim3(sgam_cdiv(HW, 32), sgam_cdiv(C, 32), B), dim3(256), 0,
                       sgam_stream(stream), x, y, C, HW, ldy);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_nhwc_to_nchw_f32(const float *x, float *y, int32_t B, int32_t C, int32_t HW, int32_t ldx,
                                     void *stream) {
    if (!x || !y || B <= 0 || C <= 0 || HW <= 0 || ldx < C) return SGAM_EINVAL;
    SGAM_KLAUNCH(nhwc_to_nchw_kernel, dim3(sgam_cdiv(HW, 32), sgam_cdiv(C, 32), B), dim3(256), 0,
                       sgam_stream(stream), x, y, C, HW, ldx);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_encode_head_f32(const float *x, const uint8_t *mask, const float *w, const float *bias, float *y,
                                    int32_t B, int32_t HW, int32_t ldy, void *stream) {
    if (!x || !w || !bias || !y || B <= 0 || HW <= 0 || ldy < 4 || ldy % 4 != 0) return SGAM_EINVAL;
    if (!sgam_aligned16(y)) return SGAM_EALIGN;
    SGAM_KLAUNCH(encode_head_kernel, dim3(sgam_cdiv(HW, 256), B), dim3(256), 0, sgam_stream(stream), x, mask, w,
                       bias, y, HW, ldy);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_frame_feedback_f32(const float *dec, const float *lut256, int32_t dataset_norm, uint8_t *rgb_u8,
                                       float *rgb_f, float *depth, int32_t B, int32_t HW, void *stream) {
    if (!dec || !lut256 || B <= 0 || HW <= 0 || (dataset_norm != 1 && dataset_norm != 2)) return SGAM_EINVAL;
    SGAM_KLAUNCH(frame_feedback_kernel, dim3(sgam_cdiv(HW, 256), B), dim3(256), 0, sgam_stream(stream), dec,
                       lut256, dataset_norm, rgb_u8, rgb_f, depth, HW);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}
