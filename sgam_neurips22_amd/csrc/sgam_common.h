// Shared helpers for the gfx950 kernels of libsgam_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sgam_hip.h"

#define SGAM_LAUNCH_CHECK()                          \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

static inline hipStream_t sgam_stream(void *s) { return (hipStream_t)s; }

// ---- in-library kernel timeline (sgam_prof_* in include/sgam_hip.h; implemented in layout.hip) ----
// Every kernel launch of the library goes through SGAM_KLAUNCH.  When profiling is enabled (bench.py's untimed
// roofline pass, never inside a captured graph) the launch is bracketed by two HIP events recorded on the launch
// stream, tagged with the kernel's name as written at the launch site (+ the enclosing function's signature, which
// resolves symbolic template arguments) and the algorithmic work announced through sgam_i_prof_work().
extern "C" int sgam_i_prof_on;
extern "C" void sgam_i_prof_begin(const char *kernel, const char *where, hipStream_t s);
extern "C" void sgam_i_prof_end(hipStream_t s);
extern "C" void sgam_i_prof_work(double flops, double bytes);
extern "C" void sgam_i_prof_shape(int m, int n, int k, int ksplit);
#define SGAM_KLAUNCH(kern, grid, blk, shm, st, ...)                            \
    do {                                                                       \
        if (sgam_i_prof_on) sgam_i_prof_begin(#kern, __PRETTY_FUNCTION__, st); \
        hipLaunchKernelGGL(kern, grid, blk, shm, st, __VA_ARGS__);             \
        if (sgam_i_prof_on) sgam_i_prof_end(st);                               \
    } while (0)
static inline int sgam_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline bool sgam_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- split-fp32 range guard (sgam_f32x_set_range_flag): a device int32 the split-fp32 kernels OR 1 into when a result
// is not finite — which is what an activation beyond fp16's range (|x| >= 65520 -> inf in the hi half) turns into.
extern "C" int32_t *sgam_i_range_flag;
__device__ __forceinline__ bool sgam_not_finite(float t) { return !(__builtin_fabsf(t) <= 3.4028234663852886e38f); }

// ---- GroupNorm statistics as ORDER-INDEPENDENT accumulators.  A producing kernel leaves per-(workgroup, group) partial sums
// {sum, sum of squares}; folding them used to be a launch of its own between producer and consumer (gn_finalize_stats_kernel,
// 33 - 39 per frame) because the chunks must be added in a fixed order for run-to-run identical results.  Integer addition is
// associative: every partial (an fp64 value) is converted ONCE to a 104-bit fixed-point number — 2^-40 resolution, a 64-bit
// high word in units of 2^-8 and a low word of 32 fraction bits — and the two words are added to an accumulator with
// device-scope 64-bit atomics, fire and forget.  Whatever order the workgroups arrive in, the words end up the same.
// Record of a tensor: [B][SGAM_STATS_R replicas][32 groups][4] int64 = {sum hi, sum lo, sumsq hi, sumsq lo}, ZERO before the
// producer runs.  Replicas: same-line atomics are serialised memory-side (~36 ns each, measured: 512 workgroups adding to ONE
// replica made a 26 us kernel take 100); a workgroup adds to replica (workgroup index mod 16) and the consumer adds the sixteen
// replicas — integers again, so still exact and order-free.  |v| is clamped to 2^54 (an fp32 tensor whose group sums leave that
// range has left fp32's useful range too).
#ifndef SGAM_STATS_R
#define SGAM_STATS_R 16
#endif
__device__ __forceinline__ void sgam_stats_split(double v, long long &hi, long long &lo) {
    v = fmin(fmax(v, -0x1p54), 0x1p54);
    const double t = v * 0x1p8, h = floor(t);
    hi = (long long)h;
    lo = (long long)rint((t - h) * 0x1p32);          // t - h is exact, in [0, 1)
}
// add {s, ss} of (image b, group g) to replica `r` (any workgroup-uniform index; reduced mod SGAM_STATS_R here)
__device__ __forceinline__ void sgam_stats_acc_add(long long *acc, int b, unsigned r, int g, double s, double ss) {
    long long *a = acc + (((int64_t)b * SGAM_STATS_R + (r & (SGAM_STATS_R - 1))) * 32 + g) * 4;
    long long h, l;
    sgam_stats_split(s, h, l);
    atomicAdd(reinterpret_cast<unsigned long long *>(a), (unsigned long long)h);
    atomicAdd(reinterpret_cast<unsigned long long *>(a + 1), (unsigned long long)l);
    sgam_stats_split(ss, h, l);
    atomicAdd(reinterpret_cast<unsigned long long *>(a + 2), (unsigned long long)h);
    atomicAdd(reinterpret_cast<unsigned long long *>(a + 3), (unsigned long long)l);
}
// the finished {sum, sumsq} of (image b, group g) for ONE thread: four replicas in flight at a time (a register-light form
// for the fold kernels; the convolution prologues use the workgroup-wide form below)
__device__ __forceinline__ void sgam_stats_acc_get(const long long *acc, int b, int g, double &s, double &ss) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    const long long *a = acc + ((int64_t)b * SGAM_STATS_R * 32 + g) * 4;
    i64x2 su = {0, 0}, sv = {0, 0};
#pragma unroll 1
    for (int r0 = 0; r0 < SGAM_STATS_R; r0 += 4) {
        i64x2 u[4], v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            u[r] = *reinterpret_cast<const i64x2 *>(a + (r0 + r) * 128);
            v[r] = *reinterpret_cast<const i64x2 *>(a + (r0 + r) * 128 + 2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            su += u[r];
            sv += v[r];
        }
    }
    s = (double)su[0] * 0x1p-8 + (double)su[1] * 0x1p-40;
    ss = (double)sv[0] * 0x1p-8 + (double)sv[1] * 0x1p-40;
}
__device__ __forceinline__ void sgam_stats_finish(double s, double ss, double inv_n, float eps, float &mean, float &rstd) {
    const double m = s * inv_n;                      // as gn_finalize_stats_kernel
    double var = ss * inv_n - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}
__device__ __forceinline__ void sgam_stats_acc_mean_rstd(const long long *acc, int b, int g, double inv_n, float eps, float &mean, float &rstd) {
    double s, ss;
    sgam_stats_acc_get(acc, b, g, s, ss);
    sgam_stats_finish(s, ss, inv_n, eps, mean, rstd);
}
// Workgroup-wide form: ALL NT threads call it.  The 16 x 32 (replica, group) records of image b are fetched by NT threads at
// once (ONE memory round trip, 32 bytes per record), added with LDS integer atomics into `scratch` (>= 1 KB of LDS, 16-byte
// aligned, not otherwise in use) and finished by threads 0 .. 31: on return scratch[2 g] = mean, scratch[2 g + 1] = rstd of group
// g, visible to every thread (the function ends with a barrier).
template <int NT>
__device__ __forceinline__ void sgam_stats_acc_block_mean_rstd(const long long *acc, int b, double inv_n, float eps, float *scratch, int tid) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    static_assert((SGAM_STATS_R * 32) % NT == 0 || NT > SGAM_STATS_R * 32, "records split evenly over the threads");
    unsigned long long *sa = reinterpret_cast<unsigned long long *>(scratch);
    for (int i = tid; i < 128; i += NT) sa[i] = 0ull;
    __syncthreads();
    const long long *a = acc + (int64_t)b * SGAM_STATS_R * 128;
    constexpr int PER = (SGAM_STATS_R * 32 + NT - 1) / NT;
    i64x2 u[PER], v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int q = tid + k * NT;                  // q = replica * 32 + group: the record's four words are contiguous
        if (q < SGAM_STATS_R * 32) {
            u[k] = *reinterpret_cast<const i64x2 *>(a + q * 4);
            v[k] = *reinterpret_cast<const i64x2 *>(a + q * 4 + 2);
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int q = tid + k * NT, g = q & 31;
        if (q < SGAM_STATS_R * 32) {
            atomicAdd(&sa[g * 4 + 0], (unsigned long long)u[k][0]);
            atomicAdd(&sa[g * 4 + 1], (unsigned long long)u[k][1]);
            atomicAdd(&sa[g * 4 + 2], (unsigned long long)v[k][0]);
            atomicAdd(&sa[g * 4 + 3], (unsigned long long)v[k][1]);
        }
    }
    __syncthreads();
    float mean = 0.f, rstd = 0.f;
    if (tid < 32) {
        const double s = (double)(long long)sa[tid * 4] * 0x1p-8 + (double)(long long)sa[tid * 4 + 1] * 0x1p-40;
        const double ss = (double)(long long)sa[tid * 4 + 2] * 0x1p-8 + (double)(long long)sa[tid * 4 + 3] * 0x1p-40;
        sgam_stats_finish(s, ss, inv_n, eps, mean, rstd);
    }
    __syncthreads();                                 // the sums have been read: their LDS becomes the {mean, rstd} table
    if (tid < 32) {
        scratch[2 * tid] = mean;
        scratch[2 * tid + 1] = rstd;
    }
    __syncthreads();
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// CDNA wavefront = 64 lanes.
#define SGAM_WAVE 64

__device__ __forceinline__ float sgam_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// fp64 sum over the 64 lanes on the DPP data path (row shifts, then the two row broadcasts of wave64): six dependent
// steps of a few cycles each, where the butterfly of ds_bpermute exchanges costs 24 LDS round trips.  Fixed order:
// inclusive scan inside each row of 16, rows 0+1 / 2+3 joined by row_bcast:15, the halves by row_bcast:31; the total
// (lane 63) is returned to every lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double sgam_dpp_add_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
    const double o = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
    return v + o;            // lanes outside the row mask / without a source read 0.0
}
__device__ __forceinline__ double sgam_wave_sum_f64(double v) {
    v = sgam_dpp_add_f64<0x111, 0xf>(v);      // row_shr:1
    v = sgam_dpp_add_f64<0x112, 0xf>(v);      // row_shr:2
    v = sgam_dpp_add_f64<0x114, 0xf>(v);      // row_shr:4
    v = sgam_dpp_add_f64<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every row holds the row total
    v = sgam_dpp_add_f64<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = sgam_dpp_add_f64<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
    const long long t = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)t, 63), hi = __builtin_amdgcn_readlane((int)(t >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ float sgam_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// swish(x) = x * sigmoid(x) on the hardware transcendentals: sigmoid = rcp(1 + exp2(-x*log2e)).
// v_exp_f32 / v_rcp_f32 are 1-ulp instructions; the result differs from the libm-based expression of the
// reference by a few 1e-7 relative — far inside the 1e-4 fp32 parity budget — at ~1/5 of the VALU cost,
// which is what lets the fused GroupNorm prologue hide behind the MFMA stream.
__device__ __forceinline__ float sgam_swish(float v) {
    const float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}
