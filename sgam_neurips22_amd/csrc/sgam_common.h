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
static inline int sgam_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline bool sgam_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// CDNA wavefront = 64 lanes.
#define SGAM_WAVE 64

__device__ __forceinline__ float sgam_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double sgam_wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
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
