// Do MFMA and VALU instructions of two DIFFERENT wavefronts on the same SIMD overlap on gfx950?
// A workgroup of 512 threads puts two wavefronts on each SIMD of a CU: wavefronts 0-3 issue a stream of independent
// v_mfma_f32_32x32x16_bf16 (four accumulators), wavefronts 4-7 a stream of independent VALU ops (v_fma_f32, or v_exp_f32).
// Timed: MFMA stream alone, VALU stream alone, both together.  Overlap -> both ~ max; mutual exclusion -> both ~ sum.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_valu_overlap.hip -o scripts/micro/mfma_valu_overlap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // bit 0: MFMA wavefronts work, bit 1: VALU wavefronts work; bit 2: VALU stream = transcendental
__global__ __launch_bounds__(512) void k(float *out, int n_mfma, int n_valu) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!(MODE & 1)) return;
        f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
        bf16x8 x, y;
        for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(0.01f * (threadIdx.x + i)); y[i] = (__bf16)(0.02f * (i + 1)); }
        for (int i = 0; i < n_mfma; i += 4) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        if (!(MODE & 2)) return;
        float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f, v4 = v0 + 4.f, v5 = v0 + 5.f, v6 = v0 + 6.f, v7 = v0 + 7.f;
        for (int i = 0; i < n_valu; i += 8) {
            if (MODE & 4) {
                v0 = __builtin_amdgcn_exp2f(v0); v1 = __builtin_amdgcn_exp2f(v1); v2 = __builtin_amdgcn_exp2f(v2); v3 = __builtin_amdgcn_exp2f(v3);
                v4 = __builtin_amdgcn_exp2f(v4); v5 = __builtin_amdgcn_exp2f(v5); v6 = __builtin_amdgcn_exp2f(v6); v7 = __builtin_amdgcn_exp2f(v7);
            } else {
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"
                             "v_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
}

template <int MODE> float run(float *out, int nm, int nv) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 512>>>(out, nm, nv); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<256, 512>>>(out, nm, nv); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
int main() {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    const int NM = 4096;                                   // MFMAs per wavefront: 4096 x 32 cycles = 131 k cycles
    for (int nv : {4096, 16384, 32768}) {
        printf("MFMA/wave %d, VALU/wave %d (v_fma_f32): mfma alone %.1f us, valu alone %.1f us, both %.1f us\n", NM, nv, run<1>(out, NM, nv), run<2>(out, NM, nv), run<3>(out, NM, nv));
    }
    for (int nv : {2048, 8192}) {
        printf("MFMA/wave %d, TRANS/wave %d (v_exp_f32): mfma alone %.1f us, trans alone %.1f us, both %.1f us\n", NM, nv, run<1>(out, NM, nv), run<6>(out, NM, nv), run<7>(out, NM, nv));
    }
    return 0;
}
