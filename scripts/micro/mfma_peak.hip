// MFMA issue-rate calibration: independent / dependent accumulator chains of v_mfma_f32_32x32x16_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    f16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(seed * (threadIdx.x % 7 + e)); y[e] = (_Float16)(seed * (threadIdx.x % 5 + e)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 24 / NACC; ++r)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks, int iters, float *d, float seed, const char *tag) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, seed);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)blocks * 4 * iters * 24;
    double tf = mf * 32768.0 / (ms * 1e-3) / 1e12;
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 24 * (blocks * 4 / 1024.0));
    printf("%-28s blocks=%5d nacc=%d: %8.3f ms  %7.1f TF/s  (%.1f cyc@2.4GHz per MFMA per SIMD)\n", tag, blocks, NACC, ms, tf, cyc);
}

int main() {
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<4>(256, 20000, d, 0.f, "zeros 1 wave/SIMD");
        run<4>(512, 20000, d, 0.f, "zeros 2 waves/SIMD");
        run<4>(256, 20000, d, 0.37f, "data 1 wave/SIMD");
        run<4>(512, 20000, d, 0.37f, "data 2 waves/SIMD");
        run<1>(512, 20000, d, 0.37f, "data dependent chain 2w");
        run<2>(512, 20000, d, 0.37f, "data 2 accs 2w");
        run<8>(512, 20000, d, 0.37f, "data 8 accs 2w");
        run<4>(1024, 20000, d, 0.37f, "data 4 waves/SIMD");
    }
    return 0;
}
