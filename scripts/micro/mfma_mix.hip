// What slows an MFMA stream down?  24 x v_mfma_f32_32x32x16_f16 per iteration (4 accumulators), plus optional
// ds_read_b128 / buffer loads / VALU per iteration, 4 waves per block, 1 or 2 blocks per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NLDS, int NGL, int NVALU, bool BAR, int GMASK = 0xFFFFF, int GPAT = 0>
__global__ __launch_bounds__(256, 2) void k(float *out, const u32x4 *g, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short sm[32768];   // 64 KB
    for (int i = threadIdx.x; i < 32768; i += 256) sm[i] = (unsigned short)(i * 7 + 0x3c00);
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    const int lane = threadIdx.x & 63;
    const u32x4 *lp = reinterpret_cast<const u32x4 *>(sm) + lane * 5;     // 80-byte stride rows
    u32x4 af[16], bf[4];
    for (int i = 0; i < 16; ++i) af[i] = lp[(i * 64) & 2047];
    for (int i = 0; i < 4; ++i) bf[i] = lp[(i * 64 + 7) & 2047];
    float v = threadIdx.x * 0.001f;
    for (int it = 0; it < iters; ++it) {
        if (NGL) {
#pragma unroll
            for (int q = 0; q < NGL; ++q) bf[q & 3] ^= g[(GPAT == 0 ? ((it * NGL + q) * 256 + threadIdx.x) : ((it * NGL + q) * 64 + (threadIdx.x & 31) * 8 * 9 + (threadIdx.x >> 5))) & GMASK];
        }
        if (NLDS) {
#pragma unroll
            for (int q = 0; q < NLDS; ++q) af[q & 15] = lp[((it + q) * 64) & 2047];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int a = 0; a < 4; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[(r * 4 + a) & 15]),
                                                               __builtin_bit_cast(f16x8, bf[r & 3]), acc[a], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NVALU; ++q) v = v * 1.0001f + 0.5f;
        if (BAR) __syncthreads();
    }
    float s = v;
    for (int a = 0; a < 4; ++a)
        for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NLDS, int NGL, int NVALU, bool BAR, int GMASK = 0xFFFFF, int GPAT = 0>
void run(int blocks, int iters, float *d, const u32x4 *g, const char *tag) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NLDS, NGL, NVALU, BAR, GMASK, GPAT>), dim3(blocks), dim3(256), 0, 0, d, g, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NLDS, NGL, NVALU, BAR, GMASK, GPAT>), dim3(blocks), dim3(256), 0, 0, d, g, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double tf = (double)blocks * 4 * iters * 24 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-44s blocks=%4d: %8.3f ms  %7.1f TF/s\n", tag, blocks, ms, tf);
}

int main() {
    float *d; hipMalloc(&d, 2048 * 256 * 4);
    u32x4 *g; hipMalloc(&g, (1 << 20) * 16 + 4096); hipMemset(g, 0x11, (1 << 20) * 16);
    const int IT = 4000;
    for (int nb : {256, 512}) {
        run<0, 0, 0, false>(nb, IT, d, g, "mfma only");
        run<0, 8, 0, false, 0x7FFF>(nb, IT, d, g, "+8 global b128, 512 KB footprint, contiguous");
        run<0, 4, 0, false, 0x7FFF>(nb, IT, d, g, "+4 global b128, 512 KB footprint, contiguous");
        run<0, 8, 0, false, 0x7FFF, 1>(nb, IT, d, g, "+8 global b128, 512 KB, 32 rows x 32 B pattern");
        run<0, 4, 0, false, 0x7FFF, 1>(nb, IT, d, g, "+4 global b128, 512 KB, 32 rows x 32 B pattern");
        run<16, 4, 0, false, 0x7FFF>(nb, IT, d, g, "+16 ds_read +4 global contiguous");
        run<16, 4, 32, false, 0x7FFF>(nb, IT, d, g, "+16 ds_read +4 global contiguous +32 valu");
        run<16, 4, 64, true, 0x7FFF>(nb, IT, d, g, "+16 ds_read +4 global +64 valu + barrier");
    }
    return 0;
}
