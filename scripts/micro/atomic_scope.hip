// micro-benchmark: cost of same-address 64-bit integer atomics by scope and by address spread (per-XCC lines)
// hipcc --offload-arch=gfx950 -O3 scripts/micro/atomic_scope.hip -o /tmp/atomic_scope && /tmp/atomic_scope
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}
template <int MODE>   // 0: agent scope, one address; 1: agent scope, per-XCC line; 2: workgroup scope, per-XCC line; 3: agent, 64 replicas by block
__global__ void k(unsigned long long *acc, int *xcc_of_block, int adds) {
    const int x = xcc_id();
    if (threadIdx.x == 0 && xcc_of_block) xcc_of_block[blockIdx.x] = x;
    if (threadIdx.x < 32) {
        unsigned long long *o;
        if (MODE == 0) o = acc + threadIdx.x * 16;
        else if (MODE == 3) o = acc + ((blockIdx.x & 63) * 32 + threadIdx.x) * 16;
        else o = acc + (x * 32 + threadIdx.x) * 16;          // 128-byte line per (xcc, lane)
        for (int i = 0; i < adds; ++i) {
            if (MODE == 2) __hip_atomic_fetch_add(o, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(o, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
int main() {
    unsigned long long *acc; int *xb;
    const int NB = 512, ADDS = 64;
    hipMalloc(&acc, 64 * 32 * 16 * 8); hipMalloc(&xb, NB * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(acc, 0, 64 * 32 * 16 * 8);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) k<0><<<NB, 64>>>(acc, xb, ADDS);
            if (mode == 1) k<1><<<NB, 64>>>(acc, xb, ADDS);
            if (mode == 2) k<2><<<NB, 64>>>(acc, xb, ADDS);
            if (mode == 3) k<3><<<NB, 64>>>(acc, xb, ADDS);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(64 * 32 * 16);
            hipMemcpy(h.data(), acc, h.size() * 8, hipMemcpyDeviceToHost);
            unsigned long long tot = 0; for (auto v : h) tot += v;
            printf("mode %d rep %d: %.1f us, total %llu (expect %d)\n", mode, rep, ms * 1e3, tot, NB * 32 * ADDS);
        }
    }
    std::vector<int> hx(NB); hipMemcpy(hx.data(), xb, NB * 4, hipMemcpyDeviceToHost);
    printf("xcc of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" %d", hx[i]); printf("\n");
    return 0;
}
