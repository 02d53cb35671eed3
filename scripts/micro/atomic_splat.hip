// micro-benchmark for the forward splat's winner pass: rate of int32 atomicMax over a COHERENT scatter (consecutive lanes hit
// consecutive target cells, +- a small jitter, every cell hit by NSRC lanes of different workgroups), by scope and placement:
//   mode 0  agent-scope atomicMax into one buffer (what splat_winner_kernel does today)
//   mode 1  workgroup-scope atomicMax (performed in the issuing XCD's L2) into the band of the buffer that belongs to the
//           workgroup's XCD (read from HW_REG_XCC_ID): every cell is only ever touched through ONE L2
//   mode 2  plain stores (no atomics): the bandwidth bound of the pass
//   mode 3  agent-scope atomicMax, 64-bit cells
//   mode 4  LDS atomicMax on a tile + plain flush (target-owned tiles: the "LDS z-tile" design), one workgroup per 32x32 tile
// hipcc --offload-arch=gfx950 -O3 scripts/micro/atomic_splat.hip -o /tmp/atomic_splat && /tmp/atomic_splat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}
__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// NPIX cells, NSRC passes over them; point q = s * NPIX + pix targets cell pix + jitter(q) (|jitter| <= 2 rows of W)
template <int MODE>
__global__ __launch_bounds__(256) void k(int *win, long long *win64, int NPIX, int W, int NSRC, int *counter, int *xcc_hist) {
    if (MODE == 1) {
        // XCD-partitioned: band x of the cells belongs to XCD x; a workgroup takes the next 256-point chunk of ITS band
        __shared__ int chunk;
        const int x = xcc_id();
        const int band = NPIX / 8, chunks_per_band = band / 256 * NSRC;
        for (;;) {
            if (threadIdx.x == 0) chunk = atomicAdd(&counter[x * 32], 1);
            __syncthreads();
            const int c = chunk;
            __syncthreads();
            if (c >= chunks_per_band) return;
            const int s = c / (band / 256), pix = x * band + (c % (band / 256)) * 256 + threadIdx.x;
            const unsigned h = hash((unsigned)(s * NPIX + pix));
            int t = pix + (int)(h % 5) - 2 + ((int)((h >> 8) % 3) - 1) * W;
            t = max(x * band, min(x * band + band - 1, t));          // stay inside the band (the real kernel: slow path instead)
            __hip_atomic_fetch_max(&win[t], pix * NSRC + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        const int q = blockIdx.x * 256 + threadIdx.x;
        if (q >= NPIX * NSRC) return;
        const int s = q / NPIX, pix = q - s * NPIX;
        const unsigned h = hash((unsigned)q);
        int t = pix + (int)(h % 5) - 2 + ((int)((h >> 8) % 3) - 1) * W;
        t = max(0, min(NPIX - 1, t));
        if (MODE == 0) atomicMax(&win[t], pix * NSRC + s);
        if (MODE == 2) win[t] = pix * NSRC + s;
        if (MODE == 3) atomicMax((long long *)&win64[t], (long long)(pix * NSRC + s));
    }
}

// mode 4: target-owned 32 x 32 tiles in LDS.  The workgroup of tile T reads the candidate points of a (32 + 2*4)^2 source
// window per source (its own jitter bound: the real kernel needs a conservative window + a fallback), LDS atomicMax, plain flush
__global__ __launch_bounds__(256) void k_lds(int *win, int NPIX, int W, int NSRC) {
    __shared__ int tile[32 * 32];
    const int tiles_x = W / 32, ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    for (int i = threadIdx.x; i < 1024; i += 256) tile[i] = -1;
    __syncthreads();
    const int WIN = 40;
    for (int s = 0; s < NSRC; ++s)
        for (int i = threadIdx.x; i < WIN * WIN; i += 256) {
            const int sy = ty * 32 - 4 + i / WIN, sx = tx * 32 - 4 + i % WIN;
            if (sy < 0 || sx < 0 || sx >= W || sy * W + sx >= NPIX) continue;
            const int pix = sy * W + sx;
            const unsigned h = hash((unsigned)(s * NPIX + pix));
            int t = pix + (int)(h % 5) - 2 + ((int)((h >> 8) % 3) - 1) * W;
            t = max(0, min(NPIX - 1, t));
            const int ly = t / W - ty * 32, lx = t % W - tx * 32;
            if ((unsigned)ly < 32u && (unsigned)lx < 32u) atomicMax(&tile[ly * 32 + lx], pix * NSRC + s);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) win[(ty * 32 + i / 32) * W + tx * 32 + i % 32] = tile[i];
}

int main() {
    const int W = 2048, H = 2048, NPIX = W * H, NSRC = 3;       // 4.2 M cells, 12.6 M points = the `large_512_B16_N3` case
    int *win, *counter, *hist; long long *win64;
    hipMalloc(&win, NPIX * 4); hipMalloc(&win64, (size_t)NPIX * 8); hipMalloc(&counter, 8 * 32 * 4); hipMalloc(&hist, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<int> ref(NPIX), got(NPIX);
    for (int mode = 0; mode <= 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(win, 0xFF, NPIX * 4); hipMemset(win64, 0xFF, (size_t)NPIX * 8); hipMemset(counter, 0, 8 * 32 * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            const int nb = (NPIX * NSRC + 255) / 256;
            if (mode == 0) k<0><<<nb, 256>>>(win, win64, NPIX, W, NSRC, counter, hist);
            if (mode == 1) k<1><<<2048, 256>>>(win, win64, NPIX, W, NSRC, counter, hist);
            if (mode == 2) k<2><<<nb, 256>>>(win, win64, NPIX, W, NSRC, counter, hist);
            if (mode == 3) k<3><<<nb, 256>>>(win, win64, NPIX, W, NSRC, counter, hist);
            if (mode == 4) k_lds<<<(W / 32) * (H / 32), 256>>>(win, NPIX, W, NSRC);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(got.data(), win, NPIX * 4, hipMemcpyDeviceToHost);
            long long sum = 0; for (int v : got) sum += v;
            int bad = -1;
            if (mode == 0 && rep == 0) ref = got;
            if (mode == 4) { bad = 0; for (int i = 0; i < NPIX; ++i) bad += got[i] != ref[i]; }
            if (mode == 1 && rep == 0) {      // host reference of the band-clamped scatter: L2-local atomics must give exactly this
                std::vector<int> want(NPIX, -1);
                const int band = NPIX / 8;
                for (int s2 = 0; s2 < NSRC; ++s2)
                    for (int pix = 0; pix < NPIX; ++pix) {
                        unsigned x = (unsigned)(s2 * NPIX + pix);
                        x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
                        const int xb = pix / band;
                        int t = pix + (int)(x % 5) - 2 + ((int)((x >> 8) % 3) - 1) * W;
                        t = t < xb * band ? xb * band : (t > xb * band + band - 1 ? xb * band + band - 1 : t);
                        if (pix * NSRC + s2 > want[t]) want[t] = pix * NSRC + s2;
                    }
                bad = 0; for (int i = 0; i < NPIX; ++i) bad += got[i] != want[i];
            }
            printf("mode %d rep %d: %8.1f us  %6.1f G points/s  checksum %lld%s\n", mode, rep, ms * 1e3, NPIX * (double)NSRC / ms / 1e6, sum,
                   bad >= 0 ? (bad ? "  MISMATCH vs reference" : "  == reference") : "");
        }
    }
    return 0;
}
