// warp.hip — the two conditioning warps of SGAM as exact-semantics gather/scatter kernels.
//
//   forward splat : render_projection_from_srcs_fast, sgam/point_rendering/warp.py:193-286
//   inverse warp  : InfiniteSceneGeneration.inverse_warping, sgam/inference_pipeline.py:662-743
//
// These are HBM / latency-bound integer-and-compare kernels (no MFMA): ~(16N+17) bytes per target
// pixel.  Every float expression is written with explicit round-to-nearest intrinsics in the order the
// reference's torch-CPU path evaluates it (3-term dot products are fma chains a0*b0 -> fma(a1,b1) ->
// fma(a2,b2), as oneMKL's sgemm does for 3x3 @ 3xHW; see oracle/warp_oracle.c) so pixel indices,
// depths and masks are bit-identical to the reference, and this file is built with -ffp-contract=off.
//
// The reference's index_put_ scatter is racy (last writer wins); its deterministic meaning is "the
// LARGEST linear point index p = pixel*N + src wins" (the sequential parallel=False loop).  Here:
//   pass 1  splat_winner_kernel : one lane per source point -> atomicMax(winner[target pixel], p)
//           (integer max is order-independent => deterministic by construction)
//   pass 2  splat_resolve_kernel: 16x16 target tile + 1-pixel halo staged in LDS: each halo cell
//           re-derives (r,g,b,z) of its winning point; then per pixel 4 exact 3x3 medians
//           (19-exchange network), the per-channel `== 0` hole merge, the extrapolation mask and the
//           inverse-depth normalisation of VQModel.get_x (model.py:210-229), fused.
#include "sgam_common.h"

namespace {

__device__ __forceinline__ float dot3_fma(const float *a, float b0, float b1, float b2) {
    float acc = __fmul_rn(a[0], b0);
    acc = __fmaf_rn(a[1], b1, acc);
    acc = __fmaf_rn(a[2], b2, acc);
    return acc;
}
__device__ __forceinline__ float dot3_plain(const float *a, float b0, float b1, float b2) {
    return __fadd_rn(__fadd_rn(__fmul_rn(a[0], b0), __fmul_rn(a[1], b1)), __fmul_rn(a[2], b2));
}

struct Cam {
    float Kinv[9], T[12], Kt[9];
};

// Sources addressed through a by-value table of device pointers (one per (batch item, source)): the scene loop keeps
// every generated frame as its own HBM allocation, and the warps read them in place — no stacked copy per step.
constexpr int SGAM_MAX_SRCS = 64;   // 8 lock-stepped scenes x 5 sources (CLEVR) fit; 1 KB of kernel arguments
struct SrcTable {
    const float *feat[SGAM_MAX_SRCS];
    const float *depth[SGAM_MAX_SRCS];
};

__device__ __forceinline__ void load_cam(Cam &c, const float *Kinv, const float *T, const float *Kt) {
#pragma unroll
    for (int i = 0; i < 9; ++i) c.Kinv[i] = Kinv[i];
#pragma unroll
    for (int i = 0; i < 12; ++i) c.T[i] = T[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) c.Kt[i] = Kt[i];
}

// pixel2cam (warp.py:28-40) then bmm(R, .) + t (warp.py:215): target-camera-frame point.
__device__ __forceinline__ void to_target_cam(const Cam &c, float j, float i, float depth, float &X, float &Y,
                                              float &Z) {
    const float cx = __fmul_rn(dot3_fma(c.Kinv + 0, j, i, 1.0f), depth);
    const float cy = __fmul_rn(dot3_fma(c.Kinv + 3, j, i, 1.0f), depth);
    const float cz = __fmul_rn(dot3_fma(c.Kinv + 6, j, i, 1.0f), depth);
    X = __fadd_rn(dot3_fma(c.T + 0, cx, cy, cz), c.T[3]);
    Y = __fadd_rn(dot3_fma(c.T + 4, cx, cy, cz), c.T[7]);
    Z = __fadd_rn(dot3_fma(c.T + 8, cx, cy, cz), c.T[11]);
}

// (pix2d + 0.5).long() with the bounds mask of warp.py:225-232.  Truncation toward zero; NaN and
// out-of-range values (INT64_MIN on the reference's x86 path) are out of bounds.
__device__ __forceinline__ bool project_pixel(const Cam &c, float X, float Y, float Z, int H, int W, int &px, int &py) {
    const float u = dot3_fma(c.Kt + 0, X, Y, Z);
    const float v = dot3_fma(c.Kt + 3, X, Y, Z);
    const float w = dot3_fma(c.Kt + 6, X, Y, Z);
    const float fx = __fadd_rn(__fdiv_rn(u, w), 0.5f);
    const float fy = __fadd_rn(__fdiv_rn(v, w), 0.5f);
    const bool inb = (fx > -1.0f) && (fx < (float)W) && (fy > -1.0f) && (fy < (float)H);
    if (!inb) return false;
    px = (int)fx;
    py = (int)fy;
    return px >= 0 && px < W && py >= 0 && py < H;
}

template <bool TAB>
__global__ __launch_bounds__(256) void splat_winner_kernel(const float *__restrict__ src_depths, SrcTable tab,
                                                           const float *__restrict__ tgt_K,
                                                           const float *__restrict__ src_Kinv, const float *__restrict__ T,
                                                           int N, int H, int W, int *__restrict__ winner,
                                                           uint8_t *__restrict__ inb_mask, int *__restrict__ pix_xy) {
    const int b = blockIdx.y;
    const int HW = H * W;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;  // q = s*HW + pix: consecutive lanes -> consecutive pixels
    if (q >= HW * N) return;
    const int s = q / HW;
    const int pix = q - s * HW;
    const int bn = b * N + s;
    Cam c;
    load_cam(c, src_Kinv + 9 * bn, T + 16 * bn, tgt_K + 9 * b);
    const int i = pix / W, j = pix - i * W;
    float X, Y, Z;
    const float sd = TAB ? tab.depth[bn][pix] : src_depths[(int64_t)bn * HW + pix];
    to_target_cam(c, (float)j, (float)i, sd, X, Y, Z);
    int px = 0, py = 0;
    const bool inb = project_pixel(c, X, Y, Z, H, W, px, py);
    const int p = pix * N + s;  // the reference's linear point index (warp.py:217-218)
    if (inb) atomicMax(&winner[(int64_t)b * HW + py * W + px], p);
    if (inb_mask) inb_mask[(int64_t)b * HW * N + p] = inb ? 1 : 0;
    if (pix_xy) {
        pix_xy[((int64_t)b * HW * N + p) * 2 + 0] = px;
        pix_xy[((int64_t)b * HW * N + p) * 2 + 1] = py;
    }
}

#define SGAM_S2(a, b)        \
    {                        \
        if ((a) > (b)) {     \
            const float t_ = (a); \
            (a) = (b);       \
            (b) = t_;        \
        }                    \
    }
// exact lower median of 9 (torch.median -> sorted[4]); NaN propagates.  19-exchange network.
__device__ __forceinline__ float median9(float *p) {
    bool nan = false;
#pragma unroll
    for (int i = 0; i < 9; ++i) nan |= (p[i] != p[i]);
    SGAM_S2(p[1], p[2]); SGAM_S2(p[4], p[5]); SGAM_S2(p[7], p[8]);
    SGAM_S2(p[0], p[1]); SGAM_S2(p[3], p[4]); SGAM_S2(p[6], p[7]);
    SGAM_S2(p[1], p[2]); SGAM_S2(p[4], p[5]); SGAM_S2(p[7], p[8]);
    SGAM_S2(p[0], p[3]); SGAM_S2(p[5], p[8]); SGAM_S2(p[4], p[7]);
    SGAM_S2(p[3], p[6]); SGAM_S2(p[1], p[4]); SGAM_S2(p[2], p[5]);
    SGAM_S2(p[4], p[7]); SGAM_S2(p[4], p[2]); SGAM_S2(p[6], p[4]);
    SGAM_S2(p[4], p[2]);
    return nan ? NAN : p[4];
}

constexpr int TS = 16;        // target tile edge
constexpr int TH = TS + 2;    // with halo

__device__ __forceinline__ float normalise_depth(float md, bool hole, int dataset_norm) {
    float wd;
    if (dataset_norm == 1) {  // google_earth, model.py:215-219
        wd = __fdiv_rn(1.0f, __fadd_rn(md, 10.0f));
        wd = __fdiv_rn(__fsub_rn(wd, (float)(1.0 / 14.765625)), (float)(1.0 / 10.099975586 - 1.0 / 14.765625));
    } else {  // clevr-infinite, model.py:225-229
        const float lo = (float)1e-7;
        const float cl = (md != md) ? md : fmaxf(md, lo);
        wd = __fdiv_rn(1.0f, cl);
        wd = __fdiv_rn(__fsub_rn(wd, (float)(1.0 / 16)), (float)(1.0 / 7 - 1.0 / 16));
    }
    wd = __fsub_rn(__fmul_rn(2.0f, wd), 1.0f);
    // warped_depth * ~mask + ones * (-2) * mask
    const float m = hole ? 1.0f : 0.0f, nm = hole ? 0.0f : 1.0f;
    return __fadd_rn(__fmul_rn(wd, nm), __fmul_rn(-2.0f, m));
}

template <bool TAB>
__global__ __launch_bounds__(TS *TS) void splat_resolve_kernel(
    const float *__restrict__ src_feats, int64_t feat_cs, int64_t feat_ps, const float *__restrict__ src_depths, SrcTable tab,
    const float *__restrict__ src_Kinv, const float *__restrict__ T, const int *__restrict__ winner, int N, int H, int W,
    float r0, float r1, int use_range, int dataset_norm, float *__restrict__ merge_depths,
    float *__restrict__ merge_feats, uint8_t *__restrict__ extrap, float *__restrict__ x_out,
    float *__restrict__ proj_feats, float *__restrict__ proj_depth) {
    __shared__ float tile[4][TH][TH + 1];
    const int b = blockIdx.z;
    const int HW = H * W;
    const int ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS;
    for (int h = threadIdx.x; h < TH * TH; h += TS * TS) {
        const int hy = h / TH, hx = h - hy * TH;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, z = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const int wv = winner[(int64_t)b * HW + gy * W + gx];
            if (wv >= 0) {
                const int pix = wv / N, s = wv - pix * N;
                const int bn = b * N + s;
                const float *fp = (TAB ? tab.feat[bn] : src_feats + (int64_t)bn * 3 * HW) + (int64_t)pix * feat_ps;
                f0 = fp[0];
                f1 = fp[feat_cs];
                f2 = fp[2 * feat_cs];
                Cam c;
                load_cam(c, src_Kinv + 9 * bn, T + 16 * bn, src_Kinv);  // Kt unused here
                const int i = pix / W, j = pix - i * W;
                float X, Y;
                to_target_cam(c, (float)j, (float)i, TAB ? tab.depth[bn][pix] : src_depths[(int64_t)bn * HW + pix], X, Y, z);
            }
        }
        tile[0][hy][hx] = f0;
        tile[1][hy][hx] = f1;
        tile[2][hy][hx] = f2;
        tile[3][hy][hx] = z;
    }
    __syncthreads();
    const int ly = threadIdx.x / TS, lx = threadIdx.x - ly * TS;
    const int gy = ty0 + ly, gx = tx0 + lx;
    if (gy >= H || gx >= W) return;
    const int o = gy * W + gx;
    float merged[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) v[dy * 3 + dx] = tile[c][ly + dy][lx + dx];
        const float centre = v[4];
        const float med = median9(v);
        // mask * median + (~mask) * plane, in fp32 like warp.py:277-278 (NaN/Inf propagate the same way)
        const float mk = (centre == 0.0f) ? 1.0f : 0.0f, nmk = (centre == 0.0f) ? 0.0f : 1.0f;
        merged[c] = __fadd_rn(__fmul_rn(mk, med), __fmul_rn(nmk, centre));
        if (c < 3) {
            if (proj_feats) proj_feats[((int64_t)b * 3 + c) * HW + o] = centre;
        } else if (proj_depth) {
            proj_depth[(int64_t)b * HW + o] = centre;
        }
    }
    const float md = merged[3];
    bool hole;
    if (use_range) {  // training branch, warp.py:280-283
        const float le = (md <= r1) ? 1.0f : 0.0f, ge = (md >= r0) ? 1.0f : 0.0f;
        hole = __fsub_rn(1.0f, __fmul_rn(le, ge)) != 0.0f;
        if (md >= r1) merged[0] = merged[1] = merged[2] = 0.0f;
    } else {
        hole = md <= 0.0f;  // warp.py:285
    }
    if (merge_depths) merge_depths[(int64_t)b * HW + o] = md;
    if (extrap) extrap[(int64_t)b * HW + o] = hole ? 1 : 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (merge_feats) merge_feats[((int64_t)b * 3 + c) * HW + o] = merged[c];
        if (x_out) x_out[((int64_t)b * 4 + c) * HW + o] = merged[c];
    }
    if (x_out) x_out[((int64_t)b * 4 + 3) * HW + o] = normalise_depth(md, hole, dataset_norm);
}


// ------------------------------------------------------------------------------------------------
// Forward splat, target-owned tiles (round 4): no global atomics.
//
// The two-pass form above sends one device-scope atomicMax per source point to a [B][HW] winner buffer.  On MI355X those
// are executed at the memory side (the eight XCD L2s are not coherent with each other): measured 65 G atomics/s for a
// coherent scatter — 12.6 M points (512 x 512, B = 16, N = 3) take 190 us + the buffer's memset, against 22 us for the same
// scatter as plain stores and 37 us for the design below (scripts/micro/atomic_splat.hip); issuing them at workgroup scope
// from the XCD that owns a band of the buffer changes nothing (same instruction, same place of execution).  So:
//   pass 1  splat_project_kernel: source points in bins of 8 rows x 32 pixels (whole 128-byte lines of the depth map), one
//           wavefront per bin: target pixel of every point -> tgt[b][s][pix] = (py << 16 | px), -1 when out of bounds; and
//           the bin sets its bit in the bitmap of every target tile its bounding box meets (fire-and-forget atomicOr: a
//           handful per 256 points).  4 N bytes read + 4 N written per pixel.
//   pass 2  splat_tile_kernel<TT>: a workgroup OWNS a TT x TT tile of the target image plus its 1-pixel halo.  It reads (and
//           clears) its bitmap, and for every registered bin reads the bin's cached target pixels and resolves "largest linear
//           point index p = pix * N + s wins" (warp.py:217-218 and the oracle's sequential loop) with LDS atomicMax — integer
//           max: order-independent, deterministic — on its own (TT + 2)^2 z-tile.  Exact for ANY geometry (the boxes are
//           exact, not estimates): a wild warp only makes more bins register with a tile.  Then, still in LDS: (r, g, b, z)
//           of every winner, the four 3 x 3 medians, the `== 0` merge, the mask and the inverse-depth normalisation, outputs
//           written once (the arithmetic is splat_resolve_kernel's).
// HBM bytes per target pixel: 12 N + 33 against the algorithmic 16 N + 17 — the cached target pixel replaces
// the re-read of x / y / z planes the reference materialises.
// ------------------------------------------------------------------------------------------------
constexpr int SBW = 32, SBH = 8;      // source bin: 8 rows x 32 pixels = 256 points, one per thread
typedef short s16x2 __attribute__((ext_vector_type(2)));

// pass 1.  A WAVEFRONT owns a bin (lane = one column, four rows: four independent projections in flight per lane, the
// bin's bounding box by a wave reduction — no LDS, no barrier).  Besides the cached target pixels, every bin REGISTERS itself with
// the target tiles its bounding box (grown by the 1-pixel halo) meets: one bit per (source, bin) in the tile's bitmap, set with an
// atomicOr WITHOUT return value — a handful of fire-and-forget atomics per 256 points (a first version appended to per-tile lists
// through atomicAdd and waited a device-scope round trip per bin: 88 us for 12.6 M points against 60 without registration).
// The bitmaps are zero on entry; pass 2 clears what it reads.
template <bool TAB>
__global__ __launch_bounds__(256) void splat_project_kernel(const float *__restrict__ src_depths, SrcTable tab,
                                                            const float *__restrict__ tgt_K, const float *__restrict__ src_Kinv,
                                                            const float *__restrict__ T, int N, int H, int W, int bins_x, int nbins,
                                                            int tt, int tiles_x, int tiles, int words, int *__restrict__ tgt,
                                                            unsigned *__restrict__ tile_bits) {
    const int lane = threadIdx.x & 63;
    const int bin = blockIdx.x * 4 + (threadIdx.x >> 6), s = blockIdx.y, b = blockIdx.z;
    if (bin >= nbins) return;
    const int bn = b * N + s, HW = H * W;
    const int by = bin / bins_x, bxx = bin - by * bins_x;
    const int j = bxx * SBW + (lane & 31), i0 = by * SBH + (lane >> 5) * 4;
    Cam c;
    load_cam(c, src_Kinv + 9 * bn, T + 16 * bn, tgt_K + 9 * b);
    float sd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool valid = (i0 + r) < H && j < W;
        const int pix = (i0 + r) * W + j;
        sd[r] = valid ? (TAB ? tab.depth[bn][pix] : src_depths[(int64_t)bn * HW + pix]) : 0.f;
    }
    s16x2 lo = {32767, 32767}, hi = {-1, -1};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool valid = (i0 + r) < H && j < W;
        float X, Y, Z;
        to_target_cam(c, (float)j, (float)(i0 + r), sd[r], X, Y, Z);
        int px = 0, py = 0;
        const bool inb = valid && project_pixel(c, X, Y, Z, H, W, px, py);
        if (valid) tgt[(int64_t)bn * HW + (i0 + r) * W + j] = inb ? ((py << 16) | px) : -1;
        if (inb) {
            lo = __builtin_elementwise_min(lo, s16x2{(short)px, (short)py});
            hi = __builtin_elementwise_max(hi, s16x2{(short)px, (short)py});
        }
    }
    // bounding box of the bin's target pixels: packed 16-bit (x, y) pairs, min and max by v_pk_min_i16 / v_pk_max_i16
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo = __builtin_elementwise_min(lo, __builtin_bit_cast(s16x2, __shfl_xor(__builtin_bit_cast(int, lo), o, 64)));
        hi = __builtin_elementwise_max(hi, __builtin_bit_cast(s16x2, __shfl_xor(__builtin_bit_cast(int, hi), o, 64)));
    }
    if (lo[0] > hi[0]) return;                       // no point of this bin lands in the image
    // tiles met by the box grown by one pixel (a tile also needs the winners of its halo)
    const int tx0 = max(lo[0] - 1, 0) / tt, tx1 = min(hi[0] + 1, W - 1) / tt, ty0 = max(lo[1] - 1, 0) / tt, ty1 = min(hi[1] + 1, H - 1) / tt;
    const int nx = tx1 - tx0 + 1, cnt = nx * (ty1 - ty0 + 1);
    const int e = s * nbins + bin;
    for (int k = lane; k < cnt; k += 64) {
        const int tile = (ty0 + k / nx) * tiles_x + tx0 + k % nx;
        __hip_atomic_fetch_or(&tile_bits[((int64_t)b * tiles + tile) * words + (e >> 5)], 1u << (e & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// pass 2: one workgroup per TT x TT target tile (+ 1-pixel halo): LDS z-tile over the registered bins, then the resolve
template <int TT, bool TAB>
__global__ __launch_bounds__(256) void splat_tile_kernel(
    const float *__restrict__ src_feats, int64_t feat_cs, int64_t feat_ps, const float *__restrict__ src_depths, SrcTable tab,
    const float *__restrict__ src_Kinv, const float *__restrict__ T, const int *__restrict__ tgt, unsigned *__restrict__ tile_bits,
    int words, int N, int H, int W, int bins_x, int nbins, float r0, float r1, int use_range, int dataset_norm,
    float *__restrict__ merge_depths, float *__restrict__ merge_feats, uint8_t *__restrict__ extrap, float *__restrict__ x_out,
    float *__restrict__ proj_feats, float *__restrict__ proj_depth) {
    constexpr int TE = TT + 2;                       // tile edge with the 1-pixel halo
    constexpr int CPT = (TE * TE + 255) / 256;       // halo-tile cells per thread
    __shared__ int win[TE * TE];
    __shared__ float tile[4][TE][TE + 1];
    __shared__ float camz[SGAM_MAX_SRCS][13];        // per source: Kinv (9) and row 2 of T (4): what a winner's camera-z needs
    constexpr int CHW = 64;                          // bitmap words per round: at most 2048 registered bins in the list
    __shared__ int list[CHW * 32];
    __shared__ int cnt;
    const int b = blockIdx.z, HW = H * W;
    const int ty0 = blockIdx.y * TT, tx0 = blockIdx.x * TT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = gridDim.x * gridDim.y, tile_id = blockIdx.y * gridDim.x + blockIdx.x;
    for (int c = tid; c < TE * TE; c += 256) win[c] = -1;
    for (int c = tid; c < N * 13; c += 256) {
        const int s = c / 13, k = c - s * 13, bn = b * N + s;
        camz[s][k] = k < 9 ? src_Kinv[9 * bn + k] : T[16 * bn + 8 + (k - 9)];
    }
    unsigned *bits = tile_bits + ((int64_t)b * tiles + tile_id) * words;
    const int *tg = tgt + (int64_t)b * N * HW;
    const int col = lane & 31, row0 = (lane >> 5) * 4;
    for (int wbase = 0; wbase < words; wbase += CHW) {
        if (tid == 0) cnt = 0;
        __syncthreads();                             // (also: win[] / camz[] initialised, the previous round's list consumed)
        if (tid < CHW && wbase + tid < words) {
            unsigned m = bits[wbase + tid];
            if (m) {
                bits[wbase + tid] = 0u;              // ready for the next call (this tile's words are nobody else's now)
                int at = atomicAdd(&cnt, __popc(m));
                while (m) {
                    const int bit = __ffs(m) - 1;
                    m &= m - 1;
                    list[at++] = (wbase + tid) * 32 + bit;
                }
            }
        }
        __syncthreads();
        const int n = cnt;
        // wavefront w takes the registered bins w, w + 4, ...; a lane owns one column of the bin and four of its rows: the four
        // cached target pixels are requested together, two bins per trip (eight loads in flight per lane)
        for (int k = wave; k < n; k += 8) {
            int e[2], t[2][4], pixs[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) e[u] = (k + 4 * u) < n ? list[k + 4 * u] : -1;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int s = e[u] >= 0 ? e[u] / nbins : 0, bin = e[u] >= 0 ? e[u] - s * nbins : 0;
                const int by = bin / bins_x, bxx = bin - by * bins_x;
                const int j = bxx * SBW + col;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = by * SBH + row0 + r;
                    const bool ok = e[u] >= 0 && i < H && j < W;
                    pixs[u][r] = i * W + j;
                    t[u][r] = ok ? tg[(int64_t)s * HW + pixs[u][r]] : -1;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int s = e[u] >= 0 ? e[u] / nbins : 0;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (t[u][r] >= 0) {
                        const int lx = (t[u][r] & 0xFFFF) - (tx0 - 1), ly = (t[u][r] >> 16) - (ty0 - 1);
                        if ((unsigned)lx < (unsigned)TE && (unsigned)ly < (unsigned)TE) atomicMax(&win[ly * TE + lx], pixs[u][r] * N + s);
                    }
            }
        }
        __syncthreads();                             // every wavefront has read this round's count and list
    }
    __syncthreads();
    // (r, g, b, z) of every cell's winner; cells outside the image and empty cells are zero.  All of a thread's gathers are
    // requested before the first is used.
    {
        int wv[CPT];
        float f0[CPT], f1[CPT], f2[CPT], sd[CPT];
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int c = tid + 256 * q;
            const int hy = c / TE, hx = c - hy * TE;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            wv[q] = (c < TE * TE && gy >= 0 && gy < H && gx >= 0 && gx < W) ? win[c] : -1;
            f0[q] = f1[q] = f2[q] = sd[q] = 0.f;
            if (wv[q] >= 0) {
                const int pix = wv[q] / N, s = wv[q] - pix * N;
                const int bn = b * N + s;
                const float *fp = (TAB ? tab.feat[bn] : src_feats + (int64_t)bn * 3 * HW) + (int64_t)pix * feat_ps;
                f0[q] = fp[0];
                f1[q] = fp[feat_cs];
                f2[q] = fp[2 * feat_cs];
                sd[q] = TAB ? tab.depth[bn][pix] : src_depths[(int64_t)bn * HW + pix];
            }
        }
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int c = tid + 256 * q;
            if (c >= TE * TE) continue;
            const int hy = c / TE, hx = c - hy * TE;
            float z = 0.f;
            if (wv[q] >= 0) {
                const int pix = wv[q] / N, s = wv[q] - pix * N;
                const int i = pix / W, j = pix - i * W;
                // to_target_cam's Z, same expressions: cz.. = (Kinv row . (j, i, 1)) * depth, Z = (T row 2 . c) + T[2][3]
                const float *cz = camz[s];
                const float cx = __fmul_rn(dot3_fma(cz + 0, (float)j, (float)i, 1.0f), sd[q]);
                const float cy = __fmul_rn(dot3_fma(cz + 3, (float)j, (float)i, 1.0f), sd[q]);
                const float cc = __fmul_rn(dot3_fma(cz + 6, (float)j, (float)i, 1.0f), sd[q]);
                z = __fadd_rn(dot3_fma(cz + 9, cx, cy, cc), cz[12]);
            }
            tile[0][hy][hx] = f0[q];
            tile[1][hy][hx] = f1[q];
            tile[2][hy][hx] = f2[q];
            tile[3][hy][hx] = z;
        }
    }
    __syncthreads();
    for (int q = tid; q < TT * TT; q += 256) {
        const int ly = q / TT, lx = q - ly * TT;
        const int gy = ty0 + ly, gx = tx0 + lx;
        if (gy >= H || gx >= W) continue;
        const int o = gy * W + gx;
        float merged[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) v[dy * 3 + dx] = tile[c][ly + dy][lx + dx];
            const float centre = v[4];
            const float med = median9(v);
            // mask * median + (~mask) * plane, in fp32 like warp.py:277-278 (NaN/Inf propagate the same way)
            const float mk = (centre == 0.0f) ? 1.0f : 0.0f, nmk = (centre == 0.0f) ? 0.0f : 1.0f;
            merged[c] = __fadd_rn(__fmul_rn(mk, med), __fmul_rn(nmk, centre));
            if (c < 3) {
                if (proj_feats) proj_feats[((int64_t)b * 3 + c) * HW + o] = centre;
            } else if (proj_depth) {
                proj_depth[(int64_t)b * HW + o] = centre;
            }
        }
        const float md = merged[3];
        bool hole;
        if (use_range) {  // training branch, warp.py:280-283
            const float le = (md <= r1) ? 1.0f : 0.0f, ge = (md >= r0) ? 1.0f : 0.0f;
            hole = __fsub_rn(1.0f, __fmul_rn(le, ge)) != 0.0f;
            if (md >= r1) merged[0] = merged[1] = merged[2] = 0.0f;
        } else {
            hole = md <= 0.0f;  // warp.py:285
        }
        if (merge_depths) merge_depths[(int64_t)b * HW + o] = md;
        if (extrap) extrap[(int64_t)b * HW + o] = hole ? 1 : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (merge_feats) merge_feats[((int64_t)b * 3 + c) * HW + o] = merged[c];
            if (x_out) x_out[((int64_t)b * 4 + c) * HW + o] = merged[c];
        }
        if (x_out) x_out[((int64_t)b * 4 + 3) * HW + o] = normalise_depth(md, hole, dataset_norm);
    }
}

// ------------------------------------------------------------------------------------------------
// inverse_warping: one lane per target pixel, sources visited in order with a running z-buffer on
// |z_reprojected - src_depth(at the TARGET pixel, sic)| (inference_pipeline.py:719-737).
// grid_sample(nearest, zeros, align_corners=False) as torch's vectorised CPU kernel evaluates it:
// ix = (x+1)*(W/2) - 0.5, round-half-even.
// ------------------------------------------------------------------------------------------------
template <bool TAB>
__global__ __launch_bounds__(256) void inverse_warp_kernel(const float *__restrict__ src_imgs, int64_t img_cs, int64_t img_ps,
                                                           const float *__restrict__ src_depths, SrcTable tab,
                                                           const float *__restrict__ tgt_depth,
                                                           const float *__restrict__ src_K,
                                                           const float *__restrict__ tgt_Kinv, const float *__restrict__ T,
                                                           int N, int H, int W, float *__restrict__ warped,
                                                           float *__restrict__ zbuf_out) {
    const int b = blockIdx.y;
    const int HW = H * W;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= HW) return;
    const int i = pix / W, j = pix - i * W;
    const float *Ki = tgt_Kinv + 9 * b;
    const float td = tgt_depth[(int64_t)b * HW + pix];
    const float cx = __fmul_rn(dot3_fma(Ki + 0, (float)j, (float)i, 1.0f), td);
    const float cy = __fmul_rn(dot3_fma(Ki + 3, (float)j, (float)i, 1.0f), td);
    const float cz = __fmul_rn(dot3_fma(Ki + 6, (float)j, (float)i, 1.0f), td);
    float res0 = 0.f, res1 = 0.f, res2 = 0.f, zbuf = 99999.0f;
    for (int s = 0; s < N; ++s) {
        const int bn = b * N + s;
        const float *K = src_K + 9 * bn, *Tm = T + 16 * bn;
        // proj = K @ T[:3]: 3x3 @ 3x4 goes through torch's small-gemm path (plain mul/add, no fma)
        float P[12];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) P[r * 4 + c] = dot3_plain(K + 3 * r, Tm[c], Tm[4 + c], Tm[8 + c]);
        const float r0[3] = {P[0], P[1], P[2]}, r1[3] = {P[4], P[5], P[6]}, r2[3] = {P[8], P[9], P[10]};
        const float X = __fadd_rn(dot3_fma(r0, cx, cy, cz), P[3]);
        const float Y = __fadd_rn(dot3_fma(r1, cx, cy, cz), P[7]);
        const float Z = __fadd_rn(dot3_fma(r2, cx, cy, cz), P[11]);
        const float xn = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, __fdiv_rn(X, Z)), (float)(W - 1)), 1.0f);
        const float yn = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, __fdiv_rn(Y, Z)), (float)(H - 1)), 1.0f);
        const float ix = __fsub_rn(__fmul_rn(__fadd_rn(xn, 1.0f), __fdiv_rn((float)W, 2.0f)), 0.5f);
        const float iy = __fsub_rn(__fmul_rn(__fadd_rn(yn, 1.0f), __fdiv_rn((float)H, 2.0f)), 0.5f);
        const float rx = rintf(ix), ry = rintf(iy);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        if (rx >= 0.0f && rx <= (float)(W - 1) && ry >= 0.0f && ry <= (float)(H - 1)) {
            const float *ip = (TAB ? tab.feat[bn] : src_imgs + (int64_t)bn * 3 * HW) + (int64_t)((int)ry * W + (int)rx) * img_ps;
            s0 = __fadd_rn(ip[0], 2.0f);
            s1 = __fadd_rn(ip[img_cs], 2.0f);
            s2 = __fadd_rn(ip[2 * img_cs], 2.0f);
        }
        const float sdep = TAB ? tab.depth[bn][pix] : src_depths[(int64_t)bn * HW + pix];
        const float diff = fabsf(__fsub_rn(Z, sdep));
        const float sum = __fadd_rn(__fadd_rn(s0, s1), s2);
        const bool mk = (diff < zbuf) && (Z >= 0.0f) && (sum > 0.0f);
        const float fm = mk ? 1.0f : 0.0f, fn = mk ? 0.0f : 1.0f;
        zbuf = __fadd_rn(__fmul_rn(fm, diff), __fmul_rn(fn, zbuf));
        res0 = __fadd_rn(__fmul_rn(__fsub_rn(s0, 2.0f), fm), __fmul_rn(fn, res0));
        res1 = __fadd_rn(__fmul_rn(__fsub_rn(s1, 2.0f), fm), __fmul_rn(fn, res1));
        res2 = __fadd_rn(__fmul_rn(__fsub_rn(s2, 2.0f), fm), __fmul_rn(fn, res2));
    }
    warped[((int64_t)b * 3 + 0) * HW + pix] = res0;
    warped[((int64_t)b * 3 + 1) * HW + pix] = res1;
    warped[((int64_t)b * 3 + 2) * HW + pix] = res2;
    if (zbuf_out) zbuf_out[(int64_t)b * HW + pix] = zbuf;
}

// standalone K12: model.py:196-199 + 210-229 when the warp comes from elsewhere (use_rgbd_integration)
__global__ __launch_bounds__(256) void depth_normalise_kernel(const float *__restrict__ depth, int compute_mask,
                                                              uint8_t *__restrict__ extrap, float *__restrict__ out,
                                                              int dataset_norm, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float d = depth[i];
    const bool hole = compute_mask ? (d <= 0.0f) : false;
    if (extrap && compute_mask) extrap[i] = hole ? 1 : 0;
    if (compute_mask) {
        out[i] = normalise_depth(d, hole, dataset_norm);
    } else {
        // the GT branch (x_scaled_inverse_depth): no clip for clevr (model.py:221), no hole merge
        float wd;
        if (dataset_norm == 1) {
            wd = __fdiv_rn(1.0f, __fadd_rn(d, 10.0f));
            wd = __fdiv_rn(__fsub_rn(wd, (float)(1.0 / 14.765625)), (float)(1.0 / 10.099975586 - 1.0 / 14.765625));
        } else {
            wd = __fdiv_rn(1.0f, d);
            wd = __fdiv_rn(__fsub_rn(wd, (float)(1.0 / 16)), (float)(1.0 / 7 - 1.0 / 16));
        }
        out[i] = __fsub_rn(__fmul_rn(2.0f, wd), 1.0f);
    }
}

}  // namespace

extern "C" int sgam_depth_normalise_f32(const float *depth, int32_t compute_mask, uint8_t *extrap, float *out,
                                        int32_t dataset_norm, int64_t n, void *stream) {
    if (!depth || !out || n <= 0 || (dataset_norm != 1 && dataset_norm != 2)) return SGAM_EINVAL;
    SGAM_KLAUNCH(depth_normalise_kernel, dim3(sgam_cdiv(n, 256)), dim3(256), 0, sgam_stream(stream), depth,
                       compute_mask, extrap, out, dataset_norm, n);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

static int forward_splat_launch(const float *src_feats, const float *src_depths, const SrcTable *tab, int64_t feat_cs,
                                int64_t feat_ps, const float *tgt_K, const float *src_Kinv, const float *T, int32_t B,
                                int32_t N, int32_t H, int32_t W, const float *depth_range, int32_t dataset_norm,
                                int32_t *winner, float *merge_depths, float *merge_feats, uint8_t *extrap, float *x_out,
                                float *proj_feats, float *proj_depth, uint8_t *inb_mask, int32_t *pix_xy, void *stream) {
    if (!tgt_K || !src_Kinv || !T || !winner) return SGAM_EINVAL;
    if (B <= 0 || N <= 0 || H <= 0 || W <= 0 || (int64_t)H * W * N >= (1ll << 31)) return SGAM_EINVAL;
    if (x_out && dataset_norm != 1 && dataset_norm != 2) return SGAM_EINVAL;
    hipStream_t s = sgam_stream(stream);
    const int HW = H * W;
    hipError_t e = hipMemsetAsync(winner, 0xFF, (size_t)B * HW * sizeof(int32_t), s);  // -1 = empty
    if (e != hipSuccess) return (int)e;
    const dim3 g1(sgam_cdiv((int64_t)HW * N, 256), B), g2(sgam_cdiv(W, TS), sgam_cdiv(H, TS), B);
    const float r0 = depth_range ? depth_range[0] : 0.f, r1 = depth_range ? depth_range[1] : 0.f;
    SrcTable none = {};
    if (tab) {
        SGAM_KLAUNCH(splat_winner_kernel<true>, g1, dim3(256), 0, s, src_depths, *tab, tgt_K, src_Kinv, T, N, H, W,
                           winner, inb_mask, pix_xy);
        SGAM_LAUNCH_CHECK();
        SGAM_KLAUNCH(splat_resolve_kernel<true>, g2, dim3(TS * TS), 0, s, src_feats, feat_cs, feat_ps, src_depths, *tab,
                           src_Kinv, T, winner, N, H, W, r0, r1, depth_range ? 1 : 0, dataset_norm, merge_depths, merge_feats,
                           extrap, x_out, proj_feats, proj_depth);
    } else {
        SGAM_KLAUNCH(splat_winner_kernel<false>, g1, dim3(256), 0, s, src_depths, none, tgt_K, src_Kinv, T, N, H, W,
                           winner, inb_mask, pix_xy);
        SGAM_LAUNCH_CHECK();
        SGAM_KLAUNCH(splat_resolve_kernel<false>, g2, dim3(TS * TS), 0, s, src_feats, feat_cs, feat_ps, src_depths, none,
                           src_Kinv, T, winner, N, H, W, r0, r1, depth_range ? 1 : 0, dataset_norm, merge_depths, merge_feats,
                           extrap, x_out, proj_feats, proj_depth);
    }
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

static bool fill_table(SrcTable &tab, const float *const *feat_ptrs, const float *const *depth_ptrs, int n) {
    if (!feat_ptrs || !depth_ptrs || n <= 0 || n > SGAM_MAX_SRCS) return false;
    for (int i = 0; i < SGAM_MAX_SRCS; ++i) {
        tab.feat[i] = i < n ? feat_ptrs[i] : nullptr;
        tab.depth[i] = i < n ? depth_ptrs[i] : nullptr;
        if (i < n && (!tab.feat[i] || !tab.depth[i])) return false;
    }
    return true;
}

extern "C" int sgam_forward_splat_f32(const float *src_feats, int64_t feat_cs, int64_t feat_ps, const float *src_depths,
                                      const float *tgt_K, const float *src_Kinv, const float *T, int32_t B, int32_t N,
                                      int32_t H, int32_t W, const float *depth_range, int32_t dataset_norm,
                                      int32_t *winner, float *merge_depths, float *merge_feats, uint8_t *extrap,
                                      float *x_out, float *proj_feats, float *proj_depth, uint8_t *inb_mask,
                                      int32_t *pix_xy, void *stream) {
    if (!src_feats || !src_depths) return SGAM_EINVAL;
    return forward_splat_launch(src_feats, src_depths, nullptr, feat_cs, feat_ps, tgt_K, src_Kinv, T, B, N, H, W, depth_range,
                                dataset_norm, winner, merge_depths, merge_feats, extrap, x_out, proj_feats, proj_depth,
                                inb_mask, pix_xy, stream);
}

extern "C" int sgam_forward_splat_srcs_f32(const float *const *src_feat_ptrs, const float *const *src_depth_ptrs,
                                           int64_t feat_cs, int64_t feat_ps, const float *tgt_K, const float *src_Kinv,
                                           const float *T, int32_t B, int32_t N, int32_t H, int32_t W,
                                           const float *depth_range, int32_t dataset_norm, int32_t *winner,
                                           float *merge_depths, float *merge_feats, uint8_t *extrap, float *x_out,
                                           float *proj_feats, float *proj_depth, uint8_t *inb_mask, int32_t *pix_xy,
                                           void *stream) {
    SrcTable tab;
    if (B <= 0 || N <= 0 || !fill_table(tab, src_feat_ptrs, src_depth_ptrs, B * N)) return SGAM_EINVAL;
    return forward_splat_launch(nullptr, nullptr, &tab, feat_cs, feat_ps, tgt_K, src_Kinv, T, B, N, H, W, depth_range,
                                dataset_norm, winner, merge_depths, merge_feats, extrap, x_out, proj_feats, proj_depth,
                                inb_mask, pix_xy, stream);
}

// ---- target-owned tiles: workspace = per-tile bitmaps of registered source bins [B][tiles][words] uint32 (ZERO on first use,
// left zero by every call) + tgt [B][N][HW] int32, each part rounded up to 256 bytes
static inline int splat_bins_x(int W) { return (W + SBW - 1) / SBW; }
static inline int splat_bins(int H, int W) { return splat_bins_x(W) * ((H + SBH - 1) / SBH); }
// 32 x 32 tiles when they still give every CU a workgroup, 16 x 16 tiles (4 x the workgroups) below
static inline int splat_tt(int B, int H, int W) { return (int64_t)B * sgam_cdiv(W, 32) * sgam_cdiv(H, 32) >= 256 ? 32 : 16; }
static inline int64_t splat_r256(int64_t v) { return (v + 255) / 256 * 256; }
static inline int splat_words(int N, int H, int W) { return (N * splat_bins(H, W) + 31) / 32; }

extern "C" int64_t sgam_forward_splat_workspace_zero_bytes(int32_t B, int32_t N, int32_t H, int32_t W) {
    if (B <= 0 || N <= 0 || N > SGAM_MAX_SRCS || H <= 0 || W <= 0 || H > 32767 || W > 32767 || (int64_t)H * W * N >= (1ll << 31)) return -1;
    const int tt = splat_tt(B, H, W);
    return splat_r256((int64_t)B * sgam_cdiv(W, tt) * sgam_cdiv(H, tt) * splat_words(N, H, W) * 4);
}

extern "C" int64_t sgam_forward_splat_workspace_bytes(int32_t B, int32_t N, int32_t H, int32_t W) {
    const int64_t z = sgam_forward_splat_workspace_zero_bytes(B, N, H, W);
    return z < 0 ? -1 : z + splat_r256((int64_t)B * N * H * W * 4);
}

static int forward_splat_tiled_launch(const float *src_feats, const float *src_depths, const SrcTable *tab, int64_t feat_cs,
                                      int64_t feat_ps, const float *tgt_K, const float *src_Kinv, const float *T, int32_t B, int32_t N,
                                      int32_t H, int32_t W, const float *depth_range, int32_t dataset_norm, void *workspace,
                                      int64_t workspace_bytes, float *merge_depths, float *merge_feats, uint8_t *extrap, float *x_out,
                                      float *proj_feats, float *proj_depth, void *stream) {
    if (!tgt_K || !src_Kinv || !T || !workspace) return SGAM_EINVAL;
    const int64_t need = sgam_forward_splat_workspace_bytes(B, N, H, W);
    if (need < 0) return SGAM_EINVAL;
    if (workspace_bytes < need || !sgam_aligned16(workspace)) return SGAM_EWORKSPACE;
    if (x_out && dataset_norm != 1 && dataset_norm != 2) return SGAM_EINVAL;
    hipStream_t s = sgam_stream(stream);
    const int bins_x = splat_bins_x(W), nbins = splat_bins(H, W), words = splat_words(N, H, W);
    const int tt = splat_tt(B, H, W), tiles_x = sgam_cdiv(W, tt), tiles = tiles_x * sgam_cdiv(H, tt);
    unsigned *tile_bits = (unsigned *)workspace;
    int *tgt = (int *)((char *)workspace + sgam_forward_splat_workspace_zero_bytes(B, N, H, W));
    const float r0 = depth_range ? depth_range[0] : 0.f, r1 = depth_range ? depth_range[1] : 0.f;
    const int use_range = depth_range ? 1 : 0;
    SrcTable none = {};
    const dim3 g1(sgam_cdiv(nbins, 4), N, B);
    if (sgam_i_prof_on) sgam_i_prof_work(0.0, (double)B * H * W * (8.0 * N));
    if (tab) SGAM_KLAUNCH(splat_project_kernel<true>, g1, dim3(256), 0, s, src_depths, *tab, tgt_K, src_Kinv, T, N, H, W, bins_x, nbins, tt, tiles_x, tiles, words, tgt, tile_bits);
    else SGAM_KLAUNCH(splat_project_kernel<false>, g1, dim3(256), 0, s, src_depths, none, tgt_K, src_Kinv, T, N, H, W, bins_x, nbins, tt, tiles_x, tiles, words, tgt, tile_bits);
    SGAM_LAUNCH_CHECK();
    if (sgam_i_prof_on) sgam_i_prof_work(0.0, (double)B * H * W * (4.0 * N + 33.0));
#define SPLAT_TILE(TT_, TAB_)                                                                                                      \
    SGAM_KLAUNCH((splat_tile_kernel<TT_, TAB_>), dim3(sgam_cdiv(W, TT_), sgam_cdiv(H, TT_), B), dim3(256), 0, s, src_feats, feat_cs,  \
                 feat_ps, src_depths, TAB_ ? *tab : none, src_Kinv, T, tgt, tile_bits, words, N, H, W, bins_x, nbins, r0, r1,          \
                 use_range, dataset_norm, merge_depths, merge_feats, extrap, x_out, proj_feats, proj_depth)
    if (tt == 32) {
        if (tab) SPLAT_TILE(32, true); else SPLAT_TILE(32, false);
    } else {
        if (tab) SPLAT_TILE(16, true); else SPLAT_TILE(16, false);
    }
#undef SPLAT_TILE
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_forward_splat_tiled_f32(const float *src_feats, int64_t feat_cs, int64_t feat_ps, const float *src_depths,
                                            const float *tgt_K, const float *src_Kinv, const float *T, int32_t B, int32_t N,
                                            int32_t H, int32_t W, const float *depth_range, int32_t dataset_norm, void *workspace,
                                            int64_t workspace_bytes, float *merge_depths, float *merge_feats, uint8_t *extrap,
                                            float *x_out, float *proj_feats, float *proj_depth, void *stream) {
    if (!src_feats || !src_depths) return SGAM_EINVAL;
    return forward_splat_tiled_launch(src_feats, src_depths, nullptr, feat_cs, feat_ps, tgt_K, src_Kinv, T, B, N, H, W, depth_range,
                                      dataset_norm, workspace, workspace_bytes, merge_depths, merge_feats, extrap, x_out, proj_feats,
                                      proj_depth, stream);
}

extern "C" int sgam_forward_splat_tiled_srcs_f32(const float *const *src_feat_ptrs, const float *const *src_depth_ptrs,
                                                 int64_t feat_cs, int64_t feat_ps, const float *tgt_K, const float *src_Kinv,
                                                 const float *T, int32_t B, int32_t N, int32_t H, int32_t W, const float *depth_range,
                                                 int32_t dataset_norm, void *workspace, int64_t workspace_bytes, float *merge_depths,
                                                 float *merge_feats, uint8_t *extrap, float *x_out, float *proj_feats,
                                                 float *proj_depth, void *stream) {
    SrcTable tab;
    if (B <= 0 || N <= 0 || !fill_table(tab, src_feat_ptrs, src_depth_ptrs, B * N)) return SGAM_EINVAL;
    return forward_splat_tiled_launch(nullptr, nullptr, &tab, feat_cs, feat_ps, tgt_K, src_Kinv, T, B, N, H, W, depth_range,
                                      dataset_norm, workspace, workspace_bytes, merge_depths, merge_feats, extrap, x_out, proj_feats,
                                      proj_depth, stream);
}

extern "C" int sgam_inverse_warp_f32(const float *src_imgs, const float *src_depths, const float *tgt_depth,
                                     const float *src_K, const float *tgt_Kinv, const float *T_tgt2src, int32_t B,
                                     int32_t N, int32_t H, int32_t W, float *warped, float *zbuf, void *stream) {
    if (!src_imgs || !src_depths || !tgt_depth || !src_K || !tgt_Kinv || !T_tgt2src || !warped) return SGAM_EINVAL;
    if (B <= 0 || N <= 0 || H <= 1 || W <= 1) return SGAM_EINVAL;
    SrcTable none = {};
    SGAM_KLAUNCH(inverse_warp_kernel<false>, dim3(sgam_cdiv((int64_t)H * W, 256), B), dim3(256), 0, sgam_stream(stream),
                       src_imgs, (int64_t)H * W, (int64_t)1, src_depths, none, tgt_depth, src_K, tgt_Kinv, T_tgt2src, N, H, W,
                       warped, zbuf);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_inverse_warp_srcs_f32(const float *const *src_img_ptrs, const float *const *src_depth_ptrs, int64_t img_cs,
                                          int64_t img_ps, const float *tgt_depth, const float *src_K, const float *tgt_Kinv,
                                          const float *T_tgt2src, int32_t B, int32_t N, int32_t H, int32_t W, float *warped,
                                          float *zbuf, void *stream) {
    SrcTable tab;
    if (B <= 0 || N <= 0 || !fill_table(tab, src_img_ptrs, src_depth_ptrs, B * N)) return SGAM_EINVAL;
    if (!tgt_depth || !src_K || !tgt_Kinv || !T_tgt2src || !warped || H <= 1 || W <= 1) return SGAM_EINVAL;
    SGAM_KLAUNCH(inverse_warp_kernel<true>, dim3(sgam_cdiv((int64_t)H * W, 256), B), dim3(256), 0, sgam_stream(stream),
                       nullptr, img_cs, img_ps, nullptr, tab, tgt_depth, src_K, tgt_Kinv, T_tgt2src, N, H, W, warped, zbuf);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}
