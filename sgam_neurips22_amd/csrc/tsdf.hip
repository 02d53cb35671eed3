// tsdf.hip — truncated-signed-distance fusion of the generated RGB-D frames and a depth render of the fused surface at
// the next target pose: the `rgbd_integration` conditioning branch of the scene loop
// (sgam/inference_pipeline.py:119-133 volume set-up, :745-838 integrate + render).
//
// The reference delegates this to Open3D 0.15.2 (pinned in its requirement.txt; not vendored, not installable here):
// ScalableTSDFVolume(voxel_length, sdf_trunc, RGB8).integrate(...) per source frame, then extract_triangle_mesh() and a
// Filament off-screen render_to_depth_image(z_in_view_space=True).  What is restated here is Open3D's published
// integration rule (ScalableTSDFVolume::Integrate / UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier):
//   * space is cut into volume units of 16^3 voxels, unit (i,j,k) spans [i,i+1) * 16 * voxel_length per axis;
//   * a frame opens every unit within sdf_trunc (per-axis box) of the back-projected depth pixels, sampled with stride 4;
//   * every voxel of an opened unit: centre -> camera; (u, v) = (int)(x fx / z + cx + 0.5), likewise v; d = depth[v][u]
//     (d <= 0: skip); sdf = (d - z) * sqrt(((u-cx)/fx)^2 + ((v-cy)/fy)^2 + 1); if sdf > -sdf_trunc:
//     tsdf <- (tsdf * w + min(1, sdf / sdf_trunc)) / (w + 1), w <- w + 1.
// The depth render is NOT Open3D's (marching-cubes mesh + rasteriser): the fused surface is ray-cast directly — per
// target pixel the ray is marched through the opened units and the first +/- zero crossing of the trilinearly
// interpolated TSDF (nearest-voxel value where a cell has unobserved corners) is refined linearly; the result is the view-space z
// of that point, 0 where nothing is hit, like the reference's depth image with inf -> 0.  Colour (TSDFVolumeColorType::RGB8,
// :123-131, 777-790) is fused with the same running mean per voxel and can be rendered alongside the depth (nearest voxel
// at the hit) — the conditioning path itself only consumes the depth; the colour serves the final coloured export.  Parity for this branch is
// therefore pinned against oracle/tsdf.py (the same rule in numpy) and analytic scenes, not against Open3D: "parity
// unpinned" at the Open3D boundary, as SURVEY.md §8c anticipates.
//
// MI355X-native layout: a direct-mapped unit table (int32 per unit of the scene's bounding box: a few MB even for
// 50 x 50 x 10 scene units — no hashing, no probing) pointing into a pool of 16^3-voxel bricks {tsdf fp32, weight fp32}
// = 32 KB each, allocated by atomic bump from a caller-sized pool (288 GB of HBM: sized once per scene, never resized);
// one workgroup integrates one brick; the brick list of a frame is built on the device (no host round trip, the
// integrate grid is a fixed-size grid-stride loop).  Built with -ffp-contract=off: every expression is evaluated as
// written so that oracle/tsdf.py reproduces the bricks bit for bit (sqrtf, not __fsqrt_rn: the latter is the approximate
// native square root in this toolchain).
#include "sgam_common.h"

#ifndef SGAM_TSDF_ZG
#define SGAM_TSDF_ZG 2        // voxels of a column fetched per group, one group ahead (integrate kernel)
#endif
#ifndef SGAM_TSDF_TOUCH_ABLATE
#define SGAM_TSDF_TOUCH_ABLATE 0
#endif
#ifndef SGAM_TSDF_LB
#define SGAM_TSDF_LB 8        // minimum waves per SIMD the integrate kernel is compiled for (register budget)
#endif

namespace {

constexpr int UR = 16;                 // voxels per unit edge (Open3D volume_unit_resolution)
constexpr int UV = UR * UR * UR;       // voxels per brick
constexpr int NEAR_BIT = 0x40000000;   // unit table entry = brick index | NEAR_BIT once the brick holds part of the band
constexpr int BRICK_MASK = 0x3FFFFFFF;
// (bricks start at 2.0 = unobserved: observed TSDF values are <= 1, so the ray cast tells the two apart in the one load)
constexpr int CS = 32;                 // the four counters sit CS int32 apart: one 128-byte line each (same-line atomics are serialised)
constexpr int RS = 8;                  // depth segments (lanes) per ray in the ray cast

typedef float F2u __attribute__((ext_vector_type(2), aligned(4)));

struct Pose {
    float m[16];                       // row-major 4x4, passed by value in the kernel arguments
};

struct TsdfGrid {
    float voxel, trunc, unit_len;      // unit_len = 16 * voxel
    int base[3], dims[3];              // unit index range: [base, base + dims) per axis (x, y, z)
};

__device__ __forceinline__ int64_t unit_slot(const TsdfGrid &g, int ux, int uy, int uz) {
    const int x = ux - g.base[0], y = uy - g.base[1], z = uz - g.base[2];
    if ((unsigned)x >= (unsigned)g.dims[0] || (unsigned)y >= (unsigned)g.dims[1] || (unsigned)z >= (unsigned)g.dims[2]) return -1;
    return ((int64_t)z * g.dims[1] + y) * g.dims[0] + x;
}

// The sources of ONE step of the scene loop (reference :757-790 integrates every source frame of the step, one call each):
// depth maps, colours and poses travel by value in the kernel arguments (no upload, no device allocation per step).
constexpr int MAX_SRC = 8;
struct SrcSet {
    const float *depth[MAX_SRC];
    const uint8_t *rgb[MAX_SRC];
    Pose c2w[MAX_SRC], w2c[MAX_SRC];
    int n;
};

// pass 1 (blockIdx.y = source): open the units around the back-projected depth samples.  A unit's stamp word is
// (step_id << 8) | mask of the sources of this step that opened it; the lane that moves the word to this step's tag appends the
// unit ONCE to the step's brick list (the union over the sources) and allocates its brick if it never had one.
__global__ __launch_bounds__(256) void tsdf_touch_kernel(const SrcSet S, int H, int W, float fx, float fy, float cx, float cy,
                                  TsdfGrid g, float depth_trunc, int stride,
                                  int *__restrict__ table, int *__restrict__ stamp, int step_id, int *__restrict__ counters,
                                  int max_bricks, int *__restrict__ list, int max_list) {
    const int k = blockIdx.y;
    const float *__restrict__ depth = S.depth[k];
    const float *c2w = S.c2w[k].m;
    const int sw = (W + stride - 1) / stride, sh = (H + stride - 1) / stride;
    // eight lanes per depth sample, one per corner of the sample's box of units: lane c takes, per axis, the low unit (bit
    // clear) or the units above it (bit set: none when the box is one unit thick there) — almost always at most one unit per
    // lane, so the chain of atomic round trips below is walked once instead of once per unit of the box
    const int i8 = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = i8 >> 3, corner = i8 & 7;
    const bool in_grid = i < sw * sh;
    const int v = in_grid ? (i / sw) * stride : 0, u = in_grid ? (i % sw) * stride : 0;
    const float d = depth[v * W + u];
    const bool live = in_grid && d > 0.f && !(d > depth_trunc);      // (no early return: the wavefront's lanes cooperate below)
    // camera point, then world (c2w row-major 4x4)
    const float xc = __fmul_rn(__fdiv_rn(__fsub_rn((float)u, cx), fx), d);
    const float yc = __fmul_rn(__fdiv_rn(__fsub_rn((float)v, cy), fy), d);
    float p[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        p[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(c2w[r * 4 + 0], xc), __fmul_rn(c2w[r * 4 + 1], yc)), __fmul_rn(c2w[r * 4 + 2], d)),
                         c2w[r * 4 + 3]);
    int lo[3], hi[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        lo[r] = (int)floorf(__fdiv_rn(__fsub_rn(p[r], g.trunc), g.unit_len));
        hi[r] = (int)floorf(__fdiv_rn(__fadd_rn(p[r], g.trunc), g.unit_len));
    }
#if SGAM_TSDF_TOUCH_ABLATE == 1
    if (lo[0] == 12345678) counters[3 * CS] = hi[0];      // timing experiment: everything below removed
    return;
#endif
    const int tag = step_id << 8, bit = 1 << k;
    const unsigned long long lanes_below = (1ull << (threadIdx.x & 63)) - 1ull;
    // The units are visited in lock step by the wavefront (trip counts padded to the wavefront's maximum) so that the three
    // counters — list length, brick pool, outside-the-box diagnostic — take ONE atomic per wavefront and visit instead of
    // one per lane: every lane of every workgroup adding to the same word was the whole cost of this kernel.
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if ((corner >> r) & 1) lo[r] = lo[r] + 1;     // the units above the low one (an empty range when hi == the old lo)
        else hi[r] = lo[r];                           // the low unit alone
    }
    const bool some = live && hi[0] >= lo[0] && hi[1] >= lo[1] && hi[2] >= lo[2];
    const int nx = some ? hi[0] - lo[0] + 1 : 1, ny = some ? hi[1] - lo[1] + 1 : 1, nz = some ? hi[2] - lo[2] + 1 : 1;
    const int trips = some ? nx * ny * nz : 0;
    int max_trips = trips;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) max_trips = max(max_trips, __shfl_xor(max_trips, o, 64));
    // ... and by the WORKGROUP: its four wavefronts' requests for list slots / bricks are joined through LDS and thread 0 takes
    // them from the two counters with one atomic each (the list counter is ONE word: per-wavefront requests — 7 000 of them on
    // the noise scene — were 15 of the kernel's 20 us)
    __shared__ int s_trips[4], s_cnt[2][2][4], s_base[2][2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s_trips[wave] = max_trips;
    __syncthreads();
    max_trips = max(max(s_trips[0], s_trips[1]), max(s_trips[2], s_trips[3]));
    int n_outside = 0;
    for (int it = 0; it < max_trips; ++it) {
        const int pb = it & 1;                            // LDS slots alternate: a fast wavefront's next visit does not overwrite this one's
        const bool act = it < trips;
        const int ux = lo[0] + it % nx, uy = lo[1] + (it / nx) % ny, uz = lo[2] + it / (nx * ny);
        const int64_t s = act ? unit_slot(g, ux, uy, uz) : -1;
        const bool outside = act && s < 0;
        n_outside += __builtin_popcountll(__builtin_amdgcn_ballot_w64(outside));       // (wave-uniform; one atomic per wavefront at the end)
        // one lane per DISTINCT unit of the wavefront's visit talks to memory (neighbouring samples open the same unit: up to
        // 64 same-address atomics otherwise).  Election is register work: peel the lowest undecided lane, claim every lane
        // that holds the same slot.
        bool leader = false;
        {
            unsigned long long todo = __builtin_amdgcn_ballot_w64(s >= 0);
            while (todo) {
                const int l = __builtin_ctzll(todo);
                const int sl = __shfl((int)s, l, 64);
                const unsigned long long same = __builtin_amdgcn_ballot_w64(s >= 0 && (int)s == sl);
                if ((threadIdx.x & 63) == l) leader = true;
                todo &= ~same;
            }
        }
        // the stamp word moves to this step's tag by a returning atomicMax (tags grow with the step: exactly one lane on the
        // whole device sees the older word — `first`), the source's bit follows by a fire-and-forget atomicOr behind it
        // (same lane, same address: in order, so the bit lands in a word that already carries the tag); the table entry is
        // requested beside the atomic, not after it
        bool first = false;
        int brick = 0;
        if (leader) {
            const int prev = atomicMax(&stamp[s], tag);
            brick = table[s];                  // (the NEAR bit, if set, rides along: entries are only ever extended)
            atomicOr(&stamp[s], bit);
            first = (prev & ~0xff) != tag;
        }
#if SGAM_TSDF_TOUCH_ABLATE == 2
        first = false;                                     // timing experiment: stamps only, no list / allocation
#endif
        const bool need = first && brick < 0;
        const unsigned long long m_need = __builtin_amdgcn_ballot_w64(need), m_first = __builtin_amdgcn_ballot_w64(first);
        if (lane == 0) {
            s_cnt[pb][0][wave] = __builtin_popcountll(m_need);
            s_cnt[pb][1][wave] = __builtin_popcountll(m_first);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int n_need = (s_cnt[pb][0][0] + s_cnt[pb][0][1]) + (s_cnt[pb][0][2] + s_cnt[pb][0][3]);
            const int n_first = (s_cnt[pb][1][0] + s_cnt[pb][1][1]) + (s_cnt[pb][1][2] + s_cnt[pb][1][3]);
            const int b0 = n_need ? atomicAdd(&counters[0 * CS], n_need) : 0;
            const int b1 = n_first ? atomicAdd(&counters[1 * CS], n_first) : 0;
            s_base[pb][0] = b0;
            s_base[pb][1] = b1;
        }
        __syncthreads();
        int off_need = s_base[pb][0], off_list = s_base[pb][1];
        for (int w = 0; w < wave; ++w) {
            off_need += s_cnt[pb][0][w];
            off_list += s_cnt[pb][1][w];
        }
        bool listed = first;
        if (need) {
            brick = off_need + __builtin_popcountll(m_need & lanes_below);
            if (brick >= max_bricks) {
                atomicAdd(&counters[3 * CS], 1);          // pool exhausted (diagnostic); unit stays closed, and out of the list
                listed = false;
            } else {
                table[s] = brick;
            }
        }
        if (first) {                                      // (a slot is reserved for every first toucher; a unit that got no brick leaves
            const int li = off_list + __builtin_popcountll(m_first & lanes_below);      //  its slot pointing at a closed unit: skipped below)
            if (li < max_list) list[li] = listed ? (int)s : -1;
        }
    }
    if (n_outside && (threadIdx.x & 63) == 0) atomicAdd(&counters[2 * CS], n_outside);      // samples outside the scene box (diagnostic)
}

// per-pixel camera-distance multiplier sqrt(((u - cx) / fx)^2 + ((v - cy) / fy)^2 + 1) of the integration rule: a function of the
// pixel alone, tabulated once per (intrinsics, size) — the integrate kernel gathers it beside the depth instead of spending
// two divisions and a square root per voxel and source (same expression, same value)
__global__ void tsdf_ray_mult_kernel(int H, int W, float fx, float fy, float cx, float cy, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int v = i / W, u = i - v * W;
    const float rx = __fdiv_rn(__fsub_rn((float)u, cx), fx), ry = __fdiv_rn(__fsub_rn((float)v, cy), fy);
    out[i] = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry)), 1.0f));
}

// a / d and b / d, correctly rounded (= __fdiv_rn) for operands and quotients in the normal range: the instruction sequence the
// compiler emits for an IEEE fp32 division — reciprocal estimate, one Newton step on it, quotient, two residual corrections —
// without its range scaling (v_div_scale / v_div_fixup: identity for normal operands), the reciprocal shared by both quotients.
// The integration rule divides x fx and y fy by the same z for every voxel and source: 13 VALU instructions instead of 22.
// Outside the normal range (z <= 0, |quotient| < 2^-126) the results are not used / vanish in the `+ cx + 0.5` that follows.
__device__ __forceinline__ void div2_same_denominator(float a, float b, float d, float &qa, float &qb) {
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    qa = __fmul_rn(a, r);
    qb = __fmul_rn(b, r);
    qa = __builtin_fmaf(__builtin_fmaf(-d, qa, a), r, qa);
    qb = __builtin_fmaf(__builtin_fmaf(-d, qb, b), r, qb);
    qa = __builtin_fmaf(__builtin_fmaf(-d, qa, a), r, qa);
    qb = __builtin_fmaf(__builtin_fmaf(-d, qb, b), r, qb);
}

// pass 2: one workgroup per listed brick (grid-stride over the device-side list), 256 lanes x 16 voxels.  A voxel is loaded
// once, takes the update of every source of the step that opened its unit — in source order, in registers: the same value
// sequence as one integrate call per source — and is stored once.
//
// integrate_brick<NS, MASK, COLOR>: the body for a brick whose opening sources are the set bits of MASK (compile time; MASK = 0
// stands for "read the run-time mask": the generic form for steps of more than three sources).  Per voxel the projections of
// ALL of the brick's sources come first and their depth / multiplier gathers are issued together (clamped addresses, no branch
// in front of a load: a branch makes the compiler wait for the loads issued before it), then the updates are applied in source
// order: one memory round trip per voxel instead of one per source behind each other, and no arithmetic for a source that
// did not open the unit (half of the (brick, source) pairs of a step: the sources open different units of the same region).
template <int NS, int MASK, bool COLOR>
__device__ __forceinline__ int integrate_brick(const SrcSet &S, int rt_mask, int W, float fx, float fy, float cx, float cy,
                                               const TsdfGrid &g, float depth_trunc, float inv_trunc, float safe_w, float safe_h,
                                               float px, float py, float oz, float *__restrict__ bt, float *__restrict__ bw,
                                               float *__restrict__ bc, const float *__restrict__ ray_mult) {
    // the part of the camera transform that does not depend on z: (w2c[r][0] * px + w2c[r][1] * py), per source and row
    float cxy[NS][3];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
#pragma unroll
        for (int r = 0; r < 3; ++r) cxy[k][r] = __fadd_rn(__fmul_rn(S.w2c[k].m[r * 4 + 0], px), __fmul_rn(S.w2c[k].m[r * 4 + 1], py));
    }
    int near = 0;                 // this brick holds an observed voxel inside the truncation band (value < 1)
    // the column of 16 voxels this lane owns (z = 0..15), fetched ZG voxels at a time, one group ahead of its use: a wavefront
    // keeps 4 * ZG KB of brick data in flight (with one voxel ahead the kernel ran at the latency x occupancy limit, 2.9 TB/s)
    constexpr int ZG = SGAM_TSDF_ZG;
    float t_nx[ZG], w_nx[ZG];
#pragma unroll
    for (int j = 0; j < ZG; ++j) {
        t_nx[j] = bt[(j << 8) | threadIdx.x];
        w_nx[j] = bw[(j << 8) | threadIdx.x];
    }
    for (int zg = 0; zg < UR; zg += ZG) {
      float t_cur[ZG], w_cur[ZG];
#pragma unroll
      for (int j = 0; j < ZG; ++j) {
          t_cur[j] = t_nx[j];
          w_cur[j] = w_nx[j];
      }
      if (zg + ZG < UR) {
#pragma unroll
          for (int j = 0; j < ZG; ++j) {
              t_nx[j] = bt[((zg + ZG + j) << 8) | threadIdx.x];
              w_nx[j] = bw[((zg + ZG + j) << 8) | threadIdx.x];
          }
      }
#pragma unroll
      for (int j = 0; j < ZG; ++j) {
        const int z = zg + j;
        const int q = (z << 8) | threadIdx.x;
        float t = t_cur[j], w = w_cur[j];
        float c3[3] = {0.f, 0.f, 0.f};
        if (COLOR) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) c3[ch] = bc[q * 3 + ch];
        }
        const float pz = __fadd_rn(oz, __fmul_rn(__fadd_rn((float)z, 0.5f), g.voxel));
        // stage 1: project into the sources, gather depth + multiplier (+ colour)
        float cz[NS], d[NS], mult[NS];
        bool in[NS];
        uint8_t rgb3[NS][3];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (MASK != 0 && !((MASK >> k) & 1)) continue;            // (compile time)
            float c[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) c[r] = __fadd_rn(__fadd_rn(cxy[k][r], __fmul_rn(S.w2c[k].m[r * 4 + 2], pz)), S.w2c[k].m[r * 4 + 3]);
            cz[k] = c[2];
            float qx, qy;
            div2_same_denominator(__fmul_rn(c[0], fx), __fmul_rn(c[1], fy), c[2], qx, qy);
            const float uf = __fadd_rn(__fadd_rn(qx, cx), 0.5f);
            const float vf = __fadd_rn(__fadd_rn(qy, cy), 0.5f);
            in[k] = (MASK != 0 || ((rt_mask >> k) & 1)) && c[2] > 0.f && uf >= 0.0001f && uf < safe_w && vf >= 0.0001f && vf < safe_h;
            const int pix = in[k] ? (int)vf * W + (int)uf : 0;
            const int ks = MASK != 0 ? k : (k < S.n ? k : 0);
            d[k] = S.depth[ks][pix];
            mult[k] = ray_mult[pix];
            if (COLOR) {
                const uint8_t *rk = S.rgb[ks] + (int64_t)pix * 3;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) rgb3[k][ch] = rk[ch];
            }
        }
        // stage 2: the updates, in source order
        bool changed = false;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (MASK != 0 && !((MASK >> k) & 1)) continue;            // (compile time)
            const float sdf = __fmul_rn(__fsub_rn(d[k], cz[k]), mult[k]);
            if (in[k] && d[k] > 0.f && !(d[k] > depth_trunc) && sdf > -g.trunc) {
                const float tv = fminf(1.0f, __fmul_rn(sdf, inv_trunc));
                const float w1 = __fadd_rn(w, 1.0f);
                t = __fdiv_rn(__fadd_rn(__fmul_rn(t, w), tv), w1);
                if (COLOR) {
                    // TSDFVolumeColorType::RGB8: color <- (color * w + rgb(u, v)) / (w + 1), per channel, 0..255
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) c3[ch] = __fdiv_rn(__fadd_rn(__fmul_rn(c3[ch], w), (float)rgb3[k][ch]), w1);
                }
                w = w1;
                near |= t < 1.0f;
                changed = true;
            }
        }
        if (changed) {
            if (t != t_cur[j]) bt[q] = t;      // (free space seen as free space again: (1 w + 1) / (w + 1) = 1 — only the weight moves)
            bw[q] = w;
            if (COLOR) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) bc[q * 3 + ch] = c3[ch];
            }
        }
      }
    }
    return near;
}

template <int NS, bool COLOR>
__global__ __launch_bounds__(256, SGAM_TSDF_LB) void tsdf_integrate_kernel(const SrcSet S, int H, int W, float fx, float fy,
                                                             float cx, float cy, TsdfGrid g,
                                                             float depth_trunc, int *__restrict__ table, const int *__restrict__ stamp,
                                                             const int *__restrict__ counters,
                                                             const int *__restrict__ list, int max_list,
                                                             float *__restrict__ tsdf, float *__restrict__ weight,
                                                             float *__restrict__ color, const float *__restrict__ ray_mult) {
    int n = counters[1 * CS];
    if (n > max_list) n = max_list;
    const float inv_trunc = __fdiv_rn(1.0f, g.trunc);
    const float safe_w = __fsub_rn((float)W, 0.0001f), safe_h = __fsub_rn((float)H, 0.0001f);
    const int x = threadIdx.x & 15, y = threadIdx.x >> 4;
    for (int li = blockIdx.x; li < n; li += gridDim.x) {
        const int s = list[li];
        if (s < 0) continue;                              // (a unit the exhausted brick pool could not open)
        const int brick = table[s] & BRICK_MASK;
        const int mask = stamp[s] & ((1 << NS) - 1) & ((1 << S.n) - 1);
        const int ux = s % g.dims[0] + g.base[0];
        const int uy = (s / g.dims[0]) % g.dims[1] + g.base[1];
        const int uz = s / (g.dims[0] * g.dims[1]) + g.base[2];
        float *bt = tsdf + (int64_t)brick * UV, *bw = weight + (int64_t)brick * UV;
        float *bc = COLOR ? color + (int64_t)brick * UV * 3 : nullptr;
        // voxel centre: unit origin + (i + 0.5) * voxel
        const float px = __fadd_rn(__fmul_rn((float)ux, g.unit_len), __fmul_rn(__fadd_rn((float)x, 0.5f), g.voxel));
        const float py = __fadd_rn(__fmul_rn((float)uy, g.unit_len), __fmul_rn(__fadd_rn((float)y, 0.5f), g.voxel));
        const float oz = __fmul_rn((float)uz, g.unit_len);
#define SGAM_BRICK(M) integrate_brick<NS, M, COLOR>(S, mask, W, fx, fy, cx, cy, g, depth_trunc, inv_trunc, safe_w, safe_h, px, py, oz, bt, bw, bc, ray_mult)
        int near;
        if (NS <= 3) {                 // one specialised body per set of opening sources (the switch is uniform over the workgroup)
            switch (mask) {
                case 1: near = SGAM_BRICK(1); break;
                case 2: near = SGAM_BRICK(NS >= 2 ? 2 : 1); break;
                case 3: near = SGAM_BRICK(NS >= 2 ? 3 : 1); break;
                case 4: near = SGAM_BRICK(NS >= 3 ? 4 : 1); break;
                case 5: near = SGAM_BRICK(NS >= 3 ? 5 : 1); break;
                case 6: near = SGAM_BRICK(NS >= 3 ? 6 : 1); break;
                case 7: near = SGAM_BRICK(NS >= 3 ? 7 : 1); break;
                default: near = 0; break;
            }
        } else {
            near = SGAM_BRICK(0);
        }
#undef SGAM_BRICK
        // a weighted mean of values <= 1 that is < 1 once stays < 1: the flag is monotone, a plain store suffices
        if (__syncthreads_or(near) && threadIdx.x == 0) atomicOr(&table[s], NEAR_BIT);
    }
}

// TSDF at voxel lattice point (ix, iy, iz) (global voxel indices, centres at (i + 0.5) * voxel); false when unobserved
__device__ __forceinline__ bool lattice(const TsdfGrid &g, const int *table, const float *tsdf, int ix, int iy, int iz, float &val) {
    const int64_t s = unit_slot(g, ix >> 4, iy >> 4, iz >> 4);        // arithmetic shift = floor division by 16
    if (s < 0) return false;
    const int brick = table[s];
    if (brick < 0) return false;
    const int q = ((iz & 15) << 8) | ((iy & 15) << 4) | (ix & 15);
    val = tsdf[(int64_t)(brick & BRICK_MASK) * UV + q];
    return val <= 1.0f;
}

// TSDF at world point p: trilinear when all eight surrounding lattice points are observed, else the nearest lattice
// point's value, else false.  (The point-extraction kernel's sampler; the ray cast carries its own copy of this rule with
// every load of a step in flight at once — SAMPLE_* below.)
__device__ __forceinline__ bool sample(const TsdfGrid &g, const int *table, const float *tsdf, const float *p, float inv_voxel,
                                       float &val) {
    float f[3];
    int i0[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float t = __fsub_rn(__fmul_rn(p[r], inv_voxel), 0.5f);      // lattice coordinate
        const float fl = floorf(t);
        i0[r] = (int)fl;
        f[r] = __fsub_rn(t, fl);
    }
    float c[8];
    bool all = true;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        all = lattice(g, table, tsdf, i0[0] + (k & 1), i0[1] + ((k >> 1) & 1), i0[2] + (k >> 2), c[k]) && all;
    if (!all) {
        // a cell with an unobserved corner (typically just behind an obliquely seen surface, where the truncation band
        // is thinner than a voxel diagonal): fall back to the nearest lattice point, if that one was observed
        return lattice(g, table, tsdf, i0[0] + (f[0] >= 0.5f ? 1 : 0), i0[1] + (f[1] >= 0.5f ? 1 : 0),
                       i0[2] + (f[2] >= 0.5f ? 1 : 0), val);
    }
    // x, then y, then z
    const float c00 = __fadd_rn(c[0], __fmul_rn(f[0], __fsub_rn(c[1], c[0])));
    const float c10 = __fadd_rn(c[2], __fmul_rn(f[0], __fsub_rn(c[3], c[2])));
    const float c01 = __fadd_rn(c[4], __fmul_rn(f[0], __fsub_rn(c[5], c[4])));
    const float c11 = __fadd_rn(c[6], __fmul_rn(f[0], __fsub_rn(c[7], c[6])));
    const float c0 = __fadd_rn(c00, __fmul_rn(f[1], __fsub_rn(c10, c00)));
    const float c1 = __fadd_rn(c01, __fmul_rn(f[1], __fsub_rn(c11, c01)));
    val = __fadd_rn(c0, __fmul_rn(f[2], __fsub_rn(c1, c0)));
    return true;
}

// ---- the ray cast's form of the same rule, written without data-dependent branches around the loads: a march step is a
// chain of dependent memory round trips (unit table -> brick values), and a branch per corner turns the eight corners of a
// cell that straddles a brick face into sixteen round trips IN SERIES — for the whole wavefront, as soon as one lane has such
// a cell (with 64 lanes: every step).  Here every table entry a step needs is requested at once (index 0 stands in for a
// unit outside the box), then every value at once: two round trips per step, whatever the lanes' cells look like.
// unit-table index of unit (ux, uy, uz) -> `idx` (0 when outside the box) and `ok`
#define TSDF_UNIT_INDEX(ux, uy, uz, idx, ok)                                                                                      \
    do {                                                                                                                          \
        const unsigned x_ = (unsigned)((ux) - g.base[0]), y_ = (unsigned)((uy) - g.base[1]), z_ = (unsigned)((uz) - g.base[2]);   \
        ok = (x_ < (unsigned)g.dims[0]) & (y_ < (unsigned)g.dims[1]) & (z_ < (unsigned)g.dims[2]);                                \
        idx = (z_ * (unsigned)g.dims[1] + y_) * (unsigned)g.dims[0] + x_;                                                         \
        idx = ok ? idx : 0u;                                                                                                      \
    } while (0)
// value of lattice point (ix, iy, iz) in the brick of table entry e (2 = unobserved when e < 0); the load is unconditional
#define TSDF_CORNER(e, ix, iy, iz, out)                                                                                           \
    do {                                                                                                                          \
        const unsigned q_ = (((unsigned)(iz)&15u) << 8) | (((unsigned)(iy)&15u) << 4) | ((unsigned)(ix)&15u);                     \
        const float x_ = tsdf[(e) >= 0 ? (size_t)((e)&BRICK_MASK) * UV + q_ : (size_t)0];                                         \
        out = (e) >= 0 ? x_ : 2.0f;                                                                                               \
    } while (0)

// depth render: per pixel, march the camera ray; parameter t = view-space z (so the result needs no conversion).
// Unopened units are crossed in one step each (exit distance of the unit along the ray + a quarter voxel), opened ones with half-voxel steps or, in observed free
// space, 0.8 x the distance the TSDF value guarantees.
//
// Every ray is cut into RS depth segments (each starts with no history and runs two voxels into the next segment so that a
// crossing on a boundary is seen by the earlier one); the nearest hit wins.  Thread mapping: a WAVEFRONT marches one
// segment of an 8 x 8 pixel tile, the RS wavefronts of a workgroup the RS segments of that tile.  Neighbouring pixels'
// rays at one depth are about a voxel apart, so the 64 lanes of a load touch a few 64-byte brick rows instead of 64
// unrelated bricks (the march is bound by the number of cache lines a wavefront's load touches, not by arithmetic); a
// segment whose samples are already behind the nearest hit found so far for its pixel (LDS, atomicMin on the float bits)
// stops — it could only find a farther crossing.
__global__ __launch_bounds__(64 * RS) void tsdf_raycast_kernel(int H, int W, float fx, float fy, float cx, float cy, const Pose c2w_, TsdfGrid g,
                                    float z_near, float z_far, const int *__restrict__ table, const float *__restrict__ tsdf,
                                    float *__restrict__ out, const float *__restrict__ color, float *__restrict__ color_out) {
    __shared__ unsigned best_bits[64];
    const float *c2w = c2w_.m;
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int tiles_x = (W + 7) >> 3;
    // workgroup b runs on XCD b % 8 (observed placement; for speed only): each XCD gets a contiguous run of tiles — a strip of
    // the image — so that the bricks its rays pierce are fetched into ONE L2 instead of eight
    const int nb = gridDim.x, per = (nb + 7) >> 3;
    int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((nb & 7) != 0) tile = blockIdx.x;                      // (a tile count that is not a multiple of 8: plain order)
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int v = ty * 8 + (lane >> 3), u = tx * 8 + (lane & 7);
    const bool live = v < H && u < W;
    const int i = v * W + u;
    if (seg == 0) best_bits[lane] = 0x7f7fffffu;              // FLT_MAX
    __syncthreads();
    // pixel centres at integer coordinates, as in the reference's pinhole set-up (cx, cy in pixel units)
    const float rx = __fdiv_rn(__fsub_rn((float)u, cx), fx), ry = __fdiv_rn(__fsub_rn((float)v, cy), fy);
    float o[3], dir[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        o[r] = c2w[r * 4 + 3];
        dir[r] = __fadd_rn(__fadd_rn(__fmul_rn(c2w[r * 4 + 0], rx), __fmul_rn(c2w[r * 4 + 1], ry)), c2w[r * 4 + 2]);   // per unit z
    }
    const float inv_voxel = __fdiv_rn(1.0f, g.voxel);
    // the march's own divisions are by quantities fixed per ray: reciprocals once, products per step (a step costs VALU
    // issue slots — six IEEE divisions were a third of them)
    const float inv_unit = __fdiv_rn(1.0f, g.unit_len);
    float inv_dir[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) inv_dir[r] = __fdiv_rn(1.0f, dir[r]);           // (inf where dir == 0: not used there)
    const float fine = __fmul_rn(0.5f, g.voxel), eps = __fmul_rn(0.25f, g.voxel);
    const float seg_len = __fdiv_rn(__fsub_rn(z_far, z_near), (float)RS);
    const float t_begin = __fadd_rn(z_near, __fmul_rn((float)seg, seg_len));
    const float t_end = fminf(z_far, __fadd_rn(__fadd_rn(t_begin, seg_len), __fmul_rn(2.0f, g.voxel)));
    float t = t_begin, prev_t = 0.f, prev_val = 0.f, depth = 0.f;
    bool prev_ok = false;
#ifdef SGAM_TSDF_DEBUG_STEPS
    int n_coarse = 0, n_fine = 0;
#endif
    while (live && t < t_end) {
        // a crossing this lane can still find lies beyond prev_t: pointless once a nearer one is known for the pixel
        if (__float_as_uint(prev_t) >= __hip_atomic_load(&best_bits[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        float p[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) p[r] = __fadd_rn(o[r], __fmul_rn(dir[r], t));
        float uf[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) uf[r] = floorf(__fmul_rn(p[r], inv_unit));
        // ---- stage 1: every unit-table entry of the step
        unsigned s_here; bool in_box;
        TSDF_UNIT_INDEX((int)uf[0], (int)uf[1], (int)uf[2], s_here, in_box);
        const float l0 = __fsub_rn(__fmul_rn(p[0], inv_voxel), 0.5f), l1 = __fsub_rn(__fmul_rn(p[1], inv_voxel), 0.5f),
                    l2 = __fsub_rn(__fmul_rn(p[2], inv_voxel), 0.5f);             // lattice coordinates
        const float fl0 = floorf(l0), fl1 = floorf(l1), fl2 = floorf(l2);
        const int ix = (int)fl0, iy = (int)fl1, iz = (int)fl2;                    // the cell's low corner
        const float f0 = __fsub_rn(l0, fl0), f1 = __fsub_rn(l1, fl1), f2 = __fsub_rn(l2, fl2);
        const bool straddles = ((ix & 15) == 15) | ((iy & 15) == 15) | ((iz & 15) == 15);
        const bool single = __builtin_amdgcn_ballot_w64(straddles) == 0;          // wave-uniform: no lane's cell crosses a brick face
        const int ux0 = ix >> 4, uy0 = iy >> 4, uz0 = iz >> 4;                    // arithmetic shift = floor division by 16
        const int ux1 = (ix + 1) >> 4, uy1 = (iy + 1) >> 4, uz1 = (iz + 1) >> 4;
        unsigned s0, s1, s2, s3, s4, s5, s6, s7;
        bool k0, k1, k2, k3, k4, k5, k6, k7;
        TSDF_UNIT_INDEX(ux0, uy0, uz0, s0, k0);
        int e_here, e0, e1, e2, e3, e4, e5, e6, e7;
        if (single) {
            e_here = table[s_here];
            e0 = table[s0];
            asm volatile("" : "+v"(e_here), "+v"(e0));       // both requests in flight before either is consumed
            e0 = k0 ? e0 : -1;
            e1 = e2 = e3 = e4 = e5 = e6 = e7 = e0;
        } else {
            // the eight units of the cell's corners: per-axis pieces shared between the corners
            const unsigned x0 = (unsigned)(ux0 - g.base[0]), x1 = (unsigned)(ux1 - g.base[0]);
            const unsigned y0 = (unsigned)(uy0 - g.base[1]), y1 = (unsigned)(uy1 - g.base[1]);
            const unsigned z0 = (unsigned)(uz0 - g.base[2]), z1 = (unsigned)(uz1 - g.base[2]);
            const unsigned dx = (unsigned)g.dims[0], dy = (unsigned)g.dims[1], dz = (unsigned)g.dims[2];
            const bool okx0 = x0 < dx, okx1 = x1 < dx, oky0 = y0 < dy, oky1 = y1 < dy, okz0 = z0 < dz, okz1 = z1 < dz;
            const unsigned r00 = (z0 * dy + y0) * dx, r10 = (z0 * dy + y1) * dx, r01 = (z1 * dy + y0) * dx, r11 = (z1 * dy + y1) * dx;
            const bool o00 = okz0 & oky0, o10 = okz0 & oky1, o01 = okz1 & oky0, o11 = okz1 & oky1;
            k1 = o00 & okx1; k2 = o10 & okx0; k3 = o10 & okx1; k4 = o01 & okx0; k5 = o01 & okx1; k6 = o11 & okx0; k7 = o11 & okx1;
            s1 = k1 ? r00 + x1 : 0u; s2 = k2 ? r10 + x0 : 0u; s3 = k3 ? r10 + x1 : 0u; s4 = k4 ? r01 + x0 : 0u;
            s5 = k5 ? r01 + x1 : 0u; s6 = k6 ? r11 + x0 : 0u; s7 = k7 ? r11 + x1 : 0u;
            e_here = table[s_here];
            e0 = table[s0]; e1 = table[s1]; e2 = table[s2]; e3 = table[s3];
            e4 = table[s4]; e5 = table[s5]; e6 = table[s6]; e7 = table[s7];
            e0 = k0 ? e0 : -1; e1 = k1 ? e1 : -1; e2 = k2 ? e2 : -1; e3 = k3 ? e3 : -1;
            e4 = k4 ? e4 : -1; e5 = k5 ? e5 : -1; e6 = k6 ? e6 : -1; e7 = k7 ? e7 : -1;
        }
        // only bricks that hold part of the truncation band can contain the surface: everything else (unopened units,
        // bricks of observed free space, bricks with nothing observed) is crossed like empty space
        const int brick_here = in_box ? e_here : -1;
        const bool open = brick_here >= 0 && (brick_here & NEAR_BIT) != 0;
        // such a unit is left in ONE step: distance (in t) to the nearest of its faces the ray is heading for
        float coarse = z_far;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (dir[r] > 0.f) coarse = fminf(coarse, __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(uf[r], 1.0f), g.unit_len), p[r]), inv_dir[r]));
            else if (dir[r] < 0.f) coarse = fminf(coarse, __fmul_rn(__fsub_rn(__fmul_rn(uf[r], g.unit_len), p[r]), inv_dir[r]));
        }
        coarse = __fadd_rn(fmaxf(coarse, 0.f), eps);
        // ---- stage 2: the eight corner values (when some lane of the wavefront marches a band brick)
        float val = 0.f;
        bool ok = false;
        if (__builtin_amdgcn_ballot_w64(open) != 0) {
            float v0, v1, v2, v3, v4, v5, v6, v7;
            if (single) {
                // the whole cell lies in one brick: four 8-byte loads (x and x + 1 are adjacent floats; 4-byte aligned)
                const unsigned q0 = (((unsigned)iz & 15u) << 8) | (((unsigned)iy & 15u) << 4) | ((unsigned)ix & 15u);
                const float *bt = tsdf + (e0 >= 0 ? (size_t)(e0 & BRICK_MASK) * UV + q0 : (size_t)0);
                const F2u a = *(const F2u *)(bt), b = *(const F2u *)(bt + 16), c = *(const F2u *)(bt + 256), d = *(const F2u *)(bt + 272);
                const bool have = e0 >= 0;
                v0 = have ? a.x : 2.0f; v1 = have ? a.y : 2.0f; v2 = have ? b.x : 2.0f; v3 = have ? b.y : 2.0f;
                v4 = have ? c.x : 2.0f; v5 = have ? c.y : 2.0f; v6 = have ? d.x : 2.0f; v7 = have ? d.y : 2.0f;
            } else {
                TSDF_CORNER(e0, ix, iy, iz, v0);
                TSDF_CORNER(e1, ix + 1, iy, iz, v1);
                TSDF_CORNER(e2, ix, iy + 1, iz, v2);
                TSDF_CORNER(e3, ix + 1, iy + 1, iz, v3);
                TSDF_CORNER(e4, ix, iy, iz + 1, v4);
                TSDF_CORNER(e5, ix + 1, iy, iz + 1, v5);
                TSDF_CORNER(e6, ix, iy + 1, iz + 1, v6);
                TSDF_CORNER(e7, ix + 1, iy + 1, iz + 1, v7);
            }
            const bool all = (v0 <= 1.0f) & (v1 <= 1.0f) & (v2 <= 1.0f) & (v3 <= 1.0f) & (v4 <= 1.0f) & (v5 <= 1.0f) & (v6 <= 1.0f) &
                             (v7 <= 1.0f);
            // a cell with an unobserved corner (typically just behind an obliquely seen surface, where the truncation band is
            // thinner than a voxel diagonal): the nearest lattice point — one of the eight corners — if that one was observed
            const bool nx = f0 >= 0.5f, ny = f1 >= 0.5f, nz = f2 >= 0.5f;
            const float a0 = nx ? v1 : v0, a1 = nx ? v3 : v2, a2 = nx ? v5 : v4, a3 = nx ? v7 : v6;
            const float b0 = ny ? a1 : a0, b1 = ny ? a3 : a2;
            const float nearest = nz ? b1 : b0;
            // trilinear: x, then y, then z
            const float c00 = __fadd_rn(v0, __fmul_rn(f0, __fsub_rn(v1, v0)));
            const float c10 = __fadd_rn(v2, __fmul_rn(f0, __fsub_rn(v3, v2)));
            const float c01 = __fadd_rn(v4, __fmul_rn(f0, __fsub_rn(v5, v4)));
            const float c11 = __fadd_rn(v6, __fmul_rn(f0, __fsub_rn(v7, v6)));
            const float c0 = __fadd_rn(c00, __fmul_rn(f1, __fsub_rn(c10, c00)));
            const float c1 = __fadd_rn(c01, __fmul_rn(f1, __fsub_rn(c11, c01)));
            const float tri = __fadd_rn(c0, __fmul_rn(f2, __fsub_rn(c1, c0)));
            ok = open & (all | (nearest <= 1.0f));
            val = ok ? (all ? tri : nearest) : 0.f;
        }
        if (ok && prev_ok && prev_val > 0.f && val <= 0.f) {
            // linear zero crossing between the two samples
            depth = __fadd_rn(prev_t, __fmul_rn(__fsub_rn(t, prev_t), __fdiv_rn(prev_val, __fsub_rn(prev_val, val))));
            if (depth > 0.f) atomicMin(&best_bits[lane], __float_as_uint(depth));
            break;
        }
        prev_ok = ok;
        prev_val = val;
        prev_t = t;
        // in front of the surface the TSDF itself bounds the free distance (val * trunc along the ray, taken at 0.8)
        // (unobserved space inside an opened unit: whole-voxel steps)
        const float stride = ok ? (val > 0.f ? fmaxf(fine, __fmul_rn(__fmul_rn(0.8f, val), g.trunc)) : fine) : g.voxel;
        t = __fadd_rn(t, open ? stride : coarse);
#ifdef SGAM_TSDF_DEBUG_STEPS
        if (open) ++n_fine; else ++n_coarse;
#endif
    }
#ifdef SGAM_TSDF_DEBUG_STEPS
    // instrumented build (scripts/tsdf_steps.py): out = 10000 x coarse steps + fine steps of the pixel, summed over its segments
    __shared__ unsigned dbg_steps[64];
    if (seg == 0) dbg_steps[lane] = 0;
    __syncthreads();
    atomicAdd(&dbg_steps[lane], (unsigned)(n_coarse * 10000 + n_fine));
    __syncthreads();
    if (live && seg == 0) out[i] = (float)dbg_steps[lane];
    return;
#endif
    __syncthreads();
    // nearest hit over the RS segments of this pixel (0 = no hit)
    const float best = __uint_as_float(best_bits[lane]);
    const bool hit = best < 3.0e38f;
    if (live && seg == 0) out[i] = hit ? best : 0.f;
    if (color_out && live) {
        // colour of the fused surface at the hit: the nearest voxel's running mean (0 where nothing is hit).  The lane that
        // found the winning crossing writes it (equal depths on two lanes: both write the same voxel's colour).
        if (seg == 0 && !hit) color_out[i * 3] = color_out[i * 3 + 1] = color_out[i * 3 + 2] = 0.f;
        if (depth > 0.f && depth == best) {
            int vi[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) vi[r] = (int)floorf(__fmul_rn(__fadd_rn(o[r], __fmul_rn(dir[r], depth)), inv_voxel));
            const int64_t s = unit_slot(g, vi[0] >> 4, vi[1] >> 4, vi[2] >> 4);
            const int brick = s >= 0 ? table[s] : -1;
            float c3[3] = {0.f, 0.f, 0.f};
            if (brick >= 0) {
                const int q = ((vi[2] & 15) << 8) | ((vi[1] & 15) << 4) | (vi[0] & 15);
                const float *bc = color + ((int64_t)(brick & BRICK_MASK) * UV + q) * 3;
                c3[0] = bc[0]; c3[1] = bc[1]; c3[2] = bc[2];
            }
            color_out[i * 3] = c3[0]; color_out[i * 3 + 1] = c3[1]; color_out[i * 3 + 2] = c3[2];
        }
    }
}

int grid_ok(const sgam_tsdf_grid *g) {
    if (!g || !(g->voxel_length > 0.f) || !(g->sdf_trunc > 0.f)) return 0;
    for (int r = 0; r < 3; ++r)
        if (g->unit_dims[r] <= 0) return 0;
    return ((int64_t)g->unit_dims[0] * g->unit_dims[1] * g->unit_dims[2]) < (1ll << 31);
}

// ------------------------------------------------------------------------------------------------
// Zero-crossing point extraction from the bricks: the `volume.extract_point_cloud()` the reference writes to
// rgbd_integrated_mesh.ply at the end of a run (sgam/inference_pipeline.py:446-450).  Open3D's published rule
// (ScalableTSDFVolume::ExtractPointCloud): for every observed voxel with |tsdf| < 0.98 and each of its +x / +y / +z
// neighbours (same test), a sign change between the two puts ONE point on the edge, linearly interpolated by the two
// |tsdf| values; colour interpolated the same way; the normal is the normalised central difference of the TSDF field at the
// point (one voxel each way, through the ray cast's sampler).  One workgroup per opened unit (grid-stride over the unit
// table), one atomic per point for its slot in the output; `key` = ((unit slot * 4096 + voxel) * 3 + axis) lets the host put
// the points in a run-independent order.  Parity: unpinned against Open3D (absent), held against the independent dense
// float64 reference (oracle/tsdf_dense.py) in tests/test_gpu_tsdf.py.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tsdf_extract_kernel(TsdfGrid g, int64_t n_units, const int *__restrict__ table,
                                                           const float *__restrict__ tsdf, const float *__restrict__ color,
                                                           unsigned long long *__restrict__ counter, int64_t max_points,
                                                           float *__restrict__ points, float *__restrict__ normals,
                                                           float *__restrict__ colors, int64_t *__restrict__ keys) {
    const float inv_voxel = __fdiv_rn(1.0f, g.voxel);
    for (int64_t slot = blockIdx.x; slot < n_units; slot += gridDim.x) {
        const int entry = table[slot];
        if (entry < 0 || !(entry & NEAR_BIT)) continue;              // unopened, or no voxel inside the truncation band
        const int brick = entry & BRICK_MASK;
        const int ux = (int)(slot % g.dims[0]) + g.base[0], uy = (int)((slot / g.dims[0]) % g.dims[1]) + g.base[1],
                  uz = (int)(slot / ((int64_t)g.dims[0] * g.dims[1])) + g.base[2];
        for (int q = threadIdx.x; q < UV; q += 256) {
            const float f0 = tsdf[(int64_t)brick * UV + q];
            if (!(f0 < 0.98f && f0 >= -0.98f)) continue;             // (unobserved voxels hold the sentinel 2.0)
            const int ix = ux * UR + (q & 15), iy = uy * UR + ((q >> 4) & 15), iz = uz * UR + (q >> 8);
            const float p0[3] = {__fmul_rn((float)ix + 0.5f, g.voxel), __fmul_rn((float)iy + 0.5f, g.voxel),
                                 __fmul_rn((float)iz + 0.5f, g.voxel)};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int jx = ix + (a == 0), jy = iy + (a == 1), jz = iz + (a == 2);
                float f1;
                if (!lattice(g, table, tsdf, jx, jy, jz, f1)) continue;
                if (!(f1 < 0.98f && f1 >= -0.98f) || !(f0 * f1 < 0.f)) continue;
                const float r0 = fabsf(f0), r1 = fabsf(f1), den = __fadd_rn(r0, r1);
                float p[3] = {p0[0], p0[1], p0[2]};
                p[a] = __fdiv_rn(__fadd_rn(__fmul_rn(p0[a], r1), __fmul_rn(__fadd_rn(p0[a], g.voxel), r0)), den);
                const unsigned long long at = atomicAdd(counter, 1ull);
                if ((int64_t)at >= max_points || !points) continue;  // counting pass / caller's buffer exhausted
                points[at * 3 + 0] = p[0];
                points[at * 3 + 1] = p[1];
                points[at * 3 + 2] = p[2];
                if (keys) keys[at] = (slot * UV + q) * 3 + a;
                if (colors && color) {
                    const int64_t s1 = unit_slot(g, jx >> 4, jy >> 4, jz >> 4);
                    const int b1 = table[s1] & BRICK_MASK;
                    const int q1 = ((jz & 15) << 8) | ((jy & 15) << 4) | (jx & 15);
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const float c0 = color[((int64_t)brick * UV + q) * 3 + ch], c1 = color[((int64_t)b1 * UV + q1) * 3 + ch];
                        colors[at * 3 + ch] = __fdiv_rn(__fadd_rn(__fmul_rn(c0, r1), __fmul_rn(c1, r0)), den);
                    }
                }
                if (normals) {
                    float n[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        float pa[3] = {p[0], p[1], p[2]}, pb[3] = {p[0], p[1], p[2]};
                        pa[r] = __fadd_rn(p[r], g.voxel);
                        pb[r] = __fsub_rn(p[r], g.voxel);
                        float va, vb;
                        const bool oa = sample(g, table, tsdf, pa, inv_voxel, va), ob = sample(g, table, tsdf, pb, inv_voxel, vb);
                        n[r] = (oa && ob) ? __fsub_rn(va, vb) : 0.f;
                    }
                    const float len = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(n[0], n[0]), __fmul_rn(n[1], n[1])), __fmul_rn(n[2], n[2])));
#pragma unroll
                    for (int r = 0; r < 3; ++r) normals[at * 3 + r] = len > 0.f ? __fdiv_rn(n[r], len) : 0.f;
                }
            }
        }
    }
}

TsdfGrid to_dev(const sgam_tsdf_grid *g) {
    TsdfGrid d;
    d.voxel = g->voxel_length;
    d.trunc = g->sdf_trunc;
    d.unit_len = g->voxel_length * (float)UR;
    for (int r = 0; r < 3; ++r) {
        d.base[r] = g->unit_base[r];
        d.dims[r] = g->unit_dims[r];
    }
    return d;
}

}  // namespace

extern "C" int sgam_tsdf_integrate_srcs_f32(const sgam_tsdf_grid *grid, const sgam_tsdf_src *srcs, int32_t n_src, int32_t H, int32_t W,
                                            float fx, float fy, float cx, float cy, float depth_trunc, int32_t step_id,
                                            int32_t *unit_table, int32_t *unit_stamp, int32_t *counters, int32_t *brick_list,
                                            int32_t max_list, float *brick_tsdf, float *brick_weight, int32_t max_bricks,
                                            float *brick_color, const float *ray_mult, void *stream) {
    if (!grid_ok(grid) || !srcs || !ray_mult || n_src <= 0 || n_src > MAX_SRC || !(fx > 0.f) || !(fy > 0.f) || !unit_table || !unit_stamp || !counters ||
        !brick_list || !brick_tsdf || !brick_weight || H <= 0 || W <= 0 || max_list <= 0 || max_bricks <= 0 || max_bricks > BRICK_MASK ||
        step_id <= 0 || step_id >= (1 << 23))
        return SGAM_EINVAL;
    SrcSet S;
    S.n = n_src;
    for (int k = 0; k < MAX_SRC; ++k) {
        const sgam_tsdf_src &src = srcs[k < n_src ? k : 0];
        if (k < n_src && (!src.depth || (src.rgb_u8 == nullptr) != (brick_color == nullptr))) return SGAM_EINVAL;   // colour: both or neither
        S.depth[k] = src.depth;
        S.rgb[k] = src.rgb_u8;
        for (int i = 0; i < 16; ++i) {
            S.c2w[k].m[i] = src.cam2world[i];
            S.w2c[k].m[i] = src.world2cam[i];
        }
    }
    const TsdfGrid g = to_dev(grid);
    hipStream_t s = sgam_stream(stream);
    hipError_t e = hipMemsetAsync(counters + 1 * CS, 0, sizeof(int32_t), s);       // this step's list length
    if (e != hipSuccess) return (int)e;
    const int stride = 4;                                                      // Open3D depth_sampling_stride
    const int ns = ((W + stride - 1) / stride) * ((H + stride - 1) / stride);
    SGAM_KLAUNCH(tsdf_touch_kernel, dim3(sgam_cdiv((int64_t)ns * 8, 256), n_src), dim3(256), 0, s, S, H, W, fx, fy, cx, cy,
                       g, depth_trunc, stride, unit_table, unit_stamp, step_id, counters, max_bricks, brick_list,
                       max_list);
    SGAM_LAUNCH_CHECK();
#define SGAM_TSDF_INTEGRATE(NS, COLOR)                                                                                            \
    SGAM_KLAUNCH((tsdf_integrate_kernel<NS, COLOR>), dim3(2048), dim3(256), 0, s, S, H, W, fx, fy, cx, cy, g, depth_trunc, unit_table, \
                 unit_stamp, counters, brick_list, max_list, brick_tsdf, brick_weight, brick_color, ray_mult)
    // kernels for 1, 2, 3 (GoogleEarth: <= 3 sources), 5 (CLEVR: <= 5) and 8 sources; a step with 4 or 6 .. 7 runs the next larger
    const int nk = n_src <= 3 ? n_src : (n_src <= 5 ? 5 : 8);
    if (brick_color) {
        if (nk == 1) SGAM_TSDF_INTEGRATE(1, true);
        else if (nk == 2) SGAM_TSDF_INTEGRATE(2, true);
        else if (nk == 3) SGAM_TSDF_INTEGRATE(3, true);
        else if (nk == 5) SGAM_TSDF_INTEGRATE(5, true);
        else SGAM_TSDF_INTEGRATE(8, true);
    } else {
        if (nk == 1) SGAM_TSDF_INTEGRATE(1, false);
        else if (nk == 2) SGAM_TSDF_INTEGRATE(2, false);
        else if (nk == 3) SGAM_TSDF_INTEGRATE(3, false);
        else if (nk == 5) SGAM_TSDF_INTEGRATE(5, false);
        else SGAM_TSDF_INTEGRATE(8, false);
    }
#undef SGAM_TSDF_INTEGRATE
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_tsdf_ray_mult_f32(int32_t H, int32_t W, float fx, float fy, float cx, float cy, float *out, void *stream) {
    if (H <= 0 || W <= 0 || !(fx > 0.f) || !(fy > 0.f) || !out) return SGAM_EINVAL;
    SGAM_KLAUNCH(tsdf_ray_mult_kernel, dim3(sgam_cdiv((int64_t)H * W, 256)), dim3(256), 0, sgam_stream(stream), H, W, fx, fy, cx, cy, out);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_tsdf_raycast_depth_f32(const sgam_tsdf_grid *grid, int32_t H, int32_t W, float fx, float fy, float cx,
                                           float cy, const float *cam2world, float z_near, float z_far, const int32_t *unit_table, const float *brick_tsdf,
                                           float *depth_out, const float *brick_color, float *color_out, void *stream) {
    if ((brick_color == nullptr) != (color_out == nullptr)) return SGAM_EINVAL;
    if (!grid_ok(grid) || !(fx > 0.f) || !(fy > 0.f) || !cam2world || !unit_table || !brick_tsdf || !depth_out || H <= 0 || W <= 0 ||
        !(z_near > 0.f) || !(z_far > z_near))
        return SGAM_EINVAL;
    const TsdfGrid g = to_dev(grid);
    Pose c2w;
    for (int i = 0; i < 16; ++i) c2w.m[i] = cam2world[i];
    SGAM_KLAUNCH(tsdf_raycast_kernel, dim3(sgam_cdiv(W, 8) * sgam_cdiv(H, 8)), dim3(64 * RS), 0, sgam_stream(stream), H, W, fx, fy,
                       cx, cy, c2w, g, z_near, z_far, unit_table, brick_tsdf, depth_out, brick_color, color_out);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_tsdf_extract_points_f32(const sgam_tsdf_grid *grid, const int32_t *unit_table, const float *brick_tsdf,
                                            const float *brick_color, uint64_t *counter, int64_t max_points, float *points,
                                            float *normals, float *colors, int64_t *keys, void *stream) {
    if (!grid_ok(grid) || !unit_table || !brick_tsdf || !counter || max_points < 0) return SGAM_EINVAL;
    if (!points && (normals || colors || keys)) return SGAM_EINVAL;           // counting pass: no outputs at all
    if (colors && !brick_color) return SGAM_EINVAL;
    const TsdfGrid g = to_dev(grid);
    const int64_t n_units = (int64_t)g.dims[0] * g.dims[1] * g.dims[2];
    const int blocks = (int)(n_units < 4096 ? n_units : 4096);
    SGAM_KLAUNCH(tsdf_extract_kernel, dim3(blocks), dim3(256), 0, sgam_stream(stream), g, n_units, unit_table, brick_tsdf, brick_color,
                 (unsigned long long *)counter, max_points, points, normals, colors, keys);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}
