// Fused single-head spatial self-attention of the VQGAN AttnBlock on the default fp32 path (gfx950 only).
//
//   o = softmax(q k^T * C^-1/2) v        q, k, v : [n tokens][C = 256] fp32 column slices of the fused q|k|v projection
//
// replaces `torch.bmm(q, k)`, `w_ * c**-0.5`, `softmax(dim=2)`, `torch.bmm(v, w_)` of the reference
// (modules/diffusionmodules/model.py:176-187) in ONE pass over the keys: the n x n score matrix (64 MB at n = 4096,
// written and read four times by the GEMM -> softmax -> GEMM chain) never exists.  Arithmetic as everywhere on this
// path: every fp32 operand is split exactly into fp16 hi + lo, a product is three v_mfma_f32_32x32x16_f16 with fp32
// accumulation; the soft-max runs in fp32 with a running maximum (the textbook online form: the result is the same
// quotient sum_j e^{s_j - m} v_j / sum_j e^{s_j - m}, rounded in a different order).
//
// Orientation.  A wavefront owns 32 queries and computes the TRANSPOSED score tile S^T = K Q^T (keys are MFMA rows,
// queries are MFMA columns): in the 32x32 accumulator layout a lane then holds 16 keys of ONE query, so the row
// statistics of the soft-max are per-lane register reductions plus one exchange with lane ^ 32, and the probabilities
// are already where the B operand of the second product O^T = V^T P^T wants them (lane = query column, 8 consecutive
// k slots) — no cross-lane traffic between the two products.  The k-slot <-> key permutation this implies,
//     key(t, h, s) = 4h + 8 (2t + s/4) + s%4        (k-step t of 16 keys, lane half h, slot s of 8),
// is baked into the V^T fragments by attn_split_kv_kernel.
//
// Work split.  Workgroup = 4 wavefronts = 128 queries sharing every K / V block through LDS (double buffered, 32 keys per
// block: 32 KB of K fragments + 32 KB of V^T fragments, already in MFMA-fragment order, so staging is a linear copy —
// done by LDS-DMA, global -> LDS without a register stop — and every ds_read_b128 of a fragment is one conflict-free
// kilobyte).  Per wavefront: the 32 x 256 query panel in registers as B fragments (128 VGPRs), the 256 x 32 output
// accumulators (128 AccVGPRs), two score accumulators the 48 S MFMAs alternate between — 372 registers, one wavefront
// per SIMD.  n / 128 query blocks do not fill 256 CUs, so the keys are cut into NSPLIT = 8 ranges (flash-decoding
// style): grid = 8 x n/128, range s on XCD s (its K / V slice stays in that XCD's L2), each workgroup leaves its
// un-normalised O, running maximum and sum; attn_combine_kernel merges the ranges.
//
// Measured (n = 4096, one MI355X, round 3): split 2.4 us + this kernel 52 us + combine 8 us (70 us for the three launches back
// to back) against 141 us for the GEMM / softmax / GEMM chain.  rocprofv3 counters of the 52 us (profiles/
// r03_pmc_attn_flash_f32x.json): a wavefront lives 90 k of the kernel's 122 k cycles (the rest is the launch and the write-back
// of 32 MB of partial O behind the last wavefront), 49 k of them are its 1 584 MFMAs — the loop runs at 0.82 of the matrix
// pipe, the kernel at 0.42.  (Rounds 1 - 2: 64 us; 6 100 of a wavefront's 10 860 VALU instructions were accumulator
// rescales, see RESCALE_TAU.)
#include "sgam_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int AD = 256;            // head dimension (= channels of the AttnBlock)
constexpr int KB = 32;             // keys per LDS block
constexpr int BLK_BYTES = KB * AD * 4;   // one block of K (or V^T) fragments: hi + lo halves = 32 KB
constexpr float P_LIFT = 10.0f;    // probabilities are lifted by 2^10 before the fp16 split (keeps the lo half normal)
constexpr float LOG2E = 1.4426950408889634f;
// Lazy rescale of the output accumulators: the exponent reference of a query (its "running maximum") only follows the true
// maximum when some query of the wavefront has outgrown it by more than 2^5.  Until then probabilities may exceed 1 — by at
// most 2^5, which the fp32 sums do not notice, and 2^(5 + P_LIFT) = 2^15 still fits the fp16 hi half of the split.  (The
// textbook form rescales whenever ANY of a wavefront's 32 queries sees a new maximum: with 16 key blocks per range that is
// nearly every trip, and one rescale is 384 VALU instructions on the 128 accumulator registers — as long as half a trip's
// MFMAs on the split-fp32 kernel, 1.5 x the trip's MFMAs on the 16-bit one.  rocprofv3: 10 860 VALU instructions per
// wavefront, 6 100 of them rescales.)  The result is the same quotient: numerator and denominator share the reference, and
// the merge of the key ranges reads the reference that was used.
constexpr float RESCALE_TAU = 5.0f;

__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// exact hi / lo fp16 split of 8 fp32 values -> one MFMA operand register quad each
__device__ __forceinline__ void split8(const float *v, u32x4 &hi, u32x4 &lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 h0 = (_Float16)v[2 * e], h1 = (_Float16)v[2 * e + 1];
        const _Float16 l0 = (_Float16)(v[2 * e] - (float)h0), l1 = (_Float16)(v[2 * e + 1] - (float)h1);
        hi[e] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
        lo[e] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k, v slices -> MFMA-fragment order.  Both outputs are [key block of 32][8 tiles][256 pieces][8 halfs] with
// piece = ((plane * 2 + k-step) * 2 + lane half) * 32 + row:
//   K fragments (A operand of S^T = K Q^T): row = key in the block, tile = 32-wide slab of d, the 8 halfs of (k-step t,
//     half h) are d = 32 tile + 16 t + 8 h + 0..7 — the layout sgam_split_rows_f32x produces;
//   V^T fragments (A operand of O^T = V^T P^T): row = d in the tile, the 8 halfs of (t, h) are the keys key(t, h, 0..7).
// One thread per (key block, tile, k-step, half, row): 8 values, hi and lo pieces, of both tensors.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_split_kv_kernel(const float *__restrict__ k, const float *__restrict__ v, int ld,
                                                            int n, unsigned short *__restrict__ kf,
                                                            unsigned short *__restrict__ vf) {
    const int gid = blockIdx.x * 256 + threadIdx.x;            // (key block, tile) * 128 + (t * 2 + h) * 32 + row
    const int row = gid & 31, h = (gid >> 5) & 1, t = (gid >> 6) & 1, tile = (gid >> 7) & 7, kb = gid >> 10;
    if (kb * KB >= n) return;
    float kv[8], vv[8];
    const float *kp = k + (int64_t)(kb * KB + row) * ld + tile * 32 + t * 16 + h * 8;
    const f32x4 k0 = *reinterpret_cast<const f32x4 *>(kp), k1 = *reinterpret_cast<const f32x4 *>(kp + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        kv[e] = k0[e];
        kv[4 + e] = k1[e];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int key = 4 * h + 8 * (2 * t + (s >> 2)) + (s & 3);
        vv[s] = v[(int64_t)(kb * KB + key) * ld + tile * 32 + row];
    }
    u32x4 hi, lo;
    const int64_t piece = ((int64_t)(kb * 8 + tile) * 256 + (t * 2 + h) * 32 + row) * 8;      // halfs; lo plane: +128 pieces
    split8(kv, hi, lo);
    *reinterpret_cast<u32x4 *>(kf + piece) = hi;
    *reinterpret_cast<u32x4 *>(kf + piece + 128 * 8) = lo;
    split8(vv, hi, lo);
    *reinterpret_cast<u32x4 *>(vf + piece) = hi;
    *reinterpret_cast<u32x4 *>(vf + piece + 128 * 8) = lo;
}

struct AttnParams {
    const unsigned short *qf;       // optional: Q already scaled, split and in B-fragment order (attn_qkv_gn_f32x_kernel):
                                    //   [n / 32][16 k-steps][hi | lo][64 lanes][8 halfs]; NULL: q below is read and split here
    const float *q;                 // [n][ld] fp32, AD columns
    const unsigned short *kf, *vf;  // fragment-ordered K and V^T
    float *ws_o;                    // [NSPLIT][n / 32][AD][32]   un-normalised O^T per (split, 32-query tile)
    float *ws_ml;                   // [NSPLIT][n][2]             running maximum (log2 domain), sum
    int ld, n, blocks_per_split;    // n = tokens of the whole batch (B images of n_img tokens, stacked along the rows)
    int n_img, nsplit;              // tokens per image (keys are attended within an image); key ranges per image (1, 2, 4, 8)
    float qscale;                   // C^-1/2 (a power of two for C = 256: folding it into q is exact)
};

constexpr int NSPLIT = 8;          // key ranges = XCDs
#ifndef SGAM_ATTN_ABLATE
#define SGAM_ATTN_ABLATE 0         // timing experiments only (results are wrong when != 0), bit mask: 1 no staging in the
#endif                             // loop, 2 no soft-max arithmetic, 4 no S MFMAs, 8 no PV MFMAs, 16 no loop at all, 32 no partial-O stores

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

__global__ __launch_bounds__(256) void attn_flash_f32x_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * BLK_BYTES];   // K buffers 0, 1 | V^T buffers 0, 1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sp = blockIdx.x % p.nsplit, qb = blockIdx.x / p.nsplit;
    const int q0 = qb * 128 + wave * 32;
    const int lq = lane & 31, lh = lane >> 5;
    // the keys of THIS query block's image: key blocks [img * n_img / 32, (img + 1) * n_img / 32), cut into nsplit ranges
    const int kb0 = ((qb * 128) / p.n_img) * (p.n_img / KB) + sp * p.blocks_per_split, kb1 = kb0 + p.blocks_per_split;
    const unsigned char *kg = reinterpret_cast<const unsigned char *>(p.kf), *vg = reinterpret_cast<const unsigned char *>(p.vf);

    // a block is already in fragment order: staging = a linear 32 KB copy, done by the LDS-DMA path (global -> LDS without
    // a register stop): each wavefront moves 8 KB as eight 1 KB pieces (lane l -> piece base + 16 l)
    // piece i of a wavefront's 8 KB share: the instruction offset moves the global address and the LDS address alike, so
    // one (address, M0) setup serves four pieces; the wavefront index is made scalar so that the LDS base stays in SGPRs
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    auto dma1 = [&](const unsigned char *g, int kb, int slot, const int i) {
        const unsigned char *src = g + (int64_t)kb * BLK_BYTES + wave_s * 8192 + (i >> 2) * 4096 + lane * 16;
        unsigned char *dst = smem + slot * BLK_BYTES + wave_s * 8192 + (i >> 2) * 4096;
        switch (i & 3) {
        case 0: __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)dst, 16, 0, 0); break;
        case 1: __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)dst, 16, 1024, 0); break;
        case 2: __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)dst, 16, 2048, 0); break;
        default: __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)dst, 16, 3072, 0); break;
        }
    };
    auto dma = [&](const unsigned char *g, int kb, int slot) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dma1(g, kb, slot, i);
    };
    // ---- query panel -> B fragments (registers): k-step t covers d = 16 t + 8 h + 0..7 of query q0 + lq
    u32x4 qh[16], ql[16];
    if (p.qf) {
        // (round 5) the fused front end left the panel as fragments: 32 coalesced kilobyte loads instead of 32 x 64 scattered 16-byte
        // pieces (one per (query row, half) — the CU's address path again, §5.5e) and 16 operand splits
        const unsigned short *qb = p.qf + (int64_t)(q0 >> 5) * (16 * 1024) + lane * 8;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            qh[t] = *reinterpret_cast<const u32x4 *>(qb + t * 1024);
            ql[t] = *reinterpret_cast<const u32x4 *>(qb + t * 1024 + 512);
        }
    } else {
        const float *qp = p.q + (int64_t)(q0 + lq) * p.ld + lh * 8;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(qp + t * 16), b = *reinterpret_cast<const f32x4 *>(qp + t * 16 + 4);
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = a[e] * p.qscale;
                v[4 + e] = b[e] * p.qscale;
            }
            split8(v, qh[t], ql[t]);
        }
    }

    f32x16 o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;      // per lane: maximum of the whole query (both halves agree), sum of OWN keys

    // Software pipeline over the key blocks (one wavefront per SIMD: nothing else hides the soft-max arithmetic, so it is
    // threaded through the MFMA stream by hand).  Trip j of the loop runs
    //     S(j+1) MFMAs   with   exp / sum / fp16 split of block j        (VALU in the shadow of the matrix pipe)
    //     PV(j)  MFMAs   with   scores -> log2 domain + maximum of block j+1
    // and ends with the (rare) rescale of O when the running maximum moved.
    //
    // Staging runs a whole trip ahead of its use.  A block takes 1 - 2 us from L2 / HBM under load and a trip lasts ~1.3 us,
    // so a block requested during the phase before the one that needs it (the round-1 / round-2 order, one barrier per trip
    // draining the whole queue) arrived late on every trip.  Now there are TWO barriers per trip, each behind a COUNTED
    // vmcnt (LDS-DMA pieces retire in issue order):
    //     S phase of trip j    issues V^T(j+1) -> the buffer PV(j-1) left at the end of trip j-1
    //     mid barrier          waits for V^T(j) (issued a trip ago; K(j+2) and V^T(j+1) = 16 pieces may still fly);
    //                          every wavefront is done with K(j+1)
    //     PV phase of trip j   issues K(j+3)   -> the buffer of K(j+1)
    //     end barrier          waits for K(j+2) (issued a trip ago; V^T(j+1), K(j+3) may still fly); all done with V^T(j)
    const int nb = kb1 - kb0;
    auto blk = [&](int j) { return kb0 + (j < nb ? j : nb - 1); };       // past the end: a valid block, result unused
    dma(kg, blk(0), 0);
    dma(vg, blk(0), 2);
    dma(kg, blk(1), 1);

    // MFMA A fragments come out of LDS through explicit ds_read_b128 statements: three rotating register sets, issued
    // two steps ahead and retired by counted waits.  (Left to the compiler the reads are cloned and hoisted until half of
    // the query panel is pushed out to AccVGPRs and copied back before every MFMA.)
    u32x4 fh[3], fl[3];
    f32x16 sacc[2];
    float s[16], pe[2];          // s: log2-domain scores of the block whose soft-max comes next
    u32x4 ph[2], pl[2];
    auto lds_off = [](const unsigned char *ptr) {
        return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)ptr;
    };
#define ATTN_DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
    // reads retired so far = all but the newest `n`; ties the fragment registers to the wait so that no MFMA moves above it
#define ATTN_DS_WAIT(n, set) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(fh[set]), "+v"(fl[set]))
    // this wavefront's DMA pieces have landed except the newest 16; then the workgroup meets (no fence: the LDS traffic of
    // this loop is asm / DMA, ordered by hand)
#define ATTN_BARRIER(cnt)                                        \
    do {                                                          \
        asm volatile("s_waitcnt vmcnt(" #cnt ")" ::: "memory");   \
        __builtin_amdgcn_s_barrier();                             \
    } while (0)
    auto smfma = [&](const int t) {            // the three product terms of k-step t, alternating accumulators
        sacc[(3 * t) & 1] = mfma16(fh[t % 3], qh[t], sacc[(3 * t) & 1]);
        sacc[(3 * t + 1) & 1] = mfma16(fh[t % 3], ql[t], sacc[(3 * t + 1) & 1]);
        sacc[(3 * t + 2) & 1] = mfma16(fl[t % 3], qh[t], sacc[(3 * t + 2) & 1]);
    };
    // fragment (hi, lo) of step `t` inside a 32 KB block -> register set: step t sits at (t / 2) * 4096 + (t % 2) * 1024,
    // its lo plane 2048 bytes further (same formula for the K block's k-steps and the V^T block's (tile, k-step) pairs)
#define ATTN_FRAG(set, base, t)                                              \
    do {                                                                      \
        ATTN_DS_READ(fh[set], base, ((t) >> 1) * 4096 + ((t) & 1) * 1024);        \
        ATTN_DS_READ(fl[set], base, ((t) >> 1) * 4096 + (2 + ((t) & 1)) * 1024);  \
    } while (0)
    ATTN_BARRIER(0);

    // ---- prologue: scores of the first block, their maximum
    {
        const unsigned lk = lds_off(smem + lane * 16);
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[0][e] = sacc[1][e] = 0.f;
        ATTN_FRAG(0, lk, 0);
        ATTN_FRAG(1, lk, 1);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t + 2 < 16) ATTN_FRAG((t + 2) % 3, lk, t + 2);
            if (t + 2 < 16) ATTN_DS_WAIT(4, t % 3);
            else if (t + 1 < 16) ATTN_DS_WAIT(2, t % 3);
            else ATTN_DS_WAIT(0, t % 3);
            smfma(t);
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = (sacc[0][e] + sacc[1][e]) * LOG2E;
            mloc = fmaxf(mloc, s[e]);
        }
        m_run = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        ATTN_BARRIER(0);                 // every wavefront has read K(0): its buffer takes K(2) (the PV phase of "trip -1")
        dma(kg, blk(2), 0);
    }

    for (int j = 0; j < ((SGAM_ATTN_ABLATE & 16) ? 0 : nb); ++j) {
        const int buf = j & 1;
        const unsigned lk = lds_off(smem + (buf ^ 1) * BLK_BYTES + lane * 16);       // K(j+1)
        const unsigned lv = lds_off(smem + (2 + buf) * BLK_BYTES + lane * 16);       // V^T(j)
        const int kb_k = blk(j + 3), kb_v = blk(j + 1);
        // ---- S(j+1) MFMAs + soft-max arithmetic of block j
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[0][e] = sacc[1][e] = 0.f;
        ATTN_FRAG(0, lk, 0);
        ATTN_FRAG(1, lk, 1);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t + 2 < 16) ATTN_FRAG((t + 2) % 3, lk, t + 2);
            if (t < 8 && !(SGAM_ATTN_ABLATE & 1)) dma1(vg, kb_v, 2 + (buf ^ 1), t);
            if (t + 2 < 16) ATTN_DS_WAIT(4, t % 3);
            else if (t + 1 < 16) ATTN_DS_WAIT(2, t % 3);
            else ATTN_DS_WAIT(0, t % 3);
            if (!(SGAM_ATTN_ABLATE & 4)) smfma(t);
            if (!(SGAM_ATTN_ABLATE & 2)) {
                // p = 2^(s - m + 10): the lift by 2^10 rides in the exponent (l_run carries it too); the fp16 split
                // truncates the hi half (v_cvt_pkrtz: two values per instruction) — hi + lo still holds 21 bits of p
                pe[t & 1] = __builtin_amdgcn_exp2f(s[t] - m_run + P_LIFT);
                l_run += pe[t & 1];
                if (t & 1) {
                    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                    const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(pe[0], pe[1]));
                    const f16x2 l = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(pe[0] - (float)h[0], pe[1] - (float)h[1]));
                    ph[t >> 3][(t & 7) >> 1] = __builtin_bit_cast(unsigned, h);
                    pl[t >> 3][(t & 7) >> 1] = __builtin_bit_cast(unsigned, l);
                }
            } else if (t & 1) {
                ph[t >> 3][(t & 7) >> 1] = pl[t >> 3][(t & 7) >> 1] = __builtin_bit_cast(unsigned, s[t]);
            }
        }
        if (!(SGAM_ATTN_ABLATE & 1)) ATTN_BARRIER(16);
        // ---- PV(j) MFMAs + log2-domain scores and maximum of block j+1.  V^T step u uses set (16 + u) % 3.
        float mloc = -INFINITY;
        ATTN_FRAG(16 % 3, lv, 0);
        ATTN_FRAG(17 % 3, lv, 1);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (u + 2 < 16) ATTN_FRAG((16 + u + 2) % 3, lv, u + 2);
            if (u < 8 && !(SGAM_ATTN_ABLATE & 1)) dma1(kg, kb_k, buf ^ 1, u);
            if (u + 2 < 16) ATTN_DS_WAIT(4, (16 + u) % 3);
            else if (u + 1 < 16) ATTN_DS_WAIT(2, (16 + u) % 3);
            else ATTN_DS_WAIT(0, (16 + u) % 3);
            const int i = u >> 1, t = u & 1, set = (16 + u) % 3;
            if (!(SGAM_ATTN_ABLATE & 8)) {
                o[i] = mfma16(fh[set], ph[t], o[i]);
                o[i] = mfma16(fh[set], pl[t], o[i]);
                o[i] = mfma16(fl[set], ph[t], o[i]);
            }
            if (u >= 4) {                              // (the last S MFMAs have left the pipe by now)
#pragma unroll
                for (int e = (u - 4) * 4 / 3; e < (u - 3) * 4 / 3; ++e) {
                    s[e] = (sacc[0][e] + sacc[1][e]) * LOG2E;          // (slot e was consumed in S step e of this trip)
                    mloc = fmaxf(mloc, s[e]);
                }
            }
        }
        if (j + 1 < nb && !(SGAM_ATTN_ABLATE & 2)) {
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float m_new = fmaxf(m_run, mloc);
            if (__any(m_new > m_run + RESCALE_TAU)) {     // wavefront-uniform, rare (see RESCALE_TAU)
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
                // the output accumulators live in AccVGPRs (MFMA C/D); scale them in place, register by register, so
                // that the allocator keeps them there instead of shuttling all 128 through VGPRs on every trip.
                // (Hazards the compiler cannot see inside the asm: the statements run in tile order, so the tile the
                // last PV MFMA wrote is read > 300 instructions after that MFMA issued, and the next MFMA that takes one
                // of these registers as SrcC is a whole S phase away.)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float x = o[i][e], tmp;
                        asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1"
                                     : "+a"(x), "=&v"(tmp)
                                     : "v"(alpha));
                        o[i][e] = x;
                    }
                m_run = m_new;
            }
        }
        if (!(SGAM_ATTN_ABLATE & 1)) ATTN_BARRIER(16);
    }
#undef ATTN_FRAG
#undef ATTN_BARRIER
#undef ATTN_DS_WAIT
#undef ATTN_DS_READ

    // ---- partial result of this key range
    l_run += __shfl_xor(l_run, 32, 64);
    if (lh == 0) {
        float *ml = p.ws_ml + ((int64_t)sp * p.n + q0 + lq) * 2;
        ml[0] = m_run;
        ml[1] = l_run;
    }
    float *wo = p.ws_o + ((int64_t)sp * (p.n / 32) + (q0 >> 5)) * (AD * 32) + lq;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (!(SGAM_ATTN_ABLATE & 32) || o[i][e] == 12345.678f) wo[(32 * i + 8 * (e >> 2) + 4 * lh + (e & 3)) * 32] = o[i][e];
    // the blocks requested past the end of the range (same piece count on every trip keeps the waits countable) must have
    // landed before this workgroup's LDS is handed to the next one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// merge the key ranges: o[q][d] = sum_s w_s O_s[d][q] / sum_s w_s l_s,  w_s = 2^(m_s - max_s m_s).
// Workgroup = (32-query tile, 32 columns of d); thread = (query, 4 d): 32 independent loads in flight per thread; the
// 32 x 32 result goes through LDS so that rows leave as 128-byte pieces.
// Round 5: the number of ranges is a template parameter.  With `ns` a run-time bound every load of the merge sat behind its own
// `s < ns` branch and its own s_waitcnt (the listing: global_load_dword / s_waitcnt vmcnt(0), forty-eight times per thread — a
// chain of dependent round trips that only the 1024 workgroups' parallelism kept at 8 us); now a thread's 2 NS statistics and
// 4 NS partial values are each requested together.  Same arithmetic, same order.
template <int NS>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float *__restrict__ ws_o, const float *__restrict__ ws_ml,
                                                           float *__restrict__ out, int ldo, int n, int32_t *range_flag) {
    __shared__ float tile[32][33];
    const int qt = blockIdx.x >> 3, dg = blockIdx.x & 7;
    const int q = threadIdx.x & 31, dsub = threadIdx.x >> 5;
    float w[NS], l[NS], o[NS][4], M = -INFINITY, L = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float *ml = ws_ml + ((int64_t)s * n + qt * 32 + q) * 2;
        w[s] = ml[0];
        l[s] = ml[1];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[s][j] = ws_o[(((int64_t)s * (n / 32) + qt) * AD + dg * 32 + dsub * 4 + j) * 32 + q];
#pragma unroll
    for (int s = 0; s < NS; ++s) M = fmaxf(M, w[s]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        w[s] = __builtin_amdgcn_exp2f(w[s] - M);
        L += w[s] * l[s];
    }
    const float inv = 1.0f / L;            // O and l both carry the 2^10 lift of the probabilities
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += w[s] * o[s][j];
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[q][dsub * 4 + j] = acc[j] * inv;
    __syncthreads();
    const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    const f32x4 v = {tile[r][c4], tile[r][c4 + 1], tile[r][c4 + 2], tile[r][c4 + 3]};
    *reinterpret_cast<f32x4 *>(out + (int64_t)(qt * 32 + r) * ldo + dg * 32 + c4) = v;
    if (range_flag && sgam_not_finite((v[0] + v[1]) + (v[2] + v[3]))) atomicOr(range_flag, 1);   // q, k or v left fp16's range
}

// ---------------------------------------------------------------------------------------------------------------------
// attn_combine + AttnBlock.proj_out (+ residual) in ONE launch (round 5; reference modules/diffusionmodules/model.py:187-191:
// h_ = self.proj_out(h_); return x + h_).  The merge of the key ranges writes [n][C] (4 MB at n = 4096) that the 1x1 convolution
// reads back one dependent launch later: two fixed costs of a B = 1 frame (a ~1.6 us graph edge + a kernel that is a chain of
// round trips) for 0.5 GFLOP.  Here a workgroup owns one 32-query tile: it merges the ranges of its tile exactly as
// attn_combine_kernel does (same weights, same order of the fused multiply-adds, same final product with 1 / L), splits the
// result into fp16 hi / lo halves on its way into LDS — the A panel of the projection, 32 x 256 — and multiplies it with the
// fragment-ordered weights of proj_out (four wavefronts side by side, 64 output channels each; weights four k-steps ahead), adds
// bias and residual and leaves the GroupNorm statistics of the block output (one chunk per 32-row tile).
// ---------------------------------------------------------------------------------------------------------------------
struct CombProjParams {
    const float *ws_o, *ws_ml;      // partial O^T [ns][n / 32][AD][32] and {max, sum} [ns][n][2] of the flash kernel
    const unsigned short *w;        // proj_out weights, fragment-ordered hi / lo planes [AD / 32][AD / 32][256 pieces][8 halfs]
    const float *bias, *res;        // [AD]; residual [n][ldr] (the block's input x) or NULL
    float *out;                     // [n][ldc]
    double *gn_partial;             // optional [n / 32][32][2]: {sum, sumsq} of the output per (32-row tile, group of 8 channels)
    int n_img;                      // tokens per image
    int32_t *range_flag;
    int n, ns, ldr, ldc;
    float inv_w_scale;
};

template <int NS>
__global__ __launch_bounds__(256) void attn_combine_proj_f32x_kernel(const CombProjParams p) {
    constexpr int LDK = AD + 8;                                      // LDS row pitch in halfs (rows 4 banks apart, 16-byte aligned)
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][32][LDK];       // hi plane, lo plane of the merged 32 x 256 tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qt = blockIdx.x;
    const int lr = lane & 31, lh = lane >> 5;
    constexpr int WD = 4, KS = AD / 16, SLABS = AD / 32;
    // weight fragments of (32-column tile nt, k-step t): pieces ((plane * 2 + t % 2) * 2 + lh) * 32 + lr of slab t / 2
    const unsigned short *wt0 = p.w + (int64_t)(wave * 2) * SLABS * 2048, *wt1 = wt0 + (int64_t)SLABS * 2048;
    u32x4 wh[WD][2], wl[WD][2];
    auto wfrag = [&](int kstep, u32x4 (&hi)[2], u32x4 (&lo)[2]) {
        const int off = (kstep >> 1) * 2048 + (((kstep & 1) * 2 + lh) * 32 + lr) * 8;
        hi[0] = *reinterpret_cast<const u32x4 *>(wt0 + off);
        lo[0] = *reinterpret_cast<const u32x4 *>(wt0 + off + 1024);
        hi[1] = *reinterpret_cast<const u32x4 *>(wt1 + off);
        lo[1] = *reinterpret_cast<const u32x4 *>(wt1 + off + 1024);
    };
    // ---- merge of the key ranges: thread = (query q, channel group dg of 32): weights as attn_combine_kernel
    const int q = tid & 31, dg = tid >> 5;
    float w[NS], l[NS], M = -INFINITY, L = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        w[s] = p.ws_ml[((int64_t)s * p.n + qt * 32 + q) * 2];
        l[s] = p.ws_ml[((int64_t)s * p.n + qt * 32 + q) * 2 + 1];
    }
    const float *ob = p.ws_o + ((int64_t)qt * AD + dg * 32) * 32 + q;          // + s * n * AD, + d * 32
    const int64_t sstride = (int64_t)p.n * AD;
    // CPS channels per step with ALL ranges' loads of the step in flight (CPS NS registers: 128 at eight ranges), two steps = two trips
    // to memory per thread.  (128 workgroups merge what attn_combine_kernel spreads over 1024: the memory-level parallelism has to
    // come from loads in flight per thread — four channels per step with the next step requested ahead measured 21.8 us for this
    // kernel, eight 17.4.)
    constexpr int CPS = NS >= 8 ? 16 : 32, NSTEP = 32 / CPS;
    for (int s = 0; s < NS; ++s) M = fmaxf(M, w[s]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        w[s] = __builtin_amdgcn_exp2f(w[s] - M);
        L += w[s] * l[s];
    }
    const float inv = 1.0f / L;
    bool bad = false;
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
        float cur[CPS][NS];
#pragma unroll
        for (int j = 0; j < CPS; ++j)
#pragma unroll
            for (int s = 0; s < NS; ++s) cur[j][s] = ob[s * sstride + (step * CPS + j) * 32];
        float a[CPS];
#pragma unroll
        for (int j = 0; j < CPS; ++j) a[j] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < CPS; ++j) a[j] += w[s] * cur[j][s];
#pragma unroll
        for (int h4 = 0; h4 < CPS / 4; ++h4) {
            unsigned hi[2], lo[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float a0 = a[4 * h4 + 2 * e] * inv, a1 = a[4 * h4 + 2 * e + 1] * inv;
                bad |= sgam_not_finite(a0 + a1);
                const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
                const _Float16 l0 = (_Float16)(a0 - (float)h0), l1 = (_Float16)(a1 - (float)h1);
                hi[e] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                lo[e] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
            }
            const int d = dg * 32 + step * CPS + 4 * h4;
            *reinterpret_cast<u32x2 *>(&sA[0][q][d]) = u32x2{hi[0], hi[1]};
            *reinterpret_cast<u32x2 *>(&sA[1][q][d]) = u32x2{lo[0], lo[1]};
        }
    }
    if (bad && p.range_flag) atomicOr(p.range_flag, 1);                // q, k or v left fp16's range (what attn_combine_kernel reports)
#pragma unroll
    for (int t = 0; t < WD; ++t) wfrag(t, wh[t], wl[t]);               // (behind the merge: its registers are the merge's loads in flight)
    // the residual rows and the bias of this lane's outputs, requested under the projection's MFMAs instead of in the epilogue
    const int m0 = qt * 32;
    float rres[2][16], rbias[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nn = wave * 64 + j * 32 + lr;
        rbias[j] = p.bias ? p.bias[nn] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) rres[j][e] = p.res ? p.res[(int64_t)(m0 + 8 * (e >> 2) + 4 * lh + (e & 3)) * p.ldr + nn] : 0.f;
    }
    __syncthreads();
    // ---- out tile = merged . Wp^T: 16 k-steps, A fragments from LDS (row lr, k = 16 t + 8 lh + 0..7), weights WD steps ahead
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
        u32x4 bh[2] = {wh[t % WD][0], wh[t % WD][1]}, bl[2] = {wl[t % WD][0], wl[t % WD][1]};
        wfrag(t + WD < KS ? t + WD : KS - 1, wh[t % WD], wl[t % WD]);
        __builtin_amdgcn_sched_barrier(0);                                 // (keeps the request in front of this step's MFMAs: gemm_gn_f32x.hip)
        const u32x4 ah = *reinterpret_cast<const u32x4 *>(&sA[0][lr][t * 16 + lh * 8]);
        const u32x4 al = *reinterpret_cast<const u32x4 *>(&sA[1][lr][t * 16 + lh * 8]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[j] = mfma16(ah, bh[j], acc[j]);
            acc[j] = mfma16(ah, bl[j], acc[j]);
            acc[j] = mfma16(al, bh[j], acc[j]);
        }
    }
    // ---- epilogue: lane = column, rows 8 (e / 4) + 4 lh + e % 4 of the 32-row tile; a half-wave writes 128 B of a row
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nn = wave * 64 + j * 32 + lr;
        const float bias = rbias[j];
        float gs = 0.f, gss = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = m0 + 8 * (e >> 2) + 4 * lh + (e & 3);
            float v = acc[j][e] * p.inv_w_scale + bias;
            v += rres[j][e];
            p.out[(int64_t)row * p.ldc + nn] = v;
            gs += v;
            gss += v * v;
        }
        if (p.range_flag && sgam_not_finite(gs)) atomicOr(p.range_flag, 1);
        if (p.gn_partial) {
            constexpr int CPO = AD / 32;                               // 8 adjacent columns are one group of the next GroupNorm
            double ds = (double)gs, dss = (double)gss;
#pragma unroll
            for (int o = 1; o < CPO; o <<= 1) {
                ds += __shfl_xor(ds, o, 64);
                dss += __shfl_xor(dss, o, 64);
            }
            ds += __shfl_xor(ds, 32, 64);
            dss += __shfl_xor(dss, 32, 64);
            if (lh == 0 && (lr % CPO) == 0) {
                double *o = p.gn_partial + ((int64_t)qt * 32 + nn / CPO) * 2;
                o[0] = ds;
                o[1] = dss;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Front end of the AttnBlock on the split-fp32 path, ONE launch (round 5): GroupNorm applied while the 64 x 256 operand panel is
// staged + the stacked q | k | v projection — the arithmetic of gemm_gn_f32x_kernel<true, 256>, operation for operation (same
// normalisation expression, the same three MFMAs per product in the same order, the same un-scale + bias) — computed TRANSPOSED
// (weights = MFMA rows, tokens = columns, MFMA row 8 j + 4 h + i of a 32-channel tile carrying channel 16 h + 4 j + i) so that a lane
// holds 16 consecutive channels of one token: q leaves row-major, K as the flash kernel's hi / lo fragment pieces (512 contiguous
// bytes per 32 lanes), V^T through an fp32 LDS transpose.  attn_split_kv_kernel and its graph edge disappear; results equal (fp32 round-off: measured, not bit-identical — the transposed MFMA chain rounds differently)
// to the gemm_gn_f32x + split sequence (tests/test_gpu_ops.py).
// ---------------------------------------------------------------------------------------------------------------------
struct QkvXParams {
    const float *x;                 // [nt][ldx] fp32 block input
    const float *mean_rstd;         // [B][32][2]
    const float *gamma, *beta;      // [AD]
    const unsigned short *w;        // split_rows planes of the ROW-PERMUTED stacked weight: [3 AD / 32][AD / 32][256 pieces][8 halfs]
    const float *bias;              // [3 AD] (natural channel order)
    unsigned short *qf;             // Q scaled by `qscale`, split, in the flash kernel's B-fragment order [nt / 32][16][hi | lo][64][8]
    unsigned short *kf, *vf;        // attn_split_kv_kernel's layout
    int32_t *range_flag;
    int ldx, n_img;
    float inv_w_scale, qscale;
};

__global__ __launch_bounds__(256, 2) void attn_qkv_gn_f32x_kernel(const QkvXParams p) {
    constexpr int LDK = AD + 8;                                       // panel row pitch in halfs
    constexpr int VLD = 64 + 4;                                       // V transpose: [128 channels][64 tokens + pad] fp32
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][64][LDK];              // hi plane, lo plane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 64, by = blockIdx.y;
    const int lr = lane & 31, lh = lane >> 5;
    const int b = m0 / p.n_img;
    constexpr int cpg = AD / 32, WD = 4, KS = AD / 16, NIT = 64 * (AD / 4) / 256;
    const unsigned short *wt = p.w + (int64_t)(by * 4 + wave) * (AD / 32) * 2048;
    auto wfrag = [&](int kstep, u32x4 &hi, u32x4 &lo) {
        const unsigned short *q = wt + (int64_t)(kstep >> 1) * 2048 + (((kstep & 1) * 2 + lh) * 32 + lr) * 8;
        hi = *reinterpret_cast<const u32x4 *>(q);
        lo = *reinterpret_cast<const u32x4 *>(q + 1024);
    };
    u32x4 wh[WD], wl[WD];
#pragma unroll
    for (int t = 0; t < WD; ++t) wfrag(t, wh[t], wl[t]);
    // ---- stage: thread -> (row, 4 consecutive channels), 16 rounds of 4 rows (gemm_gn_f32x_kernel's map and arithmetic)
    {
        const int prow = tid / (AD / 4), pc4 = (tid % (AD / 4)) * 4;
        f32x4 xr[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) xr[it] = *reinterpret_cast<const f32x4 *>(p.x + (int64_t)(m0 + prow + it * 4) * p.ldx + pc4);
        const int g = pc4 / cpg;
        const float mean = p.mean_rstd[(b * 32 + g) * 2], rstd = p.mean_rstd[(b * 32 + g) * 2 + 1];
        const f32x4 ga = *reinterpret_cast<const f32x4 *>(p.gamma + pc4), be = *reinterpret_cast<const f32x4 *>(p.beta + pc4);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = prow + it * 4;
            f32x4 v = xr[it];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean) * rstd * ga[e] + be[e];
            unsigned hi[2], lo[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float a0 = v[2 * e], a1 = v[2 * e + 1];
                const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
                const _Float16 l0 = (_Float16)(a0 - (float)h0), l1 = (_Float16)(a1 - (float)h1);
                hi[e] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                lo[e] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
            }
            *reinterpret_cast<unsigned long long *>(&sA[0][row][pc4]) = (unsigned long long)hi[0] | ((unsigned long long)hi[1] << 32);
            *reinterpret_cast<unsigned long long *>(&sA[1][row][pc4]) = (unsigned long long)lo[0] | ((unsigned long long)lo[1] << 32);
        }
    }
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < KS; ++t) {
        const u32x4 ah = wh[t % WD], al = wl[t % WD];
        wfrag(t + WD < KS ? t + WD : KS - 1, wh[t % WD], wl[t % WD]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const u32x4 xh = *reinterpret_cast<const u32x4 *>(&sA[0][j * 32 + lr][t * 16 + lh * 8]);
            const u32x4 xl = *reinterpret_cast<const u32x4 *>(&sA[1][j * 32 + lr][t * 16 + lh * 8]);
            acc[j] = mfma16(ah, xh, acc[j]);                          // x_hi . w_hi, x_hi . w_lo, x_lo . w_hi: gemm_gn_f32x's order
            acc[j] = mfma16(al, xh, acc[j]);
            acc[j] = mfma16(ah, xl, acc[j]);
        }
    }
    // ---- epilogue: lane (token lr of column tile j, half lh) holds channels cw + 0..15 of this 128-channel slab
    const int cw = wave * 32 + 16 * lh;
    float bias[16];
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(p.bias + by * 128 + cw + e4 * 4);
        bias[4 * e4] = bv[0], bias[4 * e4 + 1] = bv[1], bias[4 * e4 + 2] = bv[2], bias[4 * e4 + 3] = bv[3];
    }
    float v[2][16], gs = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            v[j][e] = acc[j][e] * p.inv_w_scale + bias[e];
            gs += v[j][e];
        }
    if (p.range_flag && sgam_not_finite(gs)) atomicOr(p.range_flag, 1);          // an operand left fp16's range
    if (by < 2) {                                                     // Q: scaled (a power of two: exact), split, as B fragments of k-step (by 128 + cw) / 16
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned short *dst = p.qf + ((int64_t)((m0 >> 5) + j) * 16 + by * 8 + wave * 2 + lh) * 1024 + lr * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float sv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) sv[e] = v[j][8 * h + e] * p.qscale;
                u32x4 hi, lo;
                split8(sv, hi, lo);
                *reinterpret_cast<u32x4 *>(dst + h * 32 * 8) = hi;
                *reinterpret_cast<u32x4 *>(dst + h * 32 * 8 + 512) = lo;
            }
        }
    } else if (by < 4) {                                              // K pieces (plane, k-step lh, half h, key lr) of d-tile (by - 2) 4 + wave
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t kb = (m0 >> 5) + j;
            unsigned short *dst = p.kf + ((kb * 8 + (by - 2) * 4 + wave) * 256 + (lh * 2) * 32 + lr) * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 hi, lo;
                split8(&v[j][8 * h], hi, lo);
                *reinterpret_cast<u32x4 *>(dst + h * 32 * 8) = hi;
                *reinterpret_cast<u32x4 *>(dst + h * 32 * 8 + 128 * 8) = lo;
            }
        }
    } else {                                                          // V^T pieces hold 8 KEYS of one channel: transpose in LDS (fp32)
        __syncthreads();
        float *vt = reinterpret_cast<float *>(&sA[0][0][0]);          // [128][VLD]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) vt[(cw + e) * VLD + j * 32 + lr] = v[j][e];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pp = tid + 256 * k;
            const int drow = pp & 31, th = (pp >> 5) & 3, tl = (pp >> 7) & 3, kbl = pp >> 9;
            const float *src = vt + (tl * 32 + drow) * VLD + kbl * 32 + 4 * (th & 1) + 16 * (th >> 1);
            const f32x4 a = *reinterpret_cast<const f32x4 *>(src), c = *reinterpret_cast<const f32x4 *>(src + 8);
            const float vv[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
            u32x4 hi, lo;
            split8(vv, hi, lo);
            const int64_t kb = (m0 >> 5) + kbl;
            unsigned short *dst = p.vf + ((kb * 8 + (by - 4) * 4 + tl) * 256 + th * 32 + drow) * 8;
            *reinterpret_cast<u32x4 *>(dst) = hi;
            *reinterpret_cast<u32x4 *>(dst + 128 * 8) = lo;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The SMALL AttnBlocks (the two 16 x 16 mid blocks: n = 256 tokens per image, C = 512) in ONE launch instead of seven (round 5).
// The chain they ran — v^T transpose, split of k, q k^T (split-K 2 + combine), row soft-max, split of v^T, P v — is a sequence of
// 2 - 10 us launches with a graph edge each for 0.2 GFLOP; nothing in it needs more than one image's 256 keys, so a workgroup can hold a
// query tile's whole score row.  Workgroup = (32 queries, 128 output channels); its four wavefronts take a quarter of the keys each:
//   S^T = K Q^T for its keys (keys = MFMA rows, so a lane holds 16 keys of ONE query, as in the flash kernel), operands split into
//         fp16 hi / lo on the fly — Q once per workgroup into LDS, K straight from L2 a k-step ahead;
//   soft-max over ALL keys of the image exactly as softmax_rows_kernel spells it (scale, maximum, expf, sum, one reciprocal): the
//         four wavefronts exchange maxima and sums through LDS;
//   O^T += V^T P^T for its keys and the workgroup's 128 channels (probabilities lifted by 2^10 before the split, as the chain's
//         second GEMM does; V^T fragments gathered with the flash kernel's key permutation so that a lane's accumulators ARE its
//         B operand), then the four partial outputs are added through LDS in a fixed order.
// The three MFMAs of every product keep the generic kernel's order (hi.hi, q_hi.k_lo / p_hi.v_lo, lo.hi); the summation trees differ
// from the chain's (no split-K in S, four key ranges in PV): equal to fp32 round-off, not bit for bit.
// ---------------------------------------------------------------------------------------------------------------------
struct SmallAttnParams {
    const float *q, *k, *v;         // [B n][ld] fp32 columns of the q | k | v projection
    float *out;                     // [B n][ldo]
    int ld, ldo, n_img;
    float scale;
    int32_t *range_flag;
};

template <int CD, int NKT, int NW>
__global__ __launch_bounds__(64 * NW) void attn_small_f32x_kernel(const SmallAttnParams p) {
    constexpr int LDQ = CD + 8;                                       // halfs
    constexpr int NT = 64 * NW;
    constexpr int QBYTES = 2 * 32 * LDQ * 2, PBYTES = NW * 4 * 16 * 64 * 4, KBYTES = 2 * (32 * NKT * NW) * (32 + 4) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[(QBYTES + KBYTES > PBYTES ? QBYTES + KBYTES : PBYTES)];
    __shared__ float red[2][NW][32];
    unsigned short *sQ = reinterpret_cast<unsigned short *>(smem);    // [plane][32][LDQ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * 32, d0 = blockIdx.y * 128;
    const int kb = (q0 / p.n_img) * p.n_img + wave * (32 * NKT);      // this wavefront's keys of the query tile's image
    // ---- the query tile, split, into LDS: thread -> (row, float4) pieces
#pragma unroll
    for (int it = 0; it < 32 * (CD / 4) / NT; ++it) {
        const int idx = tid + NT * it, r = idx / (CD / 4), c4 = idx % (CD / 4);
        const f32x4 a = *reinterpret_cast<const f32x4 *>(p.q + (int64_t)(q0 + r) * p.ld + 4 * c4);
        unsigned hi[2], lo[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const _Float16 h0 = (_Float16)a[2 * e], h1 = (_Float16)a[2 * e + 1];
            const _Float16 l0 = (_Float16)(a[2 * e] - (float)h0), l1 = (_Float16)(a[2 * e + 1] - (float)h1);
            hi[e] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            lo[e] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
        }
        *reinterpret_cast<u32x2 *>(sQ + r * LDQ + 4 * c4) = u32x2{hi[0], hi[1]};
        *reinterpret_cast<u32x2 *>(sQ + (32 + r) * LDQ + 4 * c4) = u32x2{lo[0], lo[1]};
    }
    // ---- S^T: keys of tile i = kb + 32 i + lr are MFMA rows; k-step t covers d = 16 t + 8 lh + 0..7; K one k-step ahead
    f32x16 sacc[NKT];
#pragma unroll
    for (int i = 0; i < NKT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[i][e] = 0.f;
    // K reaches the MFMAs through LDS in chunks of 32 channels, loaded as WHOLE 128-byte lines (eight threads per key row).  (Round 5, the
    // forms before this one read the fragments straight from L2 — lane = (key row, 8 channels), 64 scattered 16- and 32-byte pieces per
    // instruction, every line touched by eight instructions: 21 - 24 us per launch whatever the read-ahead (one, four k-steps) and the
    // wavefront count (four, eight): the CU's address path, not latency, not the split arithmetic.)
    constexpr int KEYS = 32 * NKT * NW, DCH = 32, NCH = CD / DCH, KLD = DCH + 4, PER = KEYS * (DCH / 4) / NT;
    static_assert(KEYS * (DCH / 4) % NT == 0, "whole float4 pieces per thread");
    float *sK = reinterpret_cast<float *>(smem + QBYTES);             // [2][KEYS][KLD]
    const float *kbase = p.k + (int64_t)((q0 / p.n_img) * p.n_img) * p.ld;
    f32x4 kr[PER];
    auto kfetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + NT * i;
            kr[i] = *reinterpret_cast<const f32x4 *>(kbase + (int64_t)(idx / (DCH / 4)) * p.ld + c * DCH + 4 * (idx % (DCH / 4)));
        }
    };
    auto kstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + NT * i;
            *reinterpret_cast<f32x4 *>(sK + (buf * KEYS + idx / (DCH / 4)) * KLD + 4 * (idx % (DCH / 4))) = kr[i];
        }
    };
    kfetch(0);
    kstore(0);
    kfetch(1);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int tt = 0; tt < DCH / 16; ++tt) {
            const int t = c * (DCH / 16) + tt;
            const u32x4 qh = *reinterpret_cast<const u32x4 *>(sQ + lr * LDQ + t * 16 + lh * 8);
            const u32x4 ql = *reinterpret_cast<const u32x4 *>(sQ + (32 + lr) * LDQ + t * 16 + lh * 8);
#pragma unroll
            for (int i = 0; i < NKT; ++i) {
                const float *kq = sK + ((c & 1) * KEYS + wave * 32 * NKT + 32 * i + lr) * KLD + tt * 16 + lh * 8;
                const f32x4 k0 = *reinterpret_cast<const f32x4 *>(kq), k1 = *reinterpret_cast<const f32x4 *>(kq + 4);
                const float kv[8] = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
                u32x4 kh, kl;
                split8(kv, kh, kl);
                sacc[i] = mfma16(kh, qh, sacc[i]);                    // q_hi . k_hi, q_hi . k_lo, q_lo . k_hi: the generic kernel's order
                sacc[i] = mfma16(kl, qh, sacc[i]);
                sacc[i] = mfma16(kh, ql, sacc[i]);
            }
        }
        if (c + 1 < NCH) {
            kstore((c + 1) & 1);                                      // (the buffer chunk c - 1 was read from: everybody passed the last barrier)
            if (c + 2 < NCH) kfetch(c + 2);
        }
        __syncthreads();
    }
    // every V value this wavefront needs (4 d-tiles x NKT key tiles x 2 k-steps x 8 keys: 64 NKT registers) is requested before the
    // soft-max: its trip to L2 runs under the exponentials and the two barriers
    float vv[4][NKT][2][8];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const float *vp = p.v + (int64_t)(kb + 4 * lh) * p.ld + d0 + dt * 32 + lr;
#pragma unroll
        for (int i = 0; i < NKT; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) vv[dt][i][t][sl] = vp[(int64_t)(32 * i + 16 * t + 8 * (sl >> 2) + (sl & 3)) * p.ld];
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- soft-max over the image's keys (softmax_rows_kernel's arithmetic); lane = query lr, 16 NKT keys, the other half in lane ^ 32
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NKT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[i][e] *= p.scale;
            mx = fmaxf(mx, sacc[i][e]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (lh == 0) red[0][wave][lr] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0][lr], red[0][1][lr]), fmaxf(red[0][2][lr], red[0][3][lr]));
    if constexpr (NW == 8) mx = fmaxf(mx, fmaxf(fmaxf(red[0][4][lr], red[0][5][lr]), fmaxf(red[0][6][lr], red[0][7][lr])));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NKT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[i][e] = expf(sacc[i][e] - mx);
            sum += sacc[i][e];
        }
    sum += __shfl_xor(sum, 32, 64);
    if (lh == 0) red[1][wave][lr] = sum;
    __syncthreads();
    float tot = (red[1][0][lr] + red[1][1][lr]) + (red[1][2][lr] + red[1][3][lr]);
    if constexpr (NW == 8) tot += (red[1][4][lr] + red[1][5][lr]) + (red[1][6][lr] + red[1][7][lr]);
    const float inv = 1.0f / tot;
    if (p.range_flag && sgam_not_finite(inv)) atomicOr(p.range_flag, 1);
    u32x4 ph[NKT][2], pl[NKT][2];                                     // P^T fragments: k-slot s of (k-step t, half lh) = key(t, lh, s)
#pragma unroll
    for (int i = 0; i < NKT; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float pv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = (sacc[i][8 * t + e] * inv) * 1024.0f;
            split8(pv, ph[i][t], pl[i][t]);
        }
    // ---- O^T partial of this wavefront's keys for the workgroup's four 32-channel tiles
    float *part = reinterpret_cast<float *>(smem);                    // [wave][d-tile][slot e][lane]: the Q panel is no longer read ...
    __syncthreads();                                                  // ... by anybody
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        f32x16 oacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[e] = 0.f;
#pragma unroll
        for (int i = 0; i < NKT; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4 vh, vl;
                split8(vv[dt][i][t], vh, vl);
                oacc = mfma16(vh, ph[i][t], oacc);                    // p_hi . v_hi, p_hi . v_lo, p_lo . v_hi
                oacc = mfma16(vl, ph[i][t], oacc);
                oacc = mfma16(vh, pl[i][t], oacc);
            }
#pragma unroll
        for (int e = 0; e < 16; ++e) part[((wave * 4 + dt) * 16 + e) * 64 + lane] = oacc[e];
    }
    __syncthreads();
    // the NW partials of every (d-tile, slot) are added in a fixed order: wavefront w takes d-tile w % 4 and, with eight wavefronts, the slot
    // half w / 4; lane = query lr, channels d0 + 32 dt + 8 j + 4 lh + 0..3
    {
        const int dt = wave & 3, j0 = (NW == 8) ? 2 * (wave >> 2) : 0;
        float *dst = p.out + (int64_t)(q0 + lr) * p.ldo + d0 + 32 * dt + 4 * lh;
#pragma unroll
        for (int jj = 0; jj < (NW == 8 ? 2 : 4); ++jj) {
            const int j = j0 + jj;
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = 4 * j + r;
                float a[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) a[w] = part[((w * 4 + dt) * 16 + e) * 64 + lane];
                float t4 = (a[0] + a[1]) + (a[2] + a[3]);
                if constexpr (NW == 8) t4 += (a[4] + a[5]) + (a[6] + a[7]);
                o[r] = t4 * (1.0f / 1024.0f);
            }
            *reinterpret_cast<f32x4 *>(dst + 8 * j) = o;
        }
    }
}

// =====================================================================================================================
// 16-bit throughput variant (bf16 `HT` = 0 / fp16 `HT` = 1 activations, the h16.hip mode): the same pass over the keys with
// ONE MFMA per product and single-plane fragments — half the LDS block (16 KB of K + 16 KB of V^T per 32 keys), a 64-VGPR
// query panel, probabilities rounded once to 16 bits (RNE) after the exponential.  Scores, soft-max statistics and the
// output accumulation stay in fp32, as in the GEMM / softmax / GEMM chain of that mode.
// =====================================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int HT> struct HM;
template <> struct HM<0> {
    __device__ static __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ unsigned pack2(float a, float b) {
        return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
    }
    __device__ static __forceinline__ unsigned short from_f(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
};
template <> struct HM<1> {
    __device__ static __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ unsigned pack2(float a, float b) {
        return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)b) << 16);
    }
    __device__ static __forceinline__ unsigned short from_f(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
};

constexpr int HBLK_BYTES = KB * AD * 2;     // one 32-key block of 16-bit K (or V^T) fragments = 16 KB

// k, v (16-bit, row stride ld elements) -> [key block][8 tiles][(k-step * 2 + half) * 32 + row][8 halfs]; the K piece is a
// straight 16-byte copy, the V^T piece gathers the keys key(t, h, 0..7) of one d
__global__ __launch_bounds__(256) void attn_split_kv_h16_kernel(const unsigned short *__restrict__ k, const unsigned short *__restrict__ v,
                                                                int ld, int n, unsigned short *__restrict__ kf,
                                                                unsigned short *__restrict__ vf) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int row = gid & 31, h = (gid >> 5) & 1, t = (gid >> 6) & 1, tile = (gid >> 7) & 7, kb = gid >> 10;
    if (kb * KB >= n) return;
    const int64_t piece = ((int64_t)(kb * 8 + tile) * 128 + (t * 2 + h) * 32 + row) * 8;
    *reinterpret_cast<u32x4 *>(kf + piece) =
        *reinterpret_cast<const u32x4 *>(k + (int64_t)(kb * KB + row) * ld + tile * 32 + t * 16 + h * 8);
    unsigned short vv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) vv[s] = v[(int64_t)(kb * KB + 4 * h + 8 * (2 * t + (s >> 2)) + (s & 3)) * ld + tile * 32 + row];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (unsigned)vv[2 * e] | ((unsigned)vv[2 * e + 1] << 16);
    *reinterpret_cast<u32x4 *>(vf + piece) = o;
}

// ---------------------------------------------------------------------------------------------------------------------
// Front end of the AttnBlock in the 16-bit mode, ONE launch (round 5; reference modules/diffusionmodules/model.py:168-175:
// h_ = self.norm(x); q = self.q(h_); k = self.k(h_); v = self.v(h_)): GroupNorm applied while the operand panel is staged
// (y = x scale + shift from the per-channel table, fp32, one rounding to 16 bits — the arithmetic of gn_apply_kernel), the stacked
// q | k | v projection, and the outputs written where the flash kernel wants them: q row-major, K as its MFMA fragments (a lane
// holds 16 consecutive channels of one token = two 16-byte fragment pieces, 32 lanes = 512 contiguous bytes), V^T as its
// fragments through an LDS transpose.  Replaces gn_apply (a read + a write of the activation), the generic 1 x 1 GEMM and
// attn_split_kv_h16_kernel: three launches and two graph edges of a B = 1 frame per block.
//
// The product is computed TRANSPOSED: weights are the MFMA rows, tokens the columns.  MFMA row 8 j + 4 h + i of a 32-channel
// tile carries channel 16 h + 4 j + i (the packing of sgam_pack_qkv_weight_h16), so accumulator slot e of lane (token, h) is
// channel 16 h + e.  Tile: 128 channels (four wavefronts, 32 each) x 64 tokens (two column tiles share every weight fragment);
// K = C = 256 is one panel: a single barrier between staging and the 16 k-steps.
// ---------------------------------------------------------------------------------------------------------------------
struct QkvHParams {
    const unsigned short *x;        // [nt][ldx] 16-bit activation (the block input)
    const float *table;             // [B][AD][2] {scale, shift} per (image, channel): sgam_groupnorm_table_from_partials
    const unsigned short *w;        // fragment-ordered stacked weights [3 AD / 32][AD / 16][64 lanes][8 halfs]
    const float *bias;              // [3 AD]
    unsigned short *q, *kf, *vf;    // q as the flash kernel's B fragments [nt / 32][16][64][8]; K / V^T fragments (attn_split_kv_h16_kernel's layout)
    int ldx, n_img;
};

template <int HT>
__global__ __launch_bounds__(256, 2) void attn_qkv_gn_h16_kernel(const QkvHParams p) {
    constexpr int LDK = AD + 8;                                       // panel row pitch in halfs (rows 4 banks apart)
    constexpr int VLD = 64 + 8;                                       // V transpose: [128 channels][64 tokens + pad]
    __shared__ __attribute__((aligned(16))) unsigned short sB[64 * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 64, by = blockIdx.y;                  // 64 tokens; 128 of the 768 output channels
    const int lr = lane & 31, lh = lane >> 5;
    const int b = m0 / p.n_img;
    constexpr int WD = 4, KS = AD / 16;
    const unsigned short *wt = p.w + (int64_t)(by * 4 + wave) * KS * 512 + lane * 8;      // + k-step * 512 halfs
    u32x4 wf[WD];
#pragma unroll
    for (int t = 0; t < WD; ++t) wf[t] = *reinterpret_cast<const u32x4 *>(wt + t * 512);
    // ---- stage the 64 x 256 panel: thread = (row, 8 channels), eight rows each; the channel octet is the same in every round
    {
        const int c8 = (tid & 31) * 8, r0 = tid >> 5;
        float sc[8], sf[8];
        u32x4 xr[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) xr[it] = *reinterpret_cast<const u32x4 *>(p.x + (int64_t)(m0 + r0 + it * 8) * p.ldx + c8);
        {
            const float *tab = p.table + ((int64_t)b * AD + c8) * 2;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const f32x4 tv = *reinterpret_cast<const f32x4 *>(tab + e4 * 4);
                sc[2 * e4] = tv[0], sf[2 * e4] = tv[1], sc[2 * e4 + 1] = tv[2], sf[2 * e4 + 1] = tv[3];
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned u = xr[it][e];
                float f0, f1;
                if (HT == 0) f0 = __builtin_bit_cast(float, u << 16), f1 = __builtin_bit_cast(float, u & 0xffff0000u);
                else f0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)), f1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
                o[e] = HM<HT>::pack2(__builtin_fmaf(f0, sc[2 * e], sf[2 * e]), __builtin_fmaf(f1, sc[2 * e + 1], sf[2 * e + 1]));
            }
            *reinterpret_cast<u32x4 *>(&sB[(r0 + it * 8) * LDK + c8]) = o;
        }
    }
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    __syncthreads();
    // ---- 16 k-steps: weight fragment (MFMA rows) WD steps ahead in a register ring, token fragments (k = 16 t + 8 lh + 0..7 of
    // token lr of each column tile) from the panel
#pragma unroll
    for (int t = 0; t < KS; ++t) {
        const u32x4 a = wf[t % WD];
        wf[t % WD] = *reinterpret_cast<const u32x4 *>(wt + (t + WD < KS ? t + WD : KS - 1) * 512);
        __builtin_amdgcn_sched_barrier(0);                            // (keeps the request in front of this step's MFMAs: gemm_gn_f32x.hip)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            acc[j] = HM<HT>::mfma(a, *reinterpret_cast<const u32x4 *>(&sB[(j * 32 + lr) * LDK + t * 16 + lh * 8]), acc[j]);
    }
    // ---- epilogue: lane (token lr of column tile j, half lh) holds channels cw = 32 wave + 16 lh + 0..15 of this 128-channel slab
    const int cw = wave * 32 + 16 * lh;
    float bias[16];
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(p.bias + by * 128 + cw + e4 * 4);
        bias[4 * e4] = bv[0], bias[4 * e4 + 1] = bv[1], bias[4 * e4 + 2] = bv[2], bias[4 * e4 + 3] = bv[3];
    }
    u32x4 pk[2][2];                                                   // [column tile][channels 0..7 | 8..15]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; e += 2)
            pk[j][e >> 3][(e & 7) >> 1] = HM<HT>::pack2(acc[j][e] + bias[e], acc[j][e + 1] + bias[e + 1]);
    if (by < 2) {                                                     // q: the flash kernel's B fragments, k-step (by 128 + cw) / 16, lane half = channel octet
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned short *dst = p.q + ((int64_t)((m0 >> 5) + j) * 16 + by * 8 + wave * 2 + lh) * 512 + lr * 8;
            *reinterpret_cast<u32x4 *>(dst) = pk[j][0];
            *reinterpret_cast<u32x4 *>(dst + 32 * 8) = pk[j][1];
        }
    } else if (by < 4) {                                              // K fragments: piece (k-step lh, half h, key lr) of d-tile (by - 2) 4 + wave
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t kb = (m0 >> 5) + j;
            unsigned short *dst = p.kf + ((kb * 8 + (by - 2) * 4 + wave) * 128 + (lh * 2) * 32 + lr) * 8;
            *reinterpret_cast<u32x4 *>(dst) = pk[j][0];
            *reinterpret_cast<u32x4 *>(dst + 32 * 8) = pk[j][1];
        }
    } else {                                                          // V^T fragments: 8 KEYS of one channel per piece -> transpose in LDS
        __syncthreads();                                              // every wavefront is done reading the panel
        unsigned short *vt = sB;                                      // [128 channels][VLD]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                vt[(cw + e) * VLD + j * 32 + lr] = (unsigned short)((pk[j][e >> 3][(e & 7) >> 1] >> (16 * (e & 1))) & 0xffffu);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pp = tid + 256 * k;                             // (key block of the tile, d-tile, k-step * 2 + half, channel row)
            const int drow = pp & 31, th = (pp >> 5) & 3, tl = (pp >> 7) & 3, kbl = pp >> 9;
            const unsigned short *src = vt + (tl * 32 + drow) * VLD + kbl * 32 + 4 * (th & 1) + 16 * (th >> 1);
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(src), hi = *reinterpret_cast<const u32x2 *>(src + 8);
            const int64_t kb = (m0 >> 5) + kbl;
            *reinterpret_cast<u32x4 *>(p.vf + ((kb * 8 + (by - 4) * 4 + tl) * 128 + th * 32 + drow) * 8) = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    }
}

struct AttnHParams {
    int q_frag;                         // q is [n / 32][16 k-steps][64 lanes][8 halfs] (attn_qkv_gn_h16_kernel) instead of rows
    const unsigned short *q, *kf, *vf;
    float *ws_o, *ws_ml;
    int ld, n, blocks_per_split;    // n = tokens of the whole batch, as in AttnParams
    int n_img, nsplit;
    float qscale_log2e;             // C^-1/2 * log2(e): applied to the fp32 scores
};

// 128 queries per workgroup (four wavefronts), two workgroups per CU by the register budget (one per CU on the B = 1 grid of 256)
template <int HT>
__global__ __launch_bounds__(256, 2) void attn_flash_h16_kernel(const AttnHParams p) {
    constexpr int NWV = 4;
    // K buffers 0, 1 | V^T buffers 0, 1; the epilogue's transpose wants one block per wavefront
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * HBLK_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sp = blockIdx.x % p.nsplit, qb = blockIdx.x / p.nsplit;
    const int q0 = qb * (32 * NWV) + wave * 32;
    const int lq = lane & 31, lh = lane >> 5;
    const int kb0 = ((qb * (32 * NWV)) / p.n_img) * (p.n_img / KB) + sp * p.blocks_per_split, nb = p.blocks_per_split;
    const unsigned char *kg = reinterpret_cast<const unsigned char *>(p.kf), *vg = reinterpret_cast<const unsigned char *>(p.vf);
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    // a wavefront moves 16 KB / NWV of every block: 1 KB LDS-DMA pieces behind one (address, M0) setup
    constexpr int WB = HBLK_BYTES / NWV;
    auto dma = [&](const unsigned char *g, int kb, int slot) {
        const unsigned char *src = g + (int64_t)kb * HBLK_BYTES + wave_s * WB + lane * 16;
        unsigned char *dst = smem + slot * HBLK_BYTES + wave_s * WB;
        __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)dst, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)dst, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)dst, 16, 3072, 0);
    };
    auto blk = [&](int j) { return kb0 + (j < nb ? j : nb - 1); };
    dma(kg, blk(0), 0);
    dma(vg, blk(0), 2);

    u32x4 qf[16];                      // query panel: k-step t = d 16 t + 8 h + 0..7 of query q0 + lq
    if (p.q_frag) {                    // left as fragments by the fused front end: 16 coalesced kilobyte loads
        const unsigned short *qp = p.q + (int64_t)(q0 >> 5) * (16 * 512) + lane * 8;
#pragma unroll
        for (int t = 0; t < 16; ++t) qf[t] = *reinterpret_cast<const u32x4 *>(qp + t * 512);
    } else {
        const unsigned short *qp = p.q + (int64_t)(q0 + lq) * p.ld + lh * 8;
#pragma unroll
        for (int t = 0; t < 16; ++t) qf[t] = *reinterpret_cast<const u32x4 *>(qp + t * 16);
    }
    f32x16 o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    __syncthreads();

    // two wavefronts per SIMD here (half the registers of the split-fp32 kernel): the neighbour covers the soft-max
    // arithmetic and the LDS round trips, so the loop is the plain S -> soft-max -> PV order, fragments read by the compiler
    for (int j = 0; j < nb; ++j) {
        const int buf = j & 1;
        const unsigned char *lk = smem + buf * HBLK_BYTES + lane * 16, *lv = lk + 2 * HBLK_BYTES;
        if (j + 1 < nb) {
            dma(kg, kb0 + j + 1, buf ^ 1);
            dma(vg, kb0 + j + 1, 2 + (buf ^ 1));
        }
        f32x16 sa, sb;
#pragma unroll
        for (int e = 0; e < 16; ++e) sa[e] = sb[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; t += 2) {        // fragment (tile t / 2, k-step t % 2) at (t / 2) * 2048 + (t % 2) * 1024
            sa = HM<HT>::mfma(*reinterpret_cast<const u32x4 *>(lk + (t >> 1) * 2048), qf[t], sa);
            sb = HM<HT>::mfma(*reinterpret_cast<const u32x4 *>(lk + (t >> 1) * 2048 + 1024), qf[t + 1], sb);
        }
        float s[16], mloc = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = (sa[e] + sb[e]) * p.qscale_log2e;
            mloc = fmaxf(mloc, s[e]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        if (__any(m_new > m_run + RESCALE_TAU)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float x = o[i][e], tmp;
                    asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1"
                                 : "+a"(x), "=&v"(tmp)
                                 : "v"(alpha));
                    o[i][e] = x;
                }
            m_run = m_new;
        }
        u32x4 pf[2];
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const float p0 = __builtin_amdgcn_exp2f(s[e] - m_run), p1 = __builtin_amdgcn_exp2f(s[e + 1] - m_run);
            // the sum runs over the ROUNDED probabilities, so numerator and denominator see the same values
            const unsigned pk = HM<HT>::pack2(p0, p1);
            pf[e >> 3][(e & 7) >> 1] = pk;
            if (HT == 0) l_run += __builtin_bit_cast(float, pk << 16) + __builtin_bit_cast(float, pk & 0xffff0000u);
            else l_run += (float)__builtin_bit_cast(_Float16, (unsigned short)(pk & 0xffffu)) + (float)__builtin_bit_cast(_Float16, (unsigned short)(pk >> 16));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                o[i] = HM<HT>::mfma(*reinterpret_cast<const u32x4 *>(lv + i * 2048 + t * 1024), pf[t], o[i]);
        __syncthreads();
    }

    l_run += __shfl_xor(l_run, 32, 64);
    if (lh == 0) {
        float *ml = p.ws_ml + ((int64_t)sp * p.n + q0 + lq) * 2;
        ml[0] = m_run;
        ml[1] = l_run;
    }
    // The partial output of a key range leaves NORMALISED (o / l: a convex combination of 16-bit v rows, so |o / l| <= max |v|), as
    // fp16 and ROW-MAJOR [range][query][256]: 2 bytes per element instead of the 4 of the un-normalised fp32 sums (round 5: 56 MB
    // of HBM traffic per launch against 8.4 MB algorithmic was the write and re-read of 8 x 4 MB of fp32 partials).  A lane holds
    // 4 consecutive channels of ONE query per accumulator quad, so the tile goes through the (now idle) K / V buffers — each
    // wavefront its own 16 KB, 16-byte units XOR-swizzled by the row — and leaves as whole 512-byte rows, two per store
    // instruction (the first form of this epilogue stored 2 bytes per lane straight from the accumulators: half-line writes, the
    // kernel got 2 us SLOWER than with twice the bytes).  fp16's 11 bits are finer than the bf16 output and equal to the fp16
    // output's own rounding; the merge weighs range s by l_s 2^(m_s - M).  (bf16 inputs beyond fp16's range would saturate at
    // +-65504 here; activations behind a GroupNorm and a 1 x 1 projection are five orders below.)
    const float inv_l = 1.0f / l_run;
    unsigned char *tw = smem + wave * HBLK_BYTES;          // (every wavefront passed the loop's last barrier: the buffers are free)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned pk[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const float a0 = __builtin_amdgcn_fmed3f(o[i][4 * j + 2 * h2] * inv_l, -65504.f, 65504.f);
                const float a1 = __builtin_amdgcn_fmed3f(o[i][4 * j + 2 * h2 + 1] * inv_l, -65504.f, 65504.f);
                pk[h2] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)a0) |
                         ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)a1) << 16);
            }
            // channels 32 i + 8 j + 4 lh + 0..3 of query lq: 16-byte unit 4 i + j (swizzled), half lh
            *reinterpret_cast<u32x2 *>(tw + lq * 512 + (((4 * i + j) ^ lq) << 4) + lh * 8) = u32x2{pk[0], pk[1]};
        }
    // (wave-private region, but the reads below cross lanes: LDS operations of one wavefront complete in order)
    unsigned short *wo = reinterpret_cast<unsigned short *>(p.ws_o) + ((int64_t)sp * p.n + q0) * AD;
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) {
        const int row = 2 * r2 + lh;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(tw + row * 512 + ((lq ^ row) << 4));
        *reinterpret_cast<u32x4 *>(wo + (int64_t)row * AD + lq * 8) = v;
    }
}

// merge of the key ranges for the 16-bit variant: the flash kernel left NORMALISED fp16 partials o_s / l_s, row-major
// [range][query][256], so range s weighs l_s 2^(m_s - M) / sum of those; thread = (query row, 4 channels) for loads and store alike;
// output rounded to 16 bits
template <int HT, int NS>
__global__ __launch_bounds__(256) void attn_combine_h16_kernel(const float *__restrict__ ws_o_f, const float *__restrict__ ws_ml,
                                                               unsigned short *__restrict__ out, int ldo, int n) {
    const unsigned short *ws_o = reinterpret_cast<const unsigned short *>(ws_o_f);
    const int qt = blockIdx.x >> 3, dg = blockIdx.x & 7;
    const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    const int64_t row = (int64_t)qt * 32 + r;
    float w[NS], M = -INFINITY, L = 0.f;
    u32x2 o[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) o[s] = *reinterpret_cast<const u32x2 *>(ws_o + ((int64_t)s * n + row) * AD + dg * 32 + c4);
    float l[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float *ml = ws_ml + ((int64_t)s * n + row) * 2;
        w[s] = ml[0];
        l[s] = ml[1];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) M = fmaxf(M, w[s]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        w[s] = __builtin_amdgcn_exp2f(w[s] - M) * l[s];
        L += w[s];
    }
    const float inv = 1.0f / L;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[j] += w[s] * (float)__builtin_bit_cast(_Float16, (unsigned short)((o[s][j >> 1] >> (16 * (j & 1))) & 0xffffu));
    unsigned short *dst = out + row * ldo + dg * 32 + c4;
    u32x2 pk;
    pk[0] = (unsigned)HM<HT>::from_f(acc[0] * inv) | ((unsigned)HM<HT>::from_f(acc[1] * inv) << 16);
    pk[1] = (unsigned)HM<HT>::from_f(acc[2] * inv) | ((unsigned)HM<HT>::from_f(acc[3] * inv) << 16);
    *reinterpret_cast<u32x2 *>(dst) = pk;
}

// ---------------------------------------------------------------------------------------------------------------------
// Merge of the key ranges + AttnBlock.proj_out + residual of the 16-bit mode in ONE launch (round 5; the split-fp32 path's
// attn_combine_proj_f32x_kernel re-balanced for one MFMA per product).  A workgroup owns one 32-query tile: thread = (query row,
// 32 channels) merges the NS normalised fp16 partial rows exactly as attn_combine_h16_kernel does (same weights, same order of the
// fused multiply-adds, the same rounding of the merged value to 16 bits), writes the tile into LDS as the B operand of the TRANSPOSED
// projection (proj_out's weights = MFMA rows, sgam_pack_weight_tp_h16; four wavefronts x 64 output channels), adds bias and the
// residual, rounds, stores 32 bytes per lane and leaves the GroupNorm statistics of the STORED block output (one chunk per tile).
// ---------------------------------------------------------------------------------------------------------------------
struct CombProjHParams {
    const unsigned short *ws_o;     // normalised fp16 partials [ns][n][AD]
    const float *ws_ml;             // {max, sum} [ns][n][2]
    const unsigned short *w;        // proj_out weights, transposed-product fragment order [AD / 32][AD / 16][64 lanes][8 halfs]
    const float *bias;              // [AD] or NULL
    const unsigned short *res;      // residual [n][ldr] (the block's input x) or NULL
    unsigned short *out;            // [n][ldc]
    double *gn_partial;             // optional [n / 32][32][2]
    int n, ldr, ldc;
};

template <int HT, int NS>
__global__ __launch_bounds__(256) void attn_combine_proj_h16_kernel(const CombProjHParams p) {
    constexpr int LDK = AD + 8;
    __shared__ __attribute__((aligned(16))) unsigned short sB[32 * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qt = blockIdx.x, m0 = qt * 32;
    const int lr = lane & 31, lh = lane >> 5;
    constexpr int WD = 4, KS = AD / 16;
    const unsigned short *wt0 = p.w + (int64_t)(wave * 2) * KS * 512 + lane * 8, *wt1 = wt0 + (int64_t)KS * 512;
    // ---- merge: thread = (row r, channels 32 seg + 0..31): all NS x 4 sixteen-byte loads of the thread in flight at once
    {
        const int r = tid >> 3, seg = tid & 7;
        const int64_t row = (int64_t)m0 + r;
        u32x4 o[NS][4];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k) o[s][k] = *reinterpret_cast<const u32x4 *>(p.ws_o + ((int64_t)s * p.n + row) * AD + seg * 32 + k * 8);
        float w[NS], l[NS], M = -INFINITY, L = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float *ml = p.ws_ml + ((int64_t)s * p.n + row) * 2;
            w[s] = ml[0];
            l[s] = ml[1];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) M = fmaxf(M, w[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            w[s] = __builtin_amdgcn_exp2f(w[s] - M) * l[s];
            L += w[s];
        }
        const float inv = 1.0f / L;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    acc[e] += w[s] * (float)__builtin_bit_cast(_Float16, (unsigned short)((o[s][k][e >> 1] >> (16 * (e & 1))) & 0xffffu));
            u32x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = HM<HT>::pack2(acc[2 * e] * inv, acc[2 * e + 1] * inv);
            *reinterpret_cast<u32x4 *>(&sB[r * LDK + seg * 32 + k * 8]) = pk;
        }
    }
    u32x4 wf[WD][2];
#pragma unroll
    for (int t = 0; t < WD; ++t) {
        wf[t][0] = *reinterpret_cast<const u32x4 *>(wt0 + t * 512);
        wf[t][1] = *reinterpret_cast<const u32x4 *>(wt1 + t * 512);
    }
    // residual rows of this lane's outputs: token m0 + lr, channels 64 wave + 32 rt + 16 lh + 0..15, requested under the MFMAs
    u32x4 rres[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            rres[rt][h] = p.res ? *reinterpret_cast<const u32x4 *>(p.res + (int64_t)(m0 + lr) * p.ldr + wave * 64 + rt * 32 + 16 * lh + 8 * h)
                                : u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[rt][e] = 0.f;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
        const u32x4 a0 = wf[t % WD][0], a1 = wf[t % WD][1];
        const int nx = t + WD < KS ? t + WD : KS - 1;
        wf[t % WD][0] = *reinterpret_cast<const u32x4 *>(wt0 + nx * 512);
        wf[t % WD][1] = *reinterpret_cast<const u32x4 *>(wt1 + nx * 512);
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 bx = *reinterpret_cast<const u32x4 *>(&sB[lr * LDK + t * 16 + lh * 8]);
        acc[0] = HM<HT>::mfma(a0, bx, acc[0]);
        acc[1] = HM<HT>::mfma(a1, bx, acc[1]);
    }
    // ---- epilogue
    double gsum[2][2], gsq[2][2];                                     // [row tile][group of 8 channels inside the lane's 16]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int c0 = wave * 64 + rt * 32 + 16 * lh;
        float bias[16];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + c0 + e4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            bias[4 * e4] = bv[0], bias[4 * e4 + 1] = bv[1], bias[4 * e4 + 2] = bv[2], bias[4 * e4 + 3] = bv[3];
        }
        u32x4 pk[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float s = 0.f, ss = 0.f;
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const unsigned ru = rres[rt][h][e2];
                float r0, r1;
                if (HT == 0) r0 = __builtin_bit_cast(float, ru << 16), r1 = __builtin_bit_cast(float, ru & 0xffff0000u);
                else r0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(ru & 0xffffu)), r1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(ru >> 16));
                const int e = 8 * h + 2 * e2;
                const unsigned u = HM<HT>::pack2((acc[rt][e] + bias[e]) + r0, (acc[rt][e + 1] + bias[e + 1]) + r1);
                pk[h][e2] = u;
                float f0, f1;                                          // the statistics describe the STORED (rounded) tensor
                if (HT == 0) f0 = __builtin_bit_cast(float, u << 16), f1 = __builtin_bit_cast(float, u & 0xffff0000u);
                else f0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)), f1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
                s += f0 + f1;
                ss += f0 * f0 + f1 * f1;
            }
            gsum[rt][h] = (double)s;
            gsq[rt][h] = (double)ss;
        }
        unsigned short *dst = p.out + (int64_t)(m0 + lr) * p.ldc + c0;
        *reinterpret_cast<u32x4 *>(dst) = pk[0];
        *reinterpret_cast<u32x4 *>(dst + 8) = pk[1];
    }
    if (p.gn_partial) {
        // a group of the next GroupNorm = 8 channels: this lane's (rt, h); the 32 tokens of the tile sit in the 32 lanes of a half-wave
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double ds = gsum[rt][h], dss = gsq[rt][h];
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) {
                    ds += __shfl_xor(ds, o, 64);
                    dss += __shfl_xor(dss, o, 64);
                }
                if (lr == 0) {
                    const int g = (wave * 64 + rt * 32 + 16 * lh + 8 * h) / 8;
                    double *o = p.gn_partial + ((int64_t)qt * 32 + g) * 2;
                    o[0] = ds;
                    o[1] = dss;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The small AttnBlocks' attention of the 16-bit mode in one launch (attn_small_f32x_kernel's decomposition, one MFMA per product):
// replaces transpose_h16 + the q k^T GEMM + softmax_rows_h16 + the P v GEMM.  Scores and soft-max in fp32 (softmax_rows_h16_kernel's
// arithmetic: scale, maximum, __expf, sum, one reciprocal, probabilities rounded ONCE to 16 bits), fp32 accumulation, output rounded.
// ---------------------------------------------------------------------------------------------------------------------
struct SmallAttnHParams {
    const unsigned short *q, *k, *v;    // [B n][ld] 16-bit columns of the q | k | v projection
    unsigned short *out;                // [B n][ldo]
    int ld, ldo, n_img;
    float scale;
};

template <int HT, int CD, int NKT, int NW>
__global__ __launch_bounds__(64 * NW) void attn_small_h16_kernel(const SmallAttnHParams p) {
    constexpr int LDQ = CD + 8;
    constexpr int NT = 64 * NW;
    constexpr int QBYTES = 32 * LDQ * 2, PBYTES = NW * 4 * 16 * 64 * 4, KBYTES = 2 * (32 * NKT * NW) * (64 + 8) * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[(QBYTES + KBYTES > PBYTES ? QBYTES + KBYTES : PBYTES)];
    __shared__ float red[2][NW][32];
    unsigned short *sQ = reinterpret_cast<unsigned short *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * 32, d0 = blockIdx.y * 128;
    const int kb = (q0 / p.n_img) * p.n_img + wave * (32 * NKT);
#pragma unroll
    for (int it = 0; it < 32 * (CD / 8) / NT; ++it) {
        const int idx = tid + NT * it, r = idx / (CD / 8), c8 = idx % (CD / 8);
        *reinterpret_cast<u32x4 *>(sQ + r * LDQ + 8 * c8) = *reinterpret_cast<const u32x4 *>(p.q + (int64_t)(q0 + r) * p.ld + 8 * c8);
    }
    f32x16 sacc[NKT];
#pragma unroll
    for (int i = 0; i < NKT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[i][e] = 0.f;
    // K through LDS in chunks of 64 channels, loaded as whole 128-byte lines (attn_small_f32x_kernel)
    constexpr int KEYS = 32 * NKT * NW, DCH = 64, NCH = CD / DCH, KLD = DCH + 8, PER = KEYS * (DCH / 8) / NT;
    static_assert(KEYS * (DCH / 8) % NT == 0, "whole 16-byte pieces per thread");
    unsigned short *sK = reinterpret_cast<unsigned short *>(smem + QBYTES);       // [2][KEYS][KLD]
    const unsigned short *kbase = p.k + (int64_t)((q0 / p.n_img) * p.n_img) * p.ld;
    u32x4 kr[PER];
    auto kfetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + NT * i;
            kr[i] = *reinterpret_cast<const u32x4 *>(kbase + (int64_t)(idx / (DCH / 8)) * p.ld + c * DCH + 8 * (idx % (DCH / 8)));
        }
    };
    auto kstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + NT * i;
            *reinterpret_cast<u32x4 *>(sK + (buf * KEYS + idx / (DCH / 8)) * KLD + 8 * (idx % (DCH / 8))) = kr[i];
        }
    };
    kfetch(0);
    kstore(0);
    kfetch(1);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int tt = 0; tt < DCH / 16; ++tt) {
            const int t = c * (DCH / 16) + tt;
            const u32x4 qf = *reinterpret_cast<const u32x4 *>(sQ + lr * LDQ + t * 16 + lh * 8);
#pragma unroll
            for (int i = 0; i < NKT; ++i)
                sacc[i] = HM<HT>::mfma(*reinterpret_cast<const u32x4 *>(sK + ((c & 1) * KEYS + wave * 32 * NKT + 32 * i + lr) * KLD + tt * 16 + lh * 8),
                                       qf, sacc[i]);
        }
        if (c + 1 < NCH) {
            kstore((c + 1) & 1);
            if (c + 2 < NCH) kfetch(c + 2);
        }
        __syncthreads();
    }
    unsigned short vv[4][NKT][2][8];                                  // every V value of the wavefront, requested ahead of the soft-max
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const unsigned short *vp = p.v + (int64_t)(kb + 4 * lh) * p.ld + d0 + dt * 32 + lr;
#pragma unroll
        for (int i = 0; i < NKT; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) vv[dt][i][t][sl] = vp[(int64_t)(32 * i + 16 * t + 8 * (sl >> 2) + (sl & 3)) * p.ld];
    }
    __builtin_amdgcn_sched_barrier(0);
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NKT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[i][e] *= p.scale;
            mx = fmaxf(mx, sacc[i][e]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (lh == 0) red[0][wave][lr] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0][lr], red[0][1][lr]), fmaxf(red[0][2][lr], red[0][3][lr]));
    if constexpr (NW == 8) mx = fmaxf(mx, fmaxf(fmaxf(red[0][4][lr], red[0][5][lr]), fmaxf(red[0][6][lr], red[0][7][lr])));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NKT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[i][e] = __expf(sacc[i][e] - mx);
            sum += sacc[i][e];
        }
    sum += __shfl_xor(sum, 32, 64);
    if (lh == 0) red[1][wave][lr] = sum;
    __syncthreads();
    float tot = (red[1][0][lr] + red[1][1][lr]) + (red[1][2][lr] + red[1][3][lr]);
    if constexpr (NW == 8) tot += (red[1][4][lr] + red[1][5][lr]) + (red[1][6][lr] + red[1][7][lr]);
    const float inv = 1.0f / tot;
    u32x4 pf[NKT][2];
#pragma unroll
    for (int i = 0; i < NKT; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 8; e += 2) pf[i][t][e >> 1] = HM<HT>::pack2(sacc[i][8 * t + e] * inv, sacc[i][8 * t + e + 1] * inv);
    float *part = reinterpret_cast<float *>(smem);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        f32x16 oacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[e] = 0.f;
#pragma unroll
        for (int i = 0; i < NKT; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4 vf;
#pragma unroll
                for (int e = 0; e < 4; ++e) vf[e] = (unsigned)vv[dt][i][t][2 * e] | ((unsigned)vv[dt][i][t][2 * e + 1] << 16);
                oacc = HM<HT>::mfma(vf, pf[i][t], oacc);
            }
#pragma unroll
        for (int e = 0; e < 16; ++e) part[((wave * 4 + dt) * 16 + e) * 64 + lane] = oacc[e];
    }
    __syncthreads();
    {
        const int dt = wave & 3, j0 = (NW == 8) ? 2 * (wave >> 2) : 0;
        unsigned short *dst = p.out + (int64_t)(q0 + lr) * p.ldo + d0 + 32 * dt + 4 * lh;
#pragma unroll
        for (int jj = 0; jj < (NW == 8 ? 2 : 4); ++jj) {
            const int j = j0 + jj;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = 4 * j + r;
                float a[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) a[w] = part[((w * 4 + dt) * 16 + e) * 64 + lane];
                o[r] = (a[0] + a[1]) + (a[2] + a[3]);
                if constexpr (NW == 8) o[r] += (a[4] + a[5]) + (a[6] + a[7]);
            }
            *reinterpret_cast<u32x2 *>(dst + 8 * j) = u32x2{HM<HT>::pack2(o[0], o[1]), HM<HT>::pack2(o[2], o[3])};
        }
    }
}

}  // namespace

// key ranges per image: enough workgroups (n_img / 128 query blocks x ranges x images) to fill 256 CUs and as few ranges as
// that allows (every range writes a partial O) — but never more than 2048 keys per range: a range is ONE running fp32
// accumulation (soft-max sum and the MFMA accumulators), and the merge of the ranges is what keeps the summation tree
// shallow (8 ranges of 2048 keys is what the 512 x 512 configuration has always run, and what its 1e-4 parity was measured
// with; a single 16384-key range lost a digit).  8 for one 64 x 64 image, 2 from four images on.
static int attn_nsplit(int n_img, int B) {
    int ns = NSPLIT;
    while (ns > 1 && (int64_t)(n_img / 128) * (ns / 2) * B >= 256 && (n_img / KB) % (ns / 2) == 0 && n_img / (ns / 2) <= 2048) ns /= 2;
    return ns;
}

extern "C" int64_t sgam_attention_f32x_batched_workspace_bytes(int32_t n, int32_t C, int32_t B) {
    if (C != AD || n < 256 || n % 256 != 0 || B < 1 || (int64_t)B * n >= (1 << 24)) return -1;
    const int64_t nsplit = attn_nsplit(n, B), nt = (int64_t)B * n;
    // K and V^T fragments (hi + lo fp16 = 4 bytes per element each), then the per-range partial O^T and {max, sum}
    return 2 * nt * AD * 4 + nsplit * nt * AD * 4 + nsplit * nt * 2 * 4;
}

extern "C" int64_t sgam_attention_f32x_workspace_bytes(int32_t n, int32_t C) { return sgam_attention_f32x_batched_workspace_bytes(n, C, 1); }

extern "C" int sgam_attention_f32x_batched(const float *q, const float *k, const float *v, int32_t ld, int32_t n, int32_t C, int32_t B,
                                           float scale, float *out, int32_t ldo, void *workspace, int64_t workspace_bytes,
                                           void *stream) {
    if (!q || !k || !v || !out || !workspace) return SGAM_EINVAL;
    const int64_t need = sgam_attention_f32x_batched_workspace_bytes(n, C, B);
    if (need < 0 || ld < C || ld % 4 != 0 || ldo < C) return SGAM_EINVAL;
    int ex;
    if (!(scale > 0.f) || frexpf(scale, &ex) != 0.5f) return SGAM_EINVAL;   // folded into q: must be an exact power of two
    if (workspace_bytes < need) return SGAM_EWORKSPACE;
    if (!sgam_aligned16(q) || !sgam_aligned16(k) || !sgam_aligned16(v) || !sgam_aligned16(workspace)) return SGAM_EALIGN;
    const int nsplit = attn_nsplit(n, B), nt = B * n;
    if ((n / KB) % nsplit != 0 || ldo % 4 != 0 || !sgam_aligned16(out)) return SGAM_EINVAL;
    hipStream_t s = sgam_stream(stream);
    unsigned short *kf = (unsigned short *)workspace;
    unsigned short *vf = kf + (int64_t)nt * AD * 2;
    float *ws_o = (float *)(vf + (int64_t)nt * AD * 2);
    float *ws_ml = ws_o + (int64_t)nsplit * nt * AD;
    // the images are stacked along the rows: their key blocks follow one another in the fragment buffers
    SGAM_KLAUNCH(attn_split_kv_kernel, dim3(nt / KB * 8 * 128 / 256), dim3(256), 0, s, k, v, ld, nt, kf, vf);
    SGAM_LAUNCH_CHECK();
    AttnParams p;
    p.qf = nullptr; p.q = q; p.kf = kf; p.vf = vf; p.ws_o = ws_o; p.ws_ml = ws_ml;
    p.ld = ld; p.n = nt; p.n_img = n; p.nsplit = nsplit; p.blocks_per_split = n / KB / nsplit; p.qscale = scale;
    if (sgam_i_prof_on) sgam_i_prof_work(4.0 * B * n * (double)n * AD, 4.0 * 4.0 * nt * AD);   // q k^T + P v; q, k, v, o once
    SGAM_KLAUNCH(attn_flash_f32x_kernel, dim3(nt / 128 * nsplit), dim3(256), 0, s, p);
    SGAM_LAUNCH_CHECK();
    switch (nsplit) {                         // (the number of ranges is a template argument of the merge: all its loads in flight)
        case 8: SGAM_KLAUNCH(attn_combine_kernel<8>, dim3(nt / 32 * 8), dim3(256), 0, s, ws_o, ws_ml, out, ldo, nt, sgam_i_range_flag); break;
        case 4: SGAM_KLAUNCH(attn_combine_kernel<4>, dim3(nt / 32 * 8), dim3(256), 0, s, ws_o, ws_ml, out, ldo, nt, sgam_i_range_flag); break;
        case 2: SGAM_KLAUNCH(attn_combine_kernel<2>, dim3(nt / 32 * 8), dim3(256), 0, s, ws_o, ws_ml, out, ldo, nt, sgam_i_range_flag); break;
        case 1: SGAM_KLAUNCH(attn_combine_kernel<1>, dim3(nt / 32 * 8), dim3(256), 0, s, ws_o, ws_ml, out, ldo, nt, sgam_i_range_flag); break;
        default: return SGAM_EINVAL;
    }
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_attention_f32x(const float *q, const float *k, const float *v, int32_t ld, int32_t n, int32_t C,
                                   float scale, float *out, int32_t ldo, void *workspace, int64_t workspace_bytes,
                                   void *stream) {
    return sgam_attention_f32x_batched(q, k, v, ld, n, C, 1, scale, out, ldo, workspace, workspace_bytes, stream);
}

// sgam_attention_f32x_batched + AttnBlock.proj_out (+ residual) with the merge of the key ranges fused into the projection
// (attn_combine_proj_f32x_kernel): out[B n][ldc] = residual + bias + softmax(q k^T scale) v . Wp^T.  w_planes: proj_out's weights as
// sgam_split_rows_f32x lays them out (fragment-ordered hi / lo planes, [C][C]), w_scale their power-of-two scale.  gn_partial
// (optional): [B][n / 32][32][2] fp64 {sum, sumsq} of `out` per (32-row tile, group of C / 32 channels).
// Same workspace as sgam_attention_f32x_batched.
extern "C" int sgam_attention_proj_f32x_batched(const float *q, const float *k, const float *v, int32_t ld, int32_t n, int32_t C,
                                                int32_t B, float scale, const void *w_planes, float w_scale, const float *bias,
                                                const float *residual, int32_t ldr, float *out, int32_t ldc, double *gn_partial,
                                                void *workspace, int64_t workspace_bytes, void *stream) {
    if (!q || !k || !v || !out || !workspace || !w_planes || !(w_scale > 0.f)) return SGAM_EINVAL;
    const int64_t need = sgam_attention_f32x_batched_workspace_bytes(n, C, B);
    if (need < 0 || ld < C || ld % 4 != 0 || ldc < C || (residual && ldr < C)) return SGAM_EINVAL;
    int ex;
    if (!(scale > 0.f) || frexpf(scale, &ex) != 0.5f) return SGAM_EINVAL;   // folded into q: must be an exact power of two
    if (workspace_bytes < need) return SGAM_EWORKSPACE;
    if (!sgam_aligned16(q) || !sgam_aligned16(k) || !sgam_aligned16(v) || !sgam_aligned16(workspace) || !sgam_aligned16(w_planes) ||
        (gn_partial && !sgam_aligned16(gn_partial)))
        return SGAM_EALIGN;
    const int nsplit = attn_nsplit(n, B), nt = B * n;
    if ((n / KB) % nsplit != 0) return SGAM_EINVAL;
    hipStream_t s = sgam_stream(stream);
    unsigned short *kf = (unsigned short *)workspace;
    unsigned short *vf = kf + (int64_t)nt * AD * 2;
    float *ws_o = (float *)(vf + (int64_t)nt * AD * 2);
    float *ws_ml = ws_o + (int64_t)nsplit * nt * AD;
    SGAM_KLAUNCH(attn_split_kv_kernel, dim3(nt / KB * 8 * 128 / 256), dim3(256), 0, s, k, v, ld, nt, kf, vf);
    SGAM_LAUNCH_CHECK();
    AttnParams p;
    p.qf = nullptr; p.q = q; p.kf = kf; p.vf = vf; p.ws_o = ws_o; p.ws_ml = ws_ml;
    p.ld = ld; p.n = nt; p.n_img = n; p.nsplit = nsplit; p.blocks_per_split = n / KB / nsplit; p.qscale = scale;
    if (sgam_i_prof_on) sgam_i_prof_work(4.0 * B * n * (double)n * AD, 4.0 * 4.0 * nt * AD);
    SGAM_KLAUNCH(attn_flash_f32x_kernel, dim3(nt / 128 * nsplit), dim3(256), 0, s, p);
    SGAM_LAUNCH_CHECK();
    CombProjParams c;
    c.ws_o = ws_o; c.ws_ml = ws_ml; c.w = (const unsigned short *)w_planes; c.bias = bias; c.res = residual; c.out = out;
    c.gn_partial = gn_partial; c.n_img = n;
    c.range_flag = sgam_i_range_flag; c.n = nt; c.ns = nsplit; c.ldr = ldr; c.ldc = ldc;
    c.inv_w_scale = 1.0f / w_scale;
    if (sgam_i_prof_on) sgam_i_prof_shape(nt, AD, AD, 1);
    if (sgam_i_prof_on) sgam_i_prof_work(2.0 * nt * (double)AD * AD, 4.0 * ((double)nsplit * nt * AD + 2.0 * nt * AD + (double)AD * AD));
    switch (nsplit) {
        case 8: SGAM_KLAUNCH(attn_combine_proj_f32x_kernel<8>, dim3(nt / 32), dim3(256), 0, s, c); break;
        case 4: SGAM_KLAUNCH(attn_combine_proj_f32x_kernel<4>, dim3(nt / 32), dim3(256), 0, s, c); break;
        case 2: SGAM_KLAUNCH(attn_combine_proj_f32x_kernel<2>, dim3(nt / 32), dim3(256), 0, s, c); break;
        case 1: SGAM_KLAUNCH(attn_combine_proj_f32x_kernel<1>, dim3(nt / 32), dim3(256), 0, s, c); break;
        default: return SGAM_EINVAL;
    }
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int64_t sgam_attention_h16_batched_workspace_bytes(int32_t n, int32_t C, int32_t B) {
    if (C != AD || n < 256 || n % 256 != 0 || B < 1 || (int64_t)B * n >= (1 << 24)) return -1;
    const int64_t nsplit = attn_nsplit(n, B), nt = (int64_t)B * n;
    return 2 * nt * AD * 2 + nsplit * nt * AD * 4 + nsplit * nt * 2 * 4;
}

extern "C" int64_t sgam_attention_h16_workspace_bytes(int32_t n, int32_t C) { return sgam_attention_h16_batched_workspace_bytes(n, C, 1); }

extern "C" int sgam_attention_h16_batched(const void *q, const void *k, const void *v, int32_t ht, int32_t ld, int32_t n, int32_t C,
                                          int32_t B, float scale, void *out, int32_t ldo, void *workspace, int64_t workspace_bytes,
                                          void *stream) {
    if (!q || !k || !v || !out || !workspace || (ht != 0 && ht != 1)) return SGAM_EINVAL;
    const int64_t need = sgam_attention_h16_batched_workspace_bytes(n, C, B);
    if (need < 0) return SGAM_EINVAL;
    const int nsplit = attn_nsplit(n, B), nt = B * n;
    if (ld < C || ld % 8 != 0 || ldo < C || ldo % 4 != 0 || !(scale > 0.f) || (n / KB) % nsplit != 0) return SGAM_EINVAL;
    if (workspace_bytes < need) return SGAM_EWORKSPACE;
    if (!sgam_aligned16(q) || !sgam_aligned16(k) || !sgam_aligned16(v) || !sgam_aligned16(workspace) ||
        (((uintptr_t)out) & 7u) != 0)
        return SGAM_EALIGN;
    hipStream_t s = sgam_stream(stream);
    unsigned short *kf = (unsigned short *)workspace;
    unsigned short *vf = kf + (int64_t)nt * AD;
    float *ws_o = (float *)(vf + (int64_t)nt * AD);
    float *ws_ml = ws_o + (int64_t)nsplit * nt * AD;
    SGAM_KLAUNCH(attn_split_kv_h16_kernel, dim3(nt / 8), dim3(256), 0, s, (const unsigned short *)k,
                       (const unsigned short *)v, ld, nt, kf, vf);
    SGAM_LAUNCH_CHECK();
    AttnHParams p;
    p.q_frag = 0; p.q = (const unsigned short *)q; p.kf = kf; p.vf = vf; p.ws_o = ws_o; p.ws_ml = ws_ml;
    p.ld = ld; p.n = nt; p.n_img = n; p.nsplit = nsplit; p.blocks_per_split = n / KB / nsplit; p.qscale_log2e = scale * LOG2E;
    const dim3 grid(nt / 128 * nsplit), cgrid(nt / 32 * 8);
    if (sgam_i_prof_on) sgam_i_prof_work(4.0 * B * n * (double)n * AD, 4.0 * 2.0 * nt * AD);
    if (ht == 0) SGAM_KLAUNCH(attn_flash_h16_kernel<0>, grid, dim3(256), 0, s, p);
    else SGAM_KLAUNCH(attn_flash_h16_kernel<1>, grid, dim3(256), 0, s, p);
    SGAM_LAUNCH_CHECK();
#define HCOMBINE(HT_, NS_) SGAM_KLAUNCH((attn_combine_h16_kernel<HT_, NS_>), cgrid, dim3(256), 0, s, ws_o, ws_ml, (unsigned short *)out, ldo, nt)
    switch (nsplit * 2 + (ht ? 1 : 0)) {
        case 16: HCOMBINE(0, 8); break;
        case 17: HCOMBINE(1, 8); break;
        case 8: HCOMBINE(0, 4); break;
        case 9: HCOMBINE(1, 4); break;
        case 4: HCOMBINE(0, 2); break;
        case 5: HCOMBINE(1, 2); break;
        case 2: HCOMBINE(0, 1); break;
        case 3: HCOMBINE(1, 1); break;
        default: return SGAM_EINVAL;
    }
#undef HCOMBINE
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_attention_h16(const void *q, const void *k, const void *v, int32_t ht, int32_t ld, int32_t n, int32_t C,
                                  float scale, void *out, int32_t ldo, void *workspace, int64_t workspace_bytes, void *stream) {
    return sgam_attention_h16_batched(q, k, v, ht, ld, n, C, 1, scale, out, ldo, workspace, workspace_bytes, stream);
}

// ---- fused AttnBlock front end + attention of the 16-bit mode (ABI v9) -------------------------------------------------------
// stacked q | k | v weights [3 C][C] (fp32, rows q then k then v) -> the fragment order attn_qkv_gn_h16_kernel reads: per 32-row tile and
// 16-wide k-step one kilobyte, lane (h, rho) = 8 consecutive k of the row that MFMA row rho carries (channel 16 ((rho >> 2) & 1) +
// 4 (rho >> 3) + (rho & 3) of the tile)
namespace {
template <int HT>
__global__ __launch_bounds__(256) void pack_qkv_weight_h16_kernel(const float *__restrict__ w, unsigned short *__restrict__ o, int rows) {
    const int gid = blockIdx.x * 256 + threadIdx.x;                  // one 16-byte piece: (tile, k-step, h, rho)
    const int rho = gid & 31, h = (gid >> 5) & 1, ks = (gid >> 6) % (AD / 16), tile = gid / (64 * (AD / 16));
    if (tile * 32 >= rows) return;
    const int ch = 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
    const float *src = w + (int64_t)(tile * 32 + ch) * AD + ks * 16 + h * 8;
    u32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = HM<HT>::pack2(src[2 * e], src[2 * e + 1]);
    *reinterpret_cast<u32x4 *>(o + (int64_t)gid * 8) = v;
}
}  // namespace

// [rows][C] fp32 weight of a 1 x 1 convolution -> transposed-product fragment order (rows % 32 == 0, C == 256)
extern "C" int sgam_pack_weight_tp_h16(const float *w, void *w_frag, int32_t ht, int32_t rows, int32_t C, void *stream) {
    if (!w || !w_frag || C != AD || rows < 32 || rows % 32 != 0 || (ht != 0 && ht != 1)) return SGAM_EINVAL;
    if (!sgam_aligned16(w_frag)) return SGAM_EALIGN;
    const int pieces = rows * AD / 8;
    hipStream_t s = sgam_stream(stream);
    if (ht == 0) SGAM_KLAUNCH(pack_qkv_weight_h16_kernel<0>, dim3(pieces / 256), dim3(256), 0, s, w, (unsigned short *)w_frag, rows);
    else SGAM_KLAUNCH(pack_qkv_weight_h16_kernel<1>, dim3(pieces / 256), dim3(256), 0, s, w, (unsigned short *)w_frag, rows);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_pack_qkv_weight_h16(const float *w, void *w_frag, int32_t ht, int32_t C, void *stream) {
    return sgam_pack_weight_tp_h16(w, w_frag, ht, 3 * C, C, stream);
}

extern "C" int sgam_groupnorm_table_from_partials(const double *partial, int32_t nchunk, const float *gamma, const float *beta,
                                                  float *scale_shift, int32_t B, int32_t HW, int32_t C, int32_t groups, float eps,
                                                  void *stream);

extern "C" int64_t sgam_attn_block_h16_workspace_bytes(int32_t n, int32_t C, int32_t B) {
    const int64_t base = sgam_attention_h16_batched_workspace_bytes(n, C, B);
    if (base < 0) return -1;
    // + q [B n][C] 16-bit + the {scale, shift} table [B][C][2] fp32
    return base + (int64_t)B * n * AD * 2 + (int64_t)B * AD * 2 * 4;
}

// out = softmax(q k^T scale) v with q | k | v = GroupNorm(x) Wqkv^T + b — everything of the AttnBlock ahead of proj_out, 4 launches
// (table, projection, flash, merge).  x [B n][ldx] 16-bit; gn_partial / nchunk: the chunk statistics x's producer left
// ([B][nchunk][32][2] fp64, or nchunk = 0: its accumulator record); gamma, beta [C]; w_frag: sgam_pack_qkv_weight_h16; bias [3 C].
static int attn_block_h16_impl(const void *x, int32_t ldx, const double *gn_partial, int32_t nchunk, const float *gamma, const float *beta,
                               float eps, const void *w_frag, const float *bias, int32_t ht, int32_t n, int32_t C, int32_t B, float scale,
                               const void *wp_frag, const float *bp, double *gn_partial_out, void *out, int32_t ldo, void *workspace,
                               int64_t workspace_bytes, void *stream) {
    if (!x || !gn_partial || !gamma || !beta || !w_frag || !bias || !out || !workspace || (ht != 0 && ht != 1)) return SGAM_EINVAL;
    if (wp_frag && (!sgam_aligned16(wp_frag) || (bp && !sgam_aligned16(bp)) || (gn_partial_out && !sgam_aligned16(gn_partial_out)) || ldo % 8 != 0))
        return SGAM_EALIGN;
    const int64_t need = sgam_attn_block_h16_workspace_bytes(n, C, B);
    if (need < 0) return SGAM_EINVAL;
    const int nsplit = attn_nsplit(n, B), nt = B * n;
    if (ldx < C || ldx % 8 != 0 || ldo < C || ldo % 4 != 0 || !(scale > 0.f) || !(eps > 0.f) || (n / KB) % nsplit != 0 || n % 64 != 0)
        return SGAM_EINVAL;
    if (workspace_bytes < need) return SGAM_EWORKSPACE;
    if (!sgam_aligned16(x) || !sgam_aligned16(w_frag) || !sgam_aligned16(bias) || !sgam_aligned16(workspace) || (((uintptr_t)out) & 7u) != 0)
        return SGAM_EALIGN;
    hipStream_t s = sgam_stream(stream);
    unsigned short *kf = (unsigned short *)workspace;
    unsigned short *vf = kf + (int64_t)nt * AD;
    float *ws_o = (float *)(vf + (int64_t)nt * AD);
    float *ws_ml = ws_o + (int64_t)nsplit * nt * AD;
    unsigned short *qb = (unsigned short *)(ws_ml + (int64_t)nsplit * nt * 2);
    float *table = (float *)(qb + (int64_t)nt * AD);
    // (the {scale, shift} table is its own small launch: folding the chunk records inside every workgroup of the projection instead
    // measured break-even to slower in round 5 — a dependent round trip + fp64 arithmetic in front of every workgroup's staging)
    {
        const int rc = sgam_groupnorm_table_from_partials(gn_partial, nchunk, gamma, beta, table, B, n, C, 32, eps, stream);
        if (rc != SGAM_OK) return rc;
    }
    QkvHParams g;
    g.x = (const unsigned short *)x; g.table = table; g.w = (const unsigned short *)w_frag; g.bias = bias; g.q = qb; g.kf = kf; g.vf = vf;
    g.ldx = ldx; g.n_img = n;
    if (sgam_i_prof_on) sgam_i_prof_shape(nt, 3 * AD, AD, 1);
    if (sgam_i_prof_on) sgam_i_prof_work(2.0 * nt * 3.0 * AD * AD, 2.0 * (4.0 * nt * AD + 3.0 * AD * AD));
    if (ht == 0) SGAM_KLAUNCH(attn_qkv_gn_h16_kernel<0>, dim3(nt / 64, 6), dim3(256), 0, s, g);
    else SGAM_KLAUNCH(attn_qkv_gn_h16_kernel<1>, dim3(nt / 64, 6), dim3(256), 0, s, g);
    SGAM_LAUNCH_CHECK();
    AttnHParams p;
    p.q_frag = 1; p.q = qb; p.kf = kf; p.vf = vf; p.ws_o = ws_o; p.ws_ml = ws_ml;
    p.ld = AD; p.n = nt; p.n_img = n; p.nsplit = nsplit; p.blocks_per_split = n / KB / nsplit; p.qscale_log2e = scale * LOG2E;
    const dim3 grid(nt / 128 * nsplit), cgrid(nt / 32 * 8);
    if (sgam_i_prof_on) sgam_i_prof_work(4.0 * B * n * (double)n * AD, 4.0 * 2.0 * nt * AD);
    if (ht == 0) SGAM_KLAUNCH(attn_flash_h16_kernel<0>, grid, dim3(256), 0, s, p);
    else SGAM_KLAUNCH(attn_flash_h16_kernel<1>, grid, dim3(256), 0, s, p);
    SGAM_LAUNCH_CHECK();
    if (wp_frag) {                                                    // merge + proj_out + residual (= x) in one launch
        CombProjHParams c;
        c.ws_o = (const unsigned short *)ws_o; c.ws_ml = ws_ml; c.w = (const unsigned short *)wp_frag; c.bias = bp;
        c.res = (const unsigned short *)x; c.out = (unsigned short *)out; c.gn_partial = gn_partial_out; c.n = nt; c.ldr = ldx; c.ldc = ldo;
        if (sgam_i_prof_on) sgam_i_prof_shape(nt, AD, AD, 1);
        if (sgam_i_prof_on) sgam_i_prof_work(2.0 * nt * (double)AD * AD, 2.0 * ((double)nsplit * nt * AD + 2.0 * nt * AD + (double)AD * AD));
#define HCOMBPROJ(HT_, NS_) SGAM_KLAUNCH((attn_combine_proj_h16_kernel<HT_, NS_>), dim3(nt / 32), dim3(256), 0, s, c)
        switch (nsplit * 2 + (ht ? 1 : 0)) {
            case 16: HCOMBPROJ(0, 8); break;
            case 17: HCOMBPROJ(1, 8); break;
            case 8: HCOMBPROJ(0, 4); break;
            case 9: HCOMBPROJ(1, 4); break;
            case 4: HCOMBPROJ(0, 2); break;
            case 5: HCOMBPROJ(1, 2); break;
            case 2: HCOMBPROJ(0, 1); break;
            case 3: HCOMBPROJ(1, 1); break;
            default: return SGAM_EINVAL;
        }
#undef HCOMBPROJ
        SGAM_LAUNCH_CHECK();
        return SGAM_OK;
    }
#define HCOMBINE2(HT_, NS_) SGAM_KLAUNCH((attn_combine_h16_kernel<HT_, NS_>), cgrid, dim3(256), 0, s, ws_o, ws_ml, (unsigned short *)out, ldo, nt)
    switch (nsplit * 2 + (ht ? 1 : 0)) {
        case 16: HCOMBINE2(0, 8); break;
        case 17: HCOMBINE2(1, 8); break;
        case 8: HCOMBINE2(0, 4); break;
        case 9: HCOMBINE2(1, 4); break;
        case 4: HCOMBINE2(0, 2); break;
        case 5: HCOMBINE2(1, 2); break;
        case 2: HCOMBINE2(0, 1); break;
        case 3: HCOMBINE2(1, 1); break;
        default: return SGAM_EINVAL;
    }
#undef HCOMBINE2
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

// ---- the whole AttnBlock of the split-fp32 path in three launches (ABI v9) -------------------------------------------------------
extern "C" int64_t sgam_attn_block_f32x_workspace_bytes(int32_t n, int32_t C, int32_t B) {
    const int64_t base = sgam_attention_f32x_batched_workspace_bytes(n, C, B);
    return base < 0 ? -1 : base + (int64_t)B * n * AD * 4;            // + q [B n][C] fp32
}

// out = x + proj_out(attention(q, k, v)), q | k | v = GroupNorm(x) Wqkv^T + b: fused front end (attn_qkv_gn_f32x_kernel), flash kernel,
// merge + proj_out + residual (attn_combine_proj_f32x_kernel).  wqkv_planes: sgam_split_rows_f32x of the stacked [3 C][C] weight whose
// rows were permuted inside every 32-row tile so that row 8 j + 4 h + i holds channel 16 h + 4 j + i; bqkv in natural order.
static int attn_block_f32x_impl(const float *x, int32_t ldx, const float *mean_rstd,
                                const float *gamma, const float *beta, const void *wqkv_planes, float wqkv_scale, const float *bqkv, int32_t n, int32_t C, int32_t B, float scale,
                                    const void *wp_planes, float wp_scale, const float *bp, float *out, int32_t ldc, double *gn_partial,
                                    void *workspace, int64_t workspace_bytes, void *stream) {
    if (!x || !mean_rstd || !gamma || !beta || !wqkv_planes || !bqkv || !wp_planes || !out || !workspace ||
        !(wqkv_scale > 0.f) || !(wp_scale > 0.f))
        return SGAM_EINVAL;
    const int64_t need = sgam_attn_block_f32x_workspace_bytes(n, C, B);
    if (need < 0 || ldx < C || ldx % 4 != 0 || ldc < C || n % 64 != 0) return SGAM_EINVAL;
    int ex;
    if (!(scale > 0.f) || frexpf(scale, &ex) != 0.5f) return SGAM_EINVAL;
    if (workspace_bytes < need) return SGAM_EWORKSPACE;
    if (!sgam_aligned16(x) || !sgam_aligned16(gamma) || !sgam_aligned16(beta) || !sgam_aligned16(wqkv_planes) || !sgam_aligned16(bqkv) ||
        !sgam_aligned16(wp_planes) || !sgam_aligned16(workspace) || (gn_partial && !sgam_aligned16(gn_partial)))
        return SGAM_EALIGN;
    const int nsplit = attn_nsplit(n, B), nt = B * n;
    if ((n / KB) % nsplit != 0) return SGAM_EINVAL;
    hipStream_t s = sgam_stream(stream);
    unsigned short *kf = (unsigned short *)workspace;
    unsigned short *vf = kf + (int64_t)nt * AD * 2;
    float *ws_o = (float *)(vf + (int64_t)nt * AD * 2);
    float *ws_ml = ws_o + (int64_t)nsplit * nt * AD;
    unsigned short *qb = reinterpret_cast<unsigned short *>(ws_ml + (int64_t)nsplit * nt * 2);      // [nt][AD] x (hi + lo): the same bytes as fp32 rows
    QkvXParams g;
    g.x = x; g.mean_rstd = mean_rstd;
    g.gamma = gamma; g.beta = beta; g.w = (const unsigned short *)wqkv_planes; g.bias = bqkv;
    g.qf = qb; g.kf = kf; g.vf = vf; g.range_flag = sgam_i_range_flag; g.ldx = ldx; g.n_img = n; g.inv_w_scale = 1.0f / wqkv_scale;
    g.qscale = scale;
    if (sgam_i_prof_on) sgam_i_prof_shape(nt, 3 * AD, AD, 1);
    if (sgam_i_prof_on) sgam_i_prof_work(2.0 * nt * 3.0 * AD * AD, 4.0 * (4.0 * nt * AD + 3.0 * AD * AD));
    SGAM_KLAUNCH(attn_qkv_gn_f32x_kernel, dim3(nt / 64, 6), dim3(256), 0, s, g);
    SGAM_LAUNCH_CHECK();
    AttnParams p;
    p.qf = qb; p.q = nullptr; p.kf = kf; p.vf = vf; p.ws_o = ws_o; p.ws_ml = ws_ml;
    p.ld = AD; p.n = nt; p.n_img = n; p.nsplit = nsplit; p.blocks_per_split = n / KB / nsplit; p.qscale = scale;
    if (sgam_i_prof_on) sgam_i_prof_work(4.0 * B * n * (double)n * AD, 4.0 * 4.0 * nt * AD);
    SGAM_KLAUNCH(attn_flash_f32x_kernel, dim3(nt / 128 * nsplit), dim3(256), 0, s, p);
    SGAM_LAUNCH_CHECK();
    CombProjParams c;
    c.ws_o = ws_o; c.ws_ml = ws_ml; c.w = (const unsigned short *)wp_planes; c.bias = bp; c.res = x; c.out = out;
    c.gn_partial = gn_partial; c.n_img = n;
    c.range_flag = sgam_i_range_flag; c.n = nt; c.ns = nsplit; c.ldr = ldx; c.ldc = ldc;
    c.inv_w_scale = 1.0f / wp_scale;
    if (sgam_i_prof_on) sgam_i_prof_shape(nt, AD, AD, 1);
    if (sgam_i_prof_on) sgam_i_prof_work(2.0 * nt * (double)AD * AD, 4.0 * ((double)nsplit * nt * AD + 2.0 * nt * AD + (double)AD * AD));
    switch (nsplit) {
        case 8: SGAM_KLAUNCH(attn_combine_proj_f32x_kernel<8>, dim3(nt / 32), dim3(256), 0, s, c); break;
        case 4: SGAM_KLAUNCH(attn_combine_proj_f32x_kernel<4>, dim3(nt / 32), dim3(256), 0, s, c); break;
        case 2: SGAM_KLAUNCH(attn_combine_proj_f32x_kernel<2>, dim3(nt / 32), dim3(256), 0, s, c); break;
        case 1: SGAM_KLAUNCH(attn_combine_proj_f32x_kernel<1>, dim3(nt / 32), dim3(256), 0, s, c); break;
        default: return SGAM_EINVAL;
    }
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_attn_block_h16(const void *x, int32_t ldx, const double *gn_partial, int32_t nchunk, const float *gamma, const float *beta,
                                   float eps, const void *w_frag, const float *bias, int32_t ht, int32_t n, int32_t C, int32_t B, float scale,
                                   void *out, int32_t ldo, void *workspace, int64_t workspace_bytes, void *stream) {
    return attn_block_h16_impl(x, ldx, gn_partial, nchunk, gamma, beta, eps, w_frag, bias, ht, n, C, B, scale, nullptr, nullptr, nullptr, out, ldo,
                               workspace, workspace_bytes, stream);
}

// the WHOLE 16-bit AttnBlock: sgam_attn_block_h16 with the merge of the key ranges fused into proj_out + residual (= x):
// out = x + proj_out(attention) [B n][ldo] 16-bit; wp_frag: sgam_pack_weight_tp_h16 of proj_out's [C][C] weight; bp [C] or NULL;
// gn_partial_out (optional): [B][n / 32][32][2] fp64 chunk statistics of `out` for the GroupNorm that follows
extern "C" int sgam_attn_block_proj_h16(const void *x, int32_t ldx, const double *gn_partial, int32_t nchunk, const float *gamma,
                                        const float *beta, float eps, const void *w_frag, const float *bias, int32_t ht, int32_t n, int32_t C,
                                        int32_t B, float scale, const void *wp_frag, const float *bp, double *gn_partial_out, void *out,
                                        int32_t ldo, void *workspace, int64_t workspace_bytes, void *stream) {
    if (!wp_frag) return SGAM_EINVAL;
    return attn_block_h16_impl(x, ldx, gn_partial, nchunk, gamma, beta, eps, w_frag, bias, ht, n, C, B, scale, wp_frag, bp, gn_partial_out, out,
                               ldo, workspace, workspace_bytes, stream);
}

// ---- the small AttnBlocks (n = 128 / 256 tokens per image, C = 512) in one launch (ABI v9) ------------------------------------------
extern "C" int32_t sgam_attention_small_f32x_fits(int32_t n, int32_t C, int32_t B) {
    return (C == 512 && (n == 128 || n == 256) && B >= 1 && (int64_t)B * n < (1 << 24)) ? 1 : 0;
}

extern "C" int sgam_attention_small_f32x(const float *q, const float *k, const float *v, int32_t ld, int32_t n, int32_t C, int32_t B,
                                         float scale, float *out, int32_t ldo, void *stream) {
    if (!q || !k || !v || !out || sgam_attention_small_f32x_fits(n, C, B) != 1 || ld < C || ld % 4 != 0 || ldo < C || ldo % 4 != 0 ||
        !(scale > 0.f))
        return SGAM_EINVAL;
    if (!sgam_aligned16(q) || !sgam_aligned16(k) || !sgam_aligned16(v) || !sgam_aligned16(out)) return SGAM_EALIGN;
    SmallAttnParams p;
    p.q = q; p.k = k; p.v = v; p.out = out; p.ld = ld; p.ldo = ldo; p.n_img = n; p.scale = scale; p.range_flag = sgam_i_range_flag;
    const dim3 grid(B * n / 32, C / 128);
    hipStream_t s = sgam_stream(stream);
    if (sgam_i_prof_on) sgam_i_prof_work(4.0 * B * n * (double)n * C, 4.0 * 4.0 * B * n * C);
    // n = 256: eight wavefronts of one key tile each (the operand splits are a serial VALU chain: two wavefronts per SIMD halve it)
    if (n == 256) SGAM_KLAUNCH((attn_small_f32x_kernel<512, 1, 8>), grid, dim3(512), 0, s, p);
    else SGAM_KLAUNCH((attn_small_f32x_kernel<512, 1, 4>), grid, dim3(256), 0, s, p);
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_attention_small_h16(const void *q, const void *k, const void *v, int32_t ht, int32_t ld, int32_t n, int32_t C, int32_t B,
                                        float scale, void *out, int32_t ldo, void *stream) {
    if (!q || !k || !v || !out || (ht != 0 && ht != 1) || sgam_attention_small_f32x_fits(n, C, B) != 1 || ld < C || ld % 8 != 0 || ldo < C ||
        ldo % 4 != 0 || !(scale > 0.f))
        return SGAM_EINVAL;
    if (!sgam_aligned16(q) || !sgam_aligned16(k) || !sgam_aligned16(v) || (((uintptr_t)out) & 7u) != 0) return SGAM_EALIGN;
    SmallAttnHParams p;
    p.q = (const unsigned short *)q; p.k = (const unsigned short *)k; p.v = (const unsigned short *)v; p.out = (unsigned short *)out;
    p.ld = ld; p.ldo = ldo; p.n_img = n; p.scale = scale;
    const dim3 grid(B * n / 32, C / 128);
    hipStream_t s = sgam_stream(stream);
    if (sgam_i_prof_on) sgam_i_prof_work(4.0 * B * n * (double)n * C, 2.0 * 4.0 * B * n * C);
#define HSMALL(HT_, NW_) SGAM_KLAUNCH((attn_small_h16_kernel<HT_, 512, 1, NW_>), grid, dim3(64 * NW_), 0, s, p)
    if (ht == 0) { if (n == 256) HSMALL(0, 8); else HSMALL(0, 4); }
    else { if (n == 256) HSMALL(1, 8); else HSMALL(1, 4); }
#undef HSMALL
    SGAM_LAUNCH_CHECK();
    return SGAM_OK;
}

extern "C" int sgam_attn_block_f32x(const float *x, int32_t ldx, const float *mean_rstd, const float *gamma, const float *beta,
                                    const void *wqkv_planes, float wqkv_scale, const float *bqkv, int32_t n, int32_t C, int32_t B, float scale,
                                    const void *wp_planes, float wp_scale, const float *bp, float *out, int32_t ldc, double *gn_partial,
                                    void *workspace, int64_t workspace_bytes, void *stream) {
    return attn_block_f32x_impl(x, ldx, mean_rstd, gamma, beta, wqkv_planes, wqkv_scale, bqkv, n, C, B, scale, wp_planes, wp_scale,
                                bp, out, ldc, gn_partial, workspace, workspace_bytes, stream);
}
