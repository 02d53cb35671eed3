/*
 * sgam_hip.h — C ABI of libsgam_hip.so, the MI355X (gfx950) backend of SGAM's per-step
 * generative-sensing hot path.
 *
 * Conventions (SURVEY.md §8b "lower boundary"):
 *   - extern "C", plain device pointers + sizes, no torch / C++ types;
 *   - the CALLER owns every buffer (the Python side allocates them with torch);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream), never synchronises, never allocates device memory;
 *   - return 0 on success, a negative SGAM_E* code on a bad argument, or a positive
 *     hipError_t if the launch itself failed;
 *   - activations inside the VQGAN are NHWC ("pixel-major") fp32: [B][H][W][C];
 *     the NCHW <-> NHWC hops at the module boundary are sgam_nchw_to_nhwc / _nhwc_to_nchw.
 *
 * Each entry point cites the reference op (file:line under /root/reference) it replaces.
 */
#ifndef SGAM_HIP_H
#define SGAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGAM_OK 0
#define SGAM_EINVAL (-1)   /* bad shape / unsupported size */
#define SGAM_EALIGN (-2)   /* pointer or stride not aligned as required */
#define SGAM_EWORKSPACE (-3)

/* ABI version of this header; bumped on any signature change. */
int sgam_abi_version(void);
/* Human-readable build string ("gfx950 ... commit <c>; digest <d>"). */
const char *sgam_build_info(void);
/* (ABI v10) The build stamp alone: the last commit that touched the library's sources ("+dirty" when the tree differed), and the
 * first 12 hex digits of a sha256 over every source, header and compile flag (sgam_neurips22_amd/build.py). */
const char *sgam_build_commit(void);
const char *sgam_build_digest(void);

/* ------------------------------------------------------------------------------------------
 * Kernel timeline (measurement; no reference counterpart — the reference has no profiling).  While
 * enabled, every kernel launch of the library is bracketed by two HIP events recorded on the launch stream.  Not usable
 * inside stream capture; the caller synchronises the device before reading.  sgam_prof_get(i): kernel name as written at
 * the launch site, the enclosing function's signature (resolves symbolic template arguments), elapsed ms, and the
 * algorithmic FLOP / bytes the launch site announced (0 when it announced none).  sgam_prof_mark_empty records a
 * bracket around nothing (the cost of the bracket itself, to be subtracted).
 * ------------------------------------------------------------------------------------------ */
int sgam_prof_enable(int32_t on);
int sgam_prof_mark_empty(void *stream);
int32_t sgam_prof_count(void);
int sgam_prof_get(int32_t i, const char **kernel, const char **where, float *ms, double *flops, double *bytes);
/* GEMM view of record i where the launch site announced one: mnks[4] = {M, N, K, split-K factor}, zeros otherwise */
int sgam_prof_get_shape(int32_t i, int32_t *mnks);

/* ------------------------------------------------------------------------------------------
 * K1/K2/K3/K6 — convolution as implicit GEMM on the matrix cores (fp32-in/fp32-acc MFMA).
 * Replaces torch.nn.Conv2d in ResnetBlock/Upsample/Downsample/AttnBlock/VQModel:
 *   sgam/generative_sensing_module/modules/diffusionmodules/model.py:43-53 (nearest x2 + 3x3),
 *   :63-75 (pad (0,1,0,1) + 3x3 stride 2), :88-116 (3x3 / 1x1 nin_shortcut), :146-165 (1x1 q,k,v,proj)
 *   sgam/generative_sensing_module/model.py:54,62,63 (1x1 conv_in, quant_conv, post_quant_conv).
 *
 *   out[m][n] = bias[n] + residual[m][n] + sum_{ky,kx,c} X[b][iy][ix][c] * Wp[n][(ky*KW+kx)*Cin + c]
 *   with m = (b*Ho + oy)*Wo + ox, iy = oy*stride + ky - pad_t, ix = ox*stride + kx - pad_l,
 *   zero outside the (logical) input; when `upsample2x` != 0 the logical input is the nearest-
 *   neighbour 2x enlargement of the physical [Hi][Wi] map (X[b][iy>>1][ix>>1]).
 *
 *   x        [B][Hi][Wi] pixels, `lda` floats apart, first Cin floats used (Cin % 4 == 0; 32-wide K slabs, ragged tail masked)
 *   w_packed [N] rows, `ldb` floats apart, K = KH*KW*Cin contiguous (see sgam_pack_conv_weight)
 *   bias     [N] or NULL;  residual [M] rows `ldr` apart or NULL
 *   out      [M] rows `ldc` apart; only columns n < n_valid are written
 *   workspace: sgam_conv2d_workspace_bytes(...) bytes (split-K partial sums), may be NULL if 0.
 * ------------------------------------------------------------------------------------------ */
typedef struct sgam_conv_desc {
    int32_t B, Hi, Wi, Cin;       /* physical input */
    int32_t Ho, Wo, N;            /* output spatial size and channel count */
    int32_t KH, KW, stride;
    int32_t pad_t, pad_l;         /* top/left zero padding (bottom/right implied by Ho/Wo) */
    int32_t upsample2x;           /* 1: fuse F.interpolate(scale 2, nearest) in front of the conv */
    int32_t lda, ldb, ldc, ldr;   /* row strides in floats of x, w_packed, out, residual */
    int32_t n_valid;              /* columns actually stored (<= N) */
    int32_t bias_per_row;         /* 0: bias[n]; 1: bias[m] (used for the transposed V projection) */
    /* optional plan override (autotuner, sgam_neurips22_amd/tune.py); 0 = built-in heuristic.
     * (plan_bm, plan_bn) in {(128,128), (64,128), (64,64)}; plan_ksplit >= 1. */
    int32_t plan_bm, plan_bn, plan_ksplit;
} sgam_conv_desc;

int64_t sgam_conv2d_workspace_bytes(const sgam_conv_desc *d);
/* which kernel instantiation a descriptor maps to: workgroup tile BMxBN and split-K factor (for profiling). */
int sgam_conv2d_plan(const sgam_conv_desc *d, int32_t *bm, int32_t *bn, int32_t *ksplit);
int sgam_conv2d_nhwc_f32(const sgam_conv_desc *d, const float *x, const float *w_packed,
                         const float *bias, const float *residual, float *out, void *workspace,
                         int64_t workspace_bytes, void *stream);
/* Same, with the GroupNorm(+swish) of the INPUT fused into the operand staging (ResnetBlock norm1/norm2,
 * AttnBlock.norm, norm_out: diffusionmodules/model.py:119-127, 170, 429-430, 536-537): x is the RAW activation,
 * gn_scale_shift the [B][Cin][2] table of sgam_groupnorm_stats_nhwc_f32 (NULL = plain conv), gn_swish 0/1.
 * Zero padding stays zero (it pads the normalised map, as in the reference). */
int sgam_conv2d_gn_nhwc_f32(const sgam_conv_desc *d, const float *x, const float *gn_scale_shift,
                            int32_t gn_swish, const float *w_packed, const float *bias, const float *residual,
                            float *out, void *workspace, int64_t workspace_bytes, void *stream);

/* [Cout][Cin][KH][KW] (torch Conv2d.weight) -> [Cout_pad][KH*KW][Cin_pad], zero padded. */
int sgam_pack_conv_weight(const float *w_oihw, float *w_packed, int32_t Cout, int32_t Cin, int32_t KH,
                          int32_t KW, int32_t Cout_pad, int32_t Cin_pad, void *stream);

/* ------------------------------------------------------------------------------------------
 * "Split" fp32 path (conv_f32x.hip): the same fp32 convolution / GEMM as sgam_conv2d_nhwc_f32 (same descriptor, fp32
 * activations in and out) evaluated on the fp16 matrix cores with fp32-class accuracy: every operand is split
 * exactly into hi + lo fp16 pieces and a.b = a_hi.b_hi + a_hi.b_lo + a_lo.b_hi is accumulated in fp32 (products of
 * fp16 values are exact in fp32; the dropped term is <= 2^-22 |a||b|).  Held to the same parity tests as the
 * fp32-MFMA path.  Weights are pre-split by sgam_pack_conv_weight_f32x into fp16 hi / lo halves of w_scale * w
 * (w_scale a power of two lifting max|w| into (512, 1024]) stored in MFMA-fragment order: [N / 32][ldb / 32][256][8]
 * halfs, per (32-row tile, 32-element K slab) 256 pieces of 16 bytes with piece = ((plane * 2 + k-step) * 2 + k-half)
 * * 32 + row (plane 0 = hi, 1 = lo; k-step = 16 elements, k-half = 8), so the B operand of one
 * v_mfma_f32_32x32x16_f16 is one contiguous kilobyte (N rounded up to 32 rows); activations are split while staged,
 * after multiplication by the power of two a_scale (1 for ordinary activations, 1024 for softmax probabilities).
 * ldb = K elements per row, % 32 == 0 (the buffer holds 2 * ldb halfs per row of the 32-row-padded matrix); Cin % 8 == 0, and % 32 == 0 when
 * KH * KW > 1; |a_scale * x| must stay below 65504.
 * ------------------------------------------------------------------------------------------ */
/* Range guard of the split path.  An activation with |a_scale * x| >= 65520 becomes inf in its fp16 hi half, and the
 * output it feeds becomes inf / NaN.  Every split-fp32 kernel that produces final outputs (conv / GEMM epilogue, split-K
 * combine, fused-attention combine) ORs 1 into *device_flag when it writes a non-finite value.  The caller owns the int32
 * (zero it, read it at a synchronisation point); on 1 the results since the last check are invalid in this mode and
 * must be recomputed on the fp32-in MFMA path (sgam_conv2d_nhwc_f32), which has no range precondition — the Python
 * side does exactly that (VQModel.forward, InfiniteSceneGeneration.scene_expansion).  NULL (default) disables the
 * reporting.  This pointer is the library's only mutable global: one flag per process (= per GPU). */
int sgam_f32x_set_range_flag(int32_t *device_flag);
int64_t sgam_conv2d_f32x_workspace_bytes(const sgam_conv_desc *d);
int sgam_conv2d_f32x_plan(const sgam_conv_desc *d, int32_t *bm, int32_t *bn, int32_t *ksplit);
int sgam_conv2d_nhwc_f32x(const sgam_conv_desc *d, const float *x, float a_scale, const void *w_planes,
                          float w_scale, const float *bias, const float *residual, float *out, void *workspace,
                          int64_t workspace_bytes, void *stream);
int sgam_pack_conv_weight_f32x(const float *w_oihw, void *w_planes, float w_scale, int32_t Cout, int32_t Cin,
                               int32_t KH, int32_t KW, int32_t Cout_pad, int32_t Cin_pad, void *stream);
/* Same convolution, additionally delivering the GroupNorm statistics of its OUTPUT (32 groups) as per-chunk {sum, sumsq}
 * partial sums: gn_partial holds [B][chunks][32][2] doubles, chunks = sgam_conv2d_f32x_stats_chunks(d) per image — one
 * per (tile, wavefront row) from the conv epilogue when the plan has no split-K, one per 1024 outputs from the split-K
 * combine otherwise; 0 = not available for this shape (N % 128 != 0, n_valid != N, tiles straddling images, N not a
 * divisor of 1024 under split-K).  sgam_groupnorm_stats_from_partials_f32 folds them (one 32-workgroup launch);
 * sgam_conv2d_f32x_stats_mode(d) = chunks > 0. */
int32_t sgam_conv2d_f32x_stats_chunks(const sgam_conv_desc *d);
int32_t sgam_conv2d_f32x_stats_mode(const sgam_conv_desc *d);
int sgam_conv2d_stats_nhwc_f32x(const sgam_conv_desc *d, const float *x, float a_scale, const void *w_planes,
                                float w_scale, const float *bias, const float *residual, float *out,
                                double *gn_partial, void *workspace, int64_t workspace_bytes, void *stream);
int sgam_groupnorm_from_partials_f32(const float *x, const double *partial, int32_t nchunk, const float *gamma,
                                     const float *beta, float *y, int32_t B, int32_t HW, int32_t C, int32_t groups,
                                     float eps, int32_t fuse_swish, void *workspace, int64_t workspace_bytes, void *stream);
/* GroupNorm(+swish) of the INPUT fused into the operand staging: x is normalised with the per-(image, group)
 * {mean, rstd} gn_mean_rstd [B][32][2] and the affine parameters gn_gamma / gn_beta [Cin] (16-byte aligned) while the
 * 3x3 kernel stages each channel slab of its input patch — once per slab, shared by the nine taps — so no normalised
 * copy of the activation is ever written.  Zero padding applies to the normalised tensor, as in Conv2d(GroupNorm(x)).
 * Available when sgam_conv2d_f32x_gn_fusable(d) == 1 (3x3, stride 1, pad 1, no upsampling, Ho % 8 == 0, Wo % 8 (16)
 * == 0, Cin % 128 == 0, a halo-kernel tile plan).  {mean, rstd} come from sgam_groupnorm_stats_from_partials_f32 (the
 * producing convolution's partial sums) or from sgam_groupnorm_meanrstd_nhwc_f32 (any tensor).
 * gn_partial as in sgam_conv2d_stats_nhwc_f32x, may be NULL. */
int32_t sgam_conv2d_f32x_gn_fusable(const sgam_conv_desc *d);
/* 1 when the descriptor runs on a halo-staged 3x3 kernel (3x3 / stride 1 / pad 1 on the input or on its nearest-2x
 * upsampling), 0 when it runs on the generic implicit-GEMM kernel (diagnostic: profiling labels) */
int32_t sgam_conv2d_f32x_uses_halo(const sgam_conv_desc *d);
int sgam_conv2d_gn_nhwc_f32x(const sgam_conv_desc *d, const float *x, const float *gn_mean_rstd, const float *gn_gamma,
                             const float *gn_beta, int32_t gn_swish, const void *w_planes, float w_scale, const float *bias,
                             const float *residual, float *out, double *gn_partial, void *workspace, int64_t workspace_bytes,
                             void *stream);
/* The same with the statistics of x still as its PRODUCER's chunk partials ([B][chunks_in][32][2] fp64 {sum, sumsq}, what
 * sgam_conv2d_stats_nhwc_f32x / the split-K combine leave): the convolution folds them itself, no fold launch in between.  Offered
 * where the fold is cheap — sgam_conv2d_f32x_gn_foldable(d, chunks_in) == 1: at most 16 chunks (the group-major combine of the
 * 16^2 / 32^2 maps leaves 8 - 16) and at most two channel slabs per workgroup. */
int32_t sgam_conv2d_f32x_gn_foldable(const sgam_conv_desc *d, int32_t chunks_in);
int sgam_conv2d_gnp_nhwc_f32x(const sgam_conv_desc *d, const float *x, const double *gn_partial_in, int32_t chunks_in, float eps,
                              const float *gn_gamma, const float *gn_beta, int32_t gn_swish, const void *w_planes, float w_scale,
                              const float *bias, const float *residual, float *out, double *gn_partial, void *workspace,
                              int64_t workspace_bytes, void *stream);
int sgam_groupnorm_stats_from_partials_f32(const double *partial, int32_t nchunk, float *mean_rstd, int32_t B, int32_t HW,
                                           int32_t C, int32_t groups, float eps, void *stream);
int sgam_groupnorm_meanrstd_nhwc_f32(const float *x, float *mean_rstd, int32_t B, int32_t HW, int32_t C, int32_t groups,
                                     float eps, void *workspace, int64_t workspace_bytes, void *stream);
/* split a row-major fp32 matrix [N][K] (row stride ld) into the fragment-ordered B-operand layout over [Np][Kp], both
 * rounded up to 32 (zero filled): the B operand when it is an activation (ldb = Kp) */
int sgam_split_rows_f32x(const float *x, void *planes, float scale, int32_t N, int32_t K, int32_t ld, void *stream);

/* ------------------------------------------------------------------------------------------
 * 16-bit THROUGHPUT path (h16.hip): the same operators on bf16 (`ht` = 0) or fp16 (`ht` = 1) activations and
 * weights, fp32 accumulation on v_mfma_f32_32x32x16_{bf16,f16}; statistics / softmax / bias / residual math in
 * fp32.  Strides (lda, ldb, ldc, ldr) are in ELEMENTS of the respective buffer; Cin % 8 == 0.  `out_f32` != 0
 * keeps the result in fp32 (attention scores, the latent handed to the fp32 quantiser, the final RGB-D).
 * Not a parity path: index agreement with the fp32 path is reported, not guaranteed.
 * ------------------------------------------------------------------------------------------ */
int64_t sgam_conv2d_h16_workspace_bytes(const sgam_conv_desc *d);
int sgam_conv2d_h16_plan(const sgam_conv_desc *d, int32_t *bm, int32_t *bn, int32_t *ksplit);
int sgam_conv2d_nhwc_h16(const sgam_conv_desc *d, int32_t ht, const void *x, const void *w_packed,
                         const float *bias, const void *residual, void *out, int32_t out_f32, void *workspace,
                         int64_t workspace_bytes, void *stream);
/* (ABI v5) The same launch with the GroupNorm statistics of the OUTPUT left as per-chunk partial sums (gn_partial
 * [B][sgam_conv2d_h16_generic_stats_chunks(d)][32][2] doubles, the layout of sgam_conv2d_h16_stats_chunks): the 1x1 / strided
 * convolutions and the attention block's proj_out (diffusionmodules/model.py:63-75, 168-192) then feed the next Normalize
 * without a statistics pass.  Available (chunks > 0) when the kernel runs whole-K workgroups with its direct epilogue:
 * n_valid == N, N % 128 == 0, even row strides, no per-row bias, Ho * Wo a multiple of half the tile's rows. */
int32_t sgam_conv2d_h16_generic_stats_chunks(const sgam_conv_desc *d);
int sgam_conv2d_stats_nhwc_h16(const sgam_conv_desc *d, int32_t ht, const void *x, const void *w_packed, const float *bias,
                               const void *residual, void *out, int32_t out_f32, double *gn_partial, void *workspace,
                               int64_t workspace_bytes, void *stream);
/* The 3x3 / stride 1 / pad 1 convolutions of the 16-bit mode on a halo-staged kernel (csrc/h16_halo.hip; ResnetBlock
 * conv1 / conv2, Upsample.conv, conv_out: diffusionmodules/model.py:43-53, 88-102, 117-137): input patch staged once per
 * 32-channel slab, weights in MFMA-fragment order straight to registers, optional GroupNorm(+swish) of the INPUT applied
 * while staging (gn_mean_rstd [B][32][2] or NULL, gn_gamma / gn_beta [Cin]), the product computed transposed so that a lane
 * stores 16 consecutive channels of one pixel, optional statistics of the OUTPUT as per-chunk partial sums (gn_partial
 * [B][sgam_conv2d_h16_stats_chunks(d)][32][2] doubles or NULL).  Available when sgam_conv2d_h16_uses_halo(d) == 1 (3x3 s1 p1
 * on the input or its nearest-2x upsampling, Ho % 8 == 0, Wo % 8 == 0, Cin % 32 == 0, N % 128 == 0).  Maps too small to fill
 * the chip split the K slabs over grid.y (plan_ksplit, or chosen by the library): fp32 partial tiles go to `workspace`
 * (sgam_conv2d_halo_h16_workspace_bytes(d) bytes, 0 without split-K) and a combine launch adds them in a fixed order, applies
 * bias / residual, rounds and leaves the statistics.  w_frag from sgam_pack_conv_weight_h16_frag:
 * [Cout_pad / 32][KH*KW*Cin_pad / 32][128 pieces][8 halfs], piece = (k-step * 2 + k-half) * 32 + row', where row' = 8 (e / 4)
 * + 4 half + e % 4 for channel 16 half + e of the 32-channel tile (the accumulator slot order of the transposed product). */
int32_t sgam_conv2d_h16_uses_halo(const sgam_conv_desc *d);
int32_t sgam_conv2d_h16_stats_chunks(const sgam_conv_desc *d);
int64_t sgam_conv2d_halo_h16_workspace_bytes(const sgam_conv_desc *d);
int sgam_pack_conv_weight_h16_frag(const float *w_oihw, void *w_frag, int32_t ht, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW,
                                   int32_t Cout_pad, int32_t Cin_pad, void *stream);
int sgam_conv2d_halo_nhwc_h16(const sgam_conv_desc *d, int32_t ht, const void *x, const float *gn_mean_rstd,
                              const float *gn_gamma, const float *gn_beta, int32_t gn_swish, const void *w_frag,
                              const float *bias, const void *residual, void *out, int32_t out_f32, double *gn_partial,
                              void *workspace, int64_t workspace_bytes, void *stream);
/* (ABI v5) The same launch with the statistics of the INPUT still as its producer's chunk partials (gn_partial_in
 * [B][chunks_in][32][2] doubles — what sgam_conv2d_halo_nhwc_h16 leaves in `gn_partial` — instead of folded {mean, rstd}): the
 * kernel folds them itself while it stages (no fold launch between two convolutions of a ResnetBlock).  Offered where
 * sgam_conv2d_h16_gn_foldable(d, chunks_in) == 1: the 64-row tile walking <= 2 channel slabs per workgroup (the split-K plans
 * of the 16 x 16 / 32 x 32 maps), Cin % 256 == 0, chunks_in <= 16 — which is what the group-major split-K combine of those
 * maps produces (sgam_conv2d_h16_stats_chunks). */
int32_t sgam_conv2d_h16_gn_foldable(const sgam_conv_desc *d, int32_t chunks_in);
int sgam_conv2d_halo_gnp_nhwc_h16(const sgam_conv_desc *d, int32_t ht, const void *x, const double *gn_partial_in,
                                  int32_t chunks_in, float gn_eps, const float *gn_gamma, const float *gn_beta, int32_t gn_swish,
                                  const void *w_frag, const float *bias, const void *residual, void *out, int32_t out_f32,
                                  double *gn_partial, void *workspace, int64_t workspace_bytes, void *stream);
/* GroupNorm of 16-bit tensors from the partial statistics the halo kernel left / {mean, rstd} of any 16-bit tensor */
int sgam_groupnorm_from_partials_h16(const void *x, const double *partial, int32_t nchunk, const float *gamma, const float *beta,
                                     void *y, int32_t ht, int32_t B, int32_t HW, int32_t C, int32_t groups, float eps,
                                     int32_t fuse_swish, void *workspace, int64_t workspace_bytes, void *stream);
int sgam_groupnorm_meanrstd_nhwc_h16(const void *x, float *mean_rstd, int32_t ht, int32_t B, int32_t HW, int32_t C, int32_t groups,
                                     float eps, void *workspace, int64_t workspace_bytes, void *stream);
int sgam_pack_conv_weight_h16(const float *w_oihw, void *w_packed, int32_t ht, int32_t Cout, int32_t Cin,
                              int32_t KH, int32_t KW, int32_t Cout_pad, int32_t Cin_pad, void *stream);
int sgam_cast_f32_h16(const float *x, void *y, int32_t ht, int64_t n, void *stream);
int sgam_cast_h16_f32(const void *x, float *y, int32_t ht, int64_t n, void *stream);
int64_t sgam_groupnorm_h16_workspace_bytes(int32_t B, int32_t HW, int32_t C);
int sgam_groupnorm_nhwc_h16(const void *x, const float *gamma, const float *beta, void *y, int32_t ht, int32_t B,
                            int32_t HW, int32_t C, int32_t groups, float eps, int32_t fuse_swish, void *workspace,
                            int64_t workspace_bytes, void *stream);
/* p_out[r][:] (16-bit) = softmax(scale * s_in[r][:]) (fp32 scores) */
int sgam_softmax_rows_h16(const float *s_in, void *p_out, int32_t ht, int32_t rows, int32_t cols, int32_t lds,
                          int32_t ldp, float scale, void *stream);
/* (ABI v5) block-diagonal form: row r keeps the columns of its own block [r / block * block, + block), the others become
 * exact zeros — the attention of B images whose tokens do not fill the fused kernel (16 x 16 maps, C = 512) as ONE
 * (B n) x (B n) score matrix instead of B small GEMM / softmax / GEMM chains. */
int sgam_softmax_rows_blockdiag_h16(const float *s_in, void *p_out, int32_t ht, int32_t rows, int32_t cols, int32_t lds,
                                    int32_t ldp, float scale, int32_t block, void *stream);
int sgam_encode_head_h16(const float *x, const uint8_t *mask, const float *w, const float *bias, void *y,
                         int32_t ht, int32_t B, int32_t HW, int32_t ldy, void *stream);
/* y[c][p] = x[p][c] for a [HW][ldx] 16-bit matrix (v -> v^T for the P.V product) */
int sgam_transpose_h16(const void *x, void *y, int32_t ht, int32_t C, int32_t HW, int32_t ldx, void *stream);

/* ------------------------------------------------------------------------------------------
 * K4 — GroupNorm(32 groups, eps, affine) with optional fused swish, NHWC.
 * Replaces Normalize + nonlinearity: diffusionmodules/model.py:29-35, used at :119-127, :170,
 * :429-430, :536-537.  Biased variance, fp32 data, fp64 cross-thread accumulation.
 *   x,y [B][HW][C]; gamma,beta [C]; C % 128 == 0 (4 channels per lane, groups never straddle a lane)
 *   workspace: sgam_groupnorm_workspace_bytes(B, HW, C).
 * ------------------------------------------------------------------------------------------ */
int64_t sgam_groupnorm_workspace_bytes(int32_t B, int32_t HW, int32_t C);
int sgam_groupnorm_nhwc_f32(const float *x, const float *gamma, const float *beta, float *y, int32_t B,
                            int32_t HW, int32_t C, int32_t groups, float eps, int32_t fuse_swish,
                            void *workspace, int64_t workspace_bytes, void *stream);

/* Statistics half of K4 for the FUSED path: writes the per-(batch, channel) table
 *   scale_shift[b][c] = { rstd[b,g(c)] * gamma[c],  beta[c] - mean[b,g(c)] * rstd[b,g(c)] * gamma[c] }   ([B][C][2] floats)
 * which sgam_conv2d_gn_nhwc_f32 applies (with optional swish) to its A operand while staging it — the
 * normalised activation is never written to HBM.  Same constraints / workspace as sgam_groupnorm_nhwc_f32. */
int sgam_groupnorm_stats_nhwc_f32(const float *x, const float *gamma, const float *beta, float *scale_shift,
                                  int32_t B, int32_t HW, int32_t C, int32_t groups, float eps, void *workspace,
                                  int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * K5 — row softmax of the attention scores, in place: s[r][:] = softmax(scale * s[r][:]).
 * Replaces `w_ * c**-0.5` + softmax(dim=2): diffusionmodules/model.py:181-182.
 * ------------------------------------------------------------------------------------------ */
int sgam_softmax_rows_f32(float *s, int32_t rows, int32_t cols, int32_t ld, float scale, void *stream);
int sgam_softmax_rows_blockdiag_f32(float *s, int32_t rows, int32_t cols, int32_t ld, float scale, int32_t block,
                                    void *stream);   /* (ABI v5) see sgam_softmax_rows_blockdiag_h16 */

/* ------------------------------------------------------------------------------------------
 * K4-K6 fused — single-head self-attention of the AttnBlock in one pass over the keys (attention.hip):
 *   out[i][:] = sum_j softmax_j(scale * q_i . k_j) v_j,   q, k, v : [n][C] fp32 with row stride ld (the column
 *   slices of the fused q|k|v projection), out : [n][C] with row stride ldo.
 * Replaces torch.bmm(q, k) / `w_ * c**-0.5` / softmax(dim=2) / torch.bmm(v, w_) of AttnBlock.forward
 * (modules/diffusionmodules/model.py:176-187) without materialising the n x n scores.  Split-fp32 arithmetic as
 * sgam_conv2d_nhwc_f32x (three fp16 MFMAs per product, fp32 accumulation), fp32 online soft-max.
 * Supported: C == 256, n % 256 == 0, scale an exact power of two (C^-1/2 = 1/16); anything else -> SGAM_EINVAL and
 * the caller keeps the GEMM / softmax / GEMM chain.  workspace: sgam_attention_f32x_workspace_bytes(n, C) bytes,
 * 16-byte aligned (-1 = unsupported shape).
 * ------------------------------------------------------------------------------------------ */
int64_t sgam_attention_f32x_workspace_bytes(int32_t n, int32_t C);
int sgam_attention_f32x(const float *q, const float *k, const float *v, int32_t ld, int32_t n, int32_t C, float scale,
                        float *out, int32_t ldo, void *workspace, int64_t workspace_bytes, void *stream);

/* 16-bit throughput variant of the fused attention (`ht` = 0 bf16 / 1 fp16; q, k, v, out 16-bit with strides in elements;
 * one MFMA per product, fp32 scores / statistics / accumulation; same shape limits; scale need not be a power of two) */
int64_t sgam_attention_h16_workspace_bytes(int32_t n, int32_t C);
int sgam_attention_h16(const void *q, const void *k, const void *v, int32_t ht, int32_t ld, int32_t n, int32_t C, float scale,
                       void *out, int32_t ldo, void *workspace, int64_t workspace_bytes, void *stream);

/* Batched forms (ABI v4): B images of n tokens each, stacked along the rows — q, k, v, out are [B * n][...] and every query
 * attends to the keys of ITS image only (the AttnBlock at batch B, e.g. B lock-stepped scenes or B warp candidates: one
 * launch sequence instead of B).  The number of key ranges per image is chosen from (n, B) so that the launch fills the chip
 * with as few partial outputs as possible, at most 2048 keys per range (8 ranges for one 64 x 64 image, 2 from four images on); B = 1 is exactly the
 * unbatched entry point.  workspace: the matching *_batched_workspace_bytes(n, C, B). */
int64_t sgam_attention_f32x_batched_workspace_bytes(int32_t n, int32_t C, int32_t B);
int sgam_attention_f32x_batched(const float *q, const float *k, const float *v, int32_t ld, int32_t n, int32_t C, int32_t B,
                                float scale, float *out, int32_t ldo, void *workspace, int64_t workspace_bytes, void *stream);
/* (ABI v8) sgam_attention_f32x_batched followed by AttnBlock.proj_out (+ residual) — reference modules/diffusionmodules/model.py:176-191,
 * `h_ = bmm(v, w_); h_ = self.proj_out(h_); return x + h_` — with the merge of the key ranges fused into the projection: no [B n][C]
 * attention output is written and read back, one launch fewer.  w_planes / w_scale: proj_out's [C][C] weight as sgam_split_rows_f32x
 * returns it; bias [C] or NULL; residual [B n][ldr] or NULL; out [B n][ldc]; gn_partial (optional): [B][n / 32][32][2] fp64 {sum, sumsq}
 * of `out` per (32-row tile, group of C / 32 channels) for the GroupNorm that follows (fold with sgam_groupnorm_stats_from_partials_f32,
 * nchunk = n / 32).  Same shape limits and workspace as sgam_attention_f32x_batched. */
int sgam_attention_proj_f32x_batched(const float *q, const float *k, const float *v, int32_t ld, int32_t n, int32_t C, int32_t B,
                                     float scale, const void *w_planes, float w_scale, const float *bias, const float *residual,
                                     int32_t ldr, float *out, int32_t ldc, double *gn_partial, void *workspace,
                                     int64_t workspace_bytes, void *stream);
int64_t sgam_attention_h16_batched_workspace_bytes(int32_t n, int32_t C, int32_t B);
int sgam_attention_h16_batched(const void *q, const void *k, const void *v, int32_t ht, int32_t ld, int32_t n, int32_t C,
                               int32_t B, float scale, void *out, int32_t ldo, void *workspace, int64_t workspace_bytes,
                               void *stream);

/* (ABI v9) The AttnBlock of the 16-bit mode ahead of proj_out in ONE call of four launches — reference
 * modules/diffusionmodules/model.py:168-187: `h_ = self.norm(x); q = self.q(h_); k = self.k(h_); v = self.v(h_)`, then the attention
 * of sgam_attention_h16_batched.  GroupNorm is applied while the projection stages its operand (y = x scale + shift from the
 * per-channel table, fp32, one rounding to 16 bits — the arithmetic of sgam_groupnorm_from_partials_h16), the stacked q | k | v
 * projection writes q row-major and K / V^T directly in the fused attention's MFMA-fragment order: the stand-alone normalise pass, the
 * generic 1 x 1 GEMM and the fragment-split launch disappear.
 *   x          [B n][ldx] 16-bit block input (ldx % 8 == 0), C == 256, n % 256 == 0 (sgam_attention_h16_batched's shapes)
 *   gn_partial the chunk statistics x's producer left: [B][nchunk][32][2] fp64 {sum, sumsq}; gamma, beta [C] of AttnBlock.norm; eps
 *   w_frag     sgam_pack_qkv_weight_h16 of the stacked [3 C][C] fp32 weight (rows: q.weight, k.weight, v.weight): 3 C C 2 bytes
 *   bias       [3 C] fp32 (q.bias | k.bias | v.bias), 16-byte aligned
 *   out        [B n][ldo] 16-bit attention output (the operand of proj_out)
 *   workspace  sgam_attn_block_h16_workspace_bytes(n, C, B)
 * sgam_groupnorm_table_from_partials is the finalize half of sgam_groupnorm_from_partials_* on its own: the {scale, shift} table
 * [B][C][2] fp32 (scale = rstd gamma, shift = beta - mean scale) for a consumer that normalises while staging. */
/* (ABI v9) The whole AttnBlock of the split-fp32 path in three launches — reference modules/diffusionmodules/model.py:168-192,
 * `h_ = self.norm(x); q, k, v = self.q(h_), self.k(h_), self.v(h_); ...; return x + self.proj_out(h_)`: the fused front end (GroupNorm in
 * the operand staging of the stacked q | k | v projection, q row-major, K / V^T written straight in the attention's hi / lo fragment order —
 * the arithmetic of sgam_gemm_gn_f32x + the split launch it replaces, equal to fp32 round-off), the one-pass attention, and sgam_attention_proj_f32x_batched's merge +
 * proj_out + residual (= x).  mean_rstd [B][32][2] (sgam_groupnorm_stats_from_partials_f32); wqkv_planes / wqkv_scale: sgam_split_rows_f32x of
 * the stacked [3 C][C] weight with the rows of every 32-row tile permuted so that row 8 j + 4 h + i holds channel 16 h + 4 j + i; bqkv [3 C] in
 * natural order; wp_planes / wp_scale / bp: proj_out as for sgam_attention_proj_f32x_batched; gn_partial as there.  C == 256,
 * n % 256 == 0, scale a power of two; workspace: sgam_attn_block_f32x_workspace_bytes(n, C, B). */
int64_t sgam_attn_block_f32x_workspace_bytes(int32_t n, int32_t C, int32_t B);
int sgam_attn_block_f32x(const float *x, int32_t ldx, const float *mean_rstd, const float *gamma, const float *beta, const void *wqkv_planes,
                         float wqkv_scale, const float *bqkv, int32_t n, int32_t C, int32_t B, float scale, const void *wp_planes,
                         float wp_scale, const float *bp, float *out, int32_t ldc, double *gn_partial, void *workspace,
                         int64_t workspace_bytes, void *stream);
/* (ABI v9) The attention of the SMALL AttnBlocks in one launch — the 16 x 16 mid blocks (n = 256 tokens per image, C = 512; n = 128 too):
 * model.py:176-187, `w_ = bmm(q, k) * c**-0.5; w_ = softmax(w_, dim=2); h_ = bmm(v, w_)` — what otherwise runs as v^T transpose, operand
 * splits, the q k^T GEMM (+ split-K combine), sgam_softmax_rows_f32 and the P v GEMM: seven launches.  A workgroup holds a query tile's whole
 * score row (every query attends to the n keys of ITS image; B images stacked along the rows); the soft-max is sgam_softmax_rows_f32's
 * arithmetic; equal to the chain to fp32 round-off.  q, k, v: [B n][ld] fp32 (ld % 4 == 0), out [B n][ldo]; no workspace. */
int32_t sgam_attention_small_f32x_fits(int32_t n, int32_t C, int32_t B);
int sgam_attention_small_f32x(const float *q, const float *k, const float *v, int32_t ld, int32_t n, int32_t C, int32_t B, float scale,
                              float *out, int32_t ldo, void *stream);
/* 16-bit twin (`ht` = 0 bf16 / 1 fp16; q, k, v, out 16-bit, ld % 8 == 0): replaces sgam_transpose_h16 + the q k^T GEMM + sgam_softmax_rows_h16
 * + the P v GEMM; fp32 scores and soft-max (sgam_softmax_rows_h16's arithmetic), probabilities rounded once, fp32 accumulation. */
int sgam_attention_small_h16(const void *q, const void *k, const void *v, int32_t ht, int32_t ld, int32_t n, int32_t C, int32_t B, float scale,
                             void *out, int32_t ldo, void *stream);
int sgam_groupnorm_table_from_partials(const double *partial, int32_t nchunk, const float *gamma, const float *beta,
                                       float *scale_shift, int32_t B, int32_t HW, int32_t C, int32_t groups, float eps, void *stream);
int sgam_pack_qkv_weight_h16(const float *w, void *w_frag, int32_t ht, int32_t C, void *stream);
/* sgam_pack_weight_tp_h16: any [rows][C] 1 x 1 weight in that fragment order (rows % 32 == 0); sgam_attn_block_proj_h16: the WHOLE block —
 * sgam_attn_block_h16 with the merge of the key ranges fused into proj_out + residual (= x), model.py:187-191: out = x + proj_out(h_),
 * [B n][ldo] 16-bit, ldo % 8 == 0; wp_frag = sgam_pack_weight_tp_h16(proj_out.weight, rows = C); bp [C] or NULL; gn_partial_out (optional)
 * [B][n / 32][32][2] fp64 chunk statistics of the STORED output for the GroupNorm that follows. */
int sgam_pack_weight_tp_h16(const float *w, void *w_frag, int32_t ht, int32_t rows, int32_t C, void *stream);
int sgam_attn_block_proj_h16(const void *x, int32_t ldx, const double *gn_partial, int32_t nchunk, const float *gamma, const float *beta,
                             float eps, const void *w_frag, const float *bias, int32_t ht, int32_t n, int32_t C, int32_t B, float scale,
                             const void *wp_frag, const float *bp, double *gn_partial_out, void *out, int32_t ldo, void *workspace,
                             int64_t workspace_bytes, void *stream);
int64_t sgam_attn_block_h16_workspace_bytes(int32_t n, int32_t C, int32_t B);
int sgam_attn_block_h16(const void *x, int32_t ldx, const double *gn_partial, int32_t nchunk, const float *gamma, const float *beta,
                        float eps, const void *w_frag, const float *bias, int32_t ht, int32_t n, int32_t C, int32_t B, float scale,
                        void *out, int32_t ldo, void *workspace, int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * K7/K8 — nearest-codeword quantiser.  Replaces VectorQuantizer2.forward
 * (modules/vqvae/quantize.py:285-307): d = (|z|^2 + |e|^2) - 2 z.e (that expression order, fp32),
 * arg-min with first-index-of-ties, embedding gather, straight-through value z + (e - z).
 *   z        [T][D] tokens (NHWC latent), D % 32 == 0
 *   codebook [n_e][D], n_e % 64 == 0;  e_sq [n_e] = sgam_row_sumsq(codebook)
 *   dots     [T][n_e] scratch (z . e^T), caller-allocated
 *   idx_out  [T] int64;  zq_out [T][D] (may be NULL);  dist_out [T][n_e] optional (NULL to skip)
 *   workspace: sgam_vq_workspace_bytes(T, D, n_e) bytes (split-K partials for small T), may be 0
 * ------------------------------------------------------------------------------------------ */
int sgam_row_sumsq_f32(const float *x, float *out, int32_t rows, int32_t cols, void *stream);
int64_t sgam_vq_workspace_bytes(int32_t T, int32_t D, int32_t n_e);
int sgam_vq_nearest_f32(const float *z, const float *codebook, const float *e_sq, float *dots,
                        int64_t *idx_out, float *zq_out, float *dist_out, int32_t T, int32_t D,
                        int32_t n_e, int32_t straight_through, void *workspace, int64_t workspace_bytes,
                        void *stream);
/* commitment loss of VectorQuantizer2.forward (quantize.py:296-301, legacy = True):
 *   loss = mean((e[idx] - z)^2) + beta * mean((e[idx] - z)^2)   — the scalar VQModel.forward returns as `diff`
 * (model.py:144-147).  partial [T] doubles (scratch: per-token sums, folded in a fixed order); loss [1] float. */
int sgam_vq_commit_loss_f32(const float *z, const float *codebook, const int64_t *idx, double *partial, float *loss,
                            int32_t T, int32_t D, int32_t n_e, float beta, void *stream);
/* pure gather (quantize.py:368, get_codebook_entry :321-335): out[t][:] = codebook[idx[t]][:] */
int sgam_vq_gather_f32(const float *codebook, const int64_t *idx, float *out, int32_t T, int32_t D,
                       int32_t n_e, void *stream);
/* k smallest distances per token (ascending, ties -> lower index first), for the top-k infill
 * sampler (quantize.py:352-354).  vals [T][k], inds [T][k] int64; k <= 64. */
int sgam_vq_topk_f32(const float *dist, float *vals, int64_t *inds, int32_t T, int32_t n_e, int32_t k,
                     void *stream);

/* ------------------------------------------------------------------------------------------
 * Layout hops at the nn.Module boundary.
 * ------------------------------------------------------------------------------------------ */
int sgam_nchw_to_nhwc_f32(const float *x, float *y, int32_t B, int32_t C, int32_t HW, int32_t ldy,
                          void *stream); /* y[b][p][c] (row stride ldy >= C; columns >= C untouched) */
int sgam_nhwc_to_nchw_f32(const float *x, float *y, int32_t B, int32_t C, int32_t HW, int32_t ldx,
                          void *stream);
/* VQModel.encode head (model.py:107-113): cat(x, mask) -> 1x1 conv 5->4 -> NHWC with the pixel
 * stride padded to `ldy` floats (channels 4..ldy-1 zeroed) so the 3x3 conv_in runs on the MFMA path.
 *   x [B][4][HW] NCHW fp32, mask [B][HW] uint8 (0/1) or NULL (all zero), w [4][5], bias [4]. */
int sgam_encode_head_f32(const float *x, const uint8_t *mask, const float *w, const float *bias,
                         float *y, int32_t B, int32_t HW, int32_t ldy, void *stream);

/* ------------------------------------------------------------------------------------------
 * K9/K10/K12 — forward splat + 3x3 median hole fill + extrapolation mask (+ inverse-depth
 * normalisation).  Replaces render_projection_from_srcs_fast (sgam/point_rendering/warp.py:193-286,
 * pixel2cam :28-40, median_blur :306-347) and the depth normalisation of VQModel.get_x
 * (sgam/generative_sensing_module/model.py:210-229).
 *
 * Deterministic "largest linear point index wins" scatter (point p = pixel*N + src), the
 * semantics of the reference's sequential parallel=False loop (warp.py:246-249).
 *   src_feats   (B,N) images of 3 channels: element (b,n,c,pix) at
 *               src_feats[((b*N+n)*HW)*3... ] via strides: feat_cs (channel stride), feat_ps (pixel stride)
 *               - (B,N,3,H,W) tensors: feat_cs = HW, feat_ps = 1;  (B,N,H,W,3): feat_cs = 1, feat_ps = 3
 *   src_depths  [B][N][HW];  tgt_K [B][9];  src_Kinv [B*N][9] (inverse source intrinsics);
 *   T [B*N][16] source->target rigid transforms (row-major 4x4)
 *   winner      [B][HW] int32 scratch
 *   depth_range NULL at inference (mask = merge_depth <= 0, warp.py:285) or 2 floats (host pointer)
 *   dataset_norm 0: none, 1: google_earth, 2: clevr-infinite (model.py:210-229)
 * Outputs (any may be NULL): merge_depths [B][HW], merge_feats [B][3][HW], extrap [B][HW] uint8,
 *   x_out [B][4][HW] = cat(merge_feats, normalised inverse depth with holes = -2),
 *   proj_feats [B][3][HW], proj_depth [B][HW] (pre-fill planes), inb_mask [B][HW][N] uint8,
 *   pix_xy [B][HW][N][2] int32 (target pixel of every point, valid where inb_mask).
 * ------------------------------------------------------------------------------------------ */
int sgam_forward_splat_f32(const float *src_feats, int64_t feat_cs, int64_t feat_ps,
                           const float *src_depths, const float *tgt_K, const float *src_Kinv,
                           const float *T, int32_t B, int32_t N, int32_t H, int32_t W,
                           const float *depth_range, int32_t dataset_norm, int32_t *winner,
                           float *merge_depths, float *merge_feats, uint8_t *extrap, float *x_out,
                           float *proj_feats, float *proj_depth, uint8_t *inb_mask, int32_t *pix_xy,
                           void *stream);
/* Same splat with the sources addressed through a table of device pointers instead of one stacked tensor: the scene
 * loop (prepare_batch_data, inference_pipeline.py:534-537) keeps every generated frame as its own HBM allocation and
 * the warp reads them in place.  src_feat_ptrs / src_depth_ptrs: HOST arrays of B*N device pointers (entry b*N+n: the
 * [HW] x 3 features of that source, strides feat_cs / feat_ps, and its [HW] depth map); B*N <= 64 (the table travels by
 * value in the kernel arguments: no upload).  Everything else as sgam_forward_splat_f32. */
int sgam_forward_splat_srcs_f32(const float *const *src_feat_ptrs, const float *const *src_depth_ptrs, int64_t feat_cs,
                                int64_t feat_ps, const float *tgt_K, const float *src_Kinv, const float *T, int32_t B,
                                int32_t N, int32_t H, int32_t W, const float *depth_range, int32_t dataset_norm,
                                int32_t *winner, float *merge_depths, float *merge_feats, uint8_t *extrap, float *x_out,
                                float *proj_feats, float *proj_depth, uint8_t *inb_mask, int32_t *pix_xy, void *stream);

/* (ABI v6) The same forward splat WITHOUT global atomics — target-owned tiles with an LDS z-tile (csrc/warp.hip):
 * pass 1 caches every source point's target pixel and registers each 8 x 32 source bin with the target tiles its bounding
 * box meets, pass 2 gives a workgroup a 32 x 32 (16 x 16 for small launches) tile of the target image + 1-pixel halo, resolves
 * "largest point index wins" (warp.py:217-262) with LDS atomicMax over the registered bins, and finishes median fill / merge / mask /
 * depth normalisation (warp.py:264-286, model.py:210-229) from LDS.  Results are bit-identical to sgam_forward_splat_f32
 * for any geometry; the by-products inb_mask / pix_xy are only offered by the two-pass form above.
 *   workspace: sgam_forward_splat_workspace_bytes(B, N, H, W) bytes, 16-byte aligned: the per-tile bitmaps of registered
 *   source bins, then the cached target pixels [B][N][HW] int32.  Its first sgam_forward_splat_workspace_zero_bytes(B, N, H, W)
 *   bytes (the bitmaps) must be ZERO when the workspace is first used; every call leaves them zero again (no memset per
 *   call).  H, W <= 32767, N <= 64.  -1 from the queries: shape refused. */
int64_t sgam_forward_splat_workspace_bytes(int32_t B, int32_t N, int32_t H, int32_t W);
int64_t sgam_forward_splat_workspace_zero_bytes(int32_t B, int32_t N, int32_t H, int32_t W);
int sgam_forward_splat_tiled_f32(const float *src_feats, int64_t feat_cs, int64_t feat_ps, const float *src_depths,
                                 const float *tgt_K, const float *src_Kinv, const float *T, int32_t B, int32_t N, int32_t H,
                                 int32_t W, const float *depth_range, int32_t dataset_norm, void *workspace,
                                 int64_t workspace_bytes, float *merge_depths, float *merge_feats, uint8_t *extrap,
                                 float *x_out, float *proj_feats, float *proj_depth, void *stream);
int sgam_forward_splat_tiled_srcs_f32(const float *const *src_feat_ptrs, const float *const *src_depth_ptrs, int64_t feat_cs,
                                      int64_t feat_ps, const float *tgt_K, const float *src_Kinv, const float *T, int32_t B,
                                      int32_t N, int32_t H, int32_t W, const float *depth_range, int32_t dataset_norm,
                                      void *workspace, int64_t workspace_bytes, float *merge_depths, float *merge_feats,
                                      uint8_t *extrap, float *x_out, float *proj_feats, float *proj_depth, void *stream);

/* K12 standalone (VQModel.get_x, model.py:196-199 + 210-229, when the warped view is supplied by the
 * caller): compute_mask=1: extrap = depth <= 0, out = normalised inverse depth with holes = -2;
 * compute_mask=0: out = 2*norm(depth)-1 only (the ground-truth branch x_scaled_inverse_depth). */
int sgam_depth_normalise_f32(const float *depth, int32_t compute_mask, uint8_t *extrap, float *out,
                             int32_t dataset_norm, int64_t n, void *stream);

/* ------------------------------------------------------------------------------------------
 * K11 — target-depth-driven inverse warp with best-source selection.  Replaces
 * InfiniteSceneGeneration.inverse_warping (sgam/inference_pipeline.py:662-743).
 *   src_imgs [B][N][3][HW], src_depths [B][N][HW], tgt_depth [B][HW], src_K [B*N][9],
 *   tgt_Kinv [B][9], T_tgt2src [B*N][16];  warped [B][3][HW];  zbuf [B][HW] optional.
 * ------------------------------------------------------------------------------------------ */
int sgam_inverse_warp_f32(const float *src_imgs, const float *src_depths, const float *tgt_depth,
                          const float *src_K, const float *tgt_Kinv, const float *T_tgt2src, int32_t B,
                          int32_t N, int32_t H, int32_t W, float *warped, float *zbuf, void *stream);
/* Same with a pointer table (HOST arrays of B*N <= 64 device pointers) and free channel / pixel strides of the source
 * images: (3,H,W) planes: img_cs = HW, img_ps = 1; the frame store's (H,W,3): img_cs = 1, img_ps = 3. */
int sgam_inverse_warp_srcs_f32(const float *const *src_img_ptrs, const float *const *src_depth_ptrs, int64_t img_cs,
                               int64_t img_ps, const float *tgt_depth, const float *src_K, const float *tgt_Kinv,
                               const float *T_tgt2src, int32_t B, int32_t N, int32_t H, int32_t W, float *warped,
                               float *zbuf, void *stream);

/* ------------------------------------------------------------------------------------------
 * Frame feedback codec (inference_pipeline.py:898-911 then :534-537): decoder output (B,4,HW)
 * in [-1,1] -> RGB quantised to uint8 by truncation and re-expanded with the host-built 256-entry
 * table lut[u] = float32(u / 127.5 - 1.0), metric depth from the normalised inverse depth.
 *   rgb_u8 [B][HW][3] (HWC, optional), rgb_f [B][HW][3] (HWC fp32, what prepare_batch_data reloads),
 *   depth [B][HW].
 * ------------------------------------------------------------------------------------------ */
int sgam_frame_feedback_f32(const float *dec, const float *lut256, int32_t dataset_norm, uint8_t *rgb_u8,
                            float *rgb_f, float *depth, int32_t B, int32_t HW, void *stream);
/* the re-read half alone, for frames that arrive as uint8 (the seed frame): rgb_f[i] = lut256[rgb_u8[i]], n values */
int sgam_rgb_u8_to_f32(const uint8_t *rgb_u8, const float *lut256, float *rgb_f, int64_t n, void *stream);

/* Fused q | k | v projection of AttnBlock on the split-fp32 path (diffusionmodules/model.py:168-175: norm -> three 1x1
 * convolutions): out[M][N] = GroupNorm(x)[M][K] . W[N][K]^T + bias with the normalisation applied while the 64 x 256 operand
 * panel is staged (mean_rstd [B][32][2] from sgam_groupnorm_stats_*; no swish), w_planes / w_scale = a SplitWeight of the
 * stacked weights (sgam_split_rows_f32x).  sgam_gemm_gn_f32x_fits: M % 64 == 0 inside whole images of HW rows, N % 128 == 0,
 * K % 256 == 0 (K = 128 without normalisation). */
int32_t sgam_gemm_gn_f32x_fits(int32_t M, int32_t N, int32_t K, int32_t HW);
/* the same kernel for the other 1x1 convolutions / GEMMs that fit (proj_out, nin_shortcut, quant_conv): mean_rstd / gamma /
 * beta NULL = no normalisation (then K = 128 is accepted too); optional residual [M][ldr]; optional gn_partial
 * [B][HW / 64][32][2] fp64 = per-chunk statistics of `out` for the next GroupNorm (N / 32 a power of two <= 32) */
int sgam_gemm_panel_f32x(const float *x, int32_t lda, const float *mean_rstd, const float *gamma, const float *beta,
                         const void *w_planes, float w_scale, const float *bias, const float *residual, int32_t ldr, float *out,
                         int32_t ldc, double *gn_partial, int32_t M, int32_t N, int32_t K, int32_t HW, void *stream);
int sgam_gemm_gn_f32x(const float *x, int32_t lda, const float *mean_rstd, const float *gamma, const float *beta, const void *w_planes,
                      float w_scale, const float *bias, float *out, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t HW,
                      void *stream);

/* ------------------------------------------------------------------------------------------
 * f1 — TSDF fusion of the generated RGB-D frames + depth render at the target pose.  Replaces
 * InfiniteSceneGeneration.rgbd_integration (sgam/inference_pipeline.py:119-133, 745-838: Open3D 0.15.2
 * ScalableTSDFVolume.integrate / extract_triangle_mesh / OffscreenRenderer.render_to_depth_image).
 * Integration is Open3D's published rule (16^3-voxel units, stride-4 unit opening, running weighted mean of
 * min(1, sdf / sdf_trunc)); the depth render is a direct ray cast of the fused surface (first +/- zero crossing of the
 * trilinear TSDF, view-space z, 0 = nothing hit).  Colour (TSDFVolumeColorType::RGB8, :123-131, 777-790) is fused
 * with the same running mean — color <- (color * w + rgb(u, v)) / (w + 1) per channel, 0..255 — when rgb_u8 [H][W][3] and
 * brick_color [max_bricks][16*16*16][3] fp32 (0-initialised) are given (both or neither), and the ray cast can return the
 * colour of the nearest voxel at the hit (color_out [H][W][3] fp32, 0 = nothing hit); the conditioning path itself only
 * consumes the depth.
 *   unit_table [dims.z][dims.y][dims.x] int32, -1 = closed, else brick index | 0x40000000 once the brick holds part of
 *   the truncation band (the ray cast only marches those); unit_stamp same shape, 0-initialised;
 *   counters int32[4 * 32], zero-initialised: counter k at index 32 k (one 128-byte line each: same-line atomics are serialised)
 *   = {bricks allocated, length of this step's brick list, samples outside the box, pool overflows};
 *   brick_tsdf [max_bricks][16*16*16] fp32 initialised to 2.0 (= unobserved: observed values are <= 1, so the ray cast
 *   needs no weight loads), brick_weight same shape, 0-initialised; brick_list int32[max_list] scratch.
 *   cam2world / world2cam: row-major 4x4 HOST values (copied into the kernel arguments: no upload, no device allocation
 *   per frame); intrinsics by value.
 * All state is caller-owned; nothing is synchronised or read back.
 * ------------------------------------------------------------------------------------------ */
typedef struct sgam_tsdf_grid {
    float voxel_length, sdf_trunc;
    int32_t unit_base[3];    /* unit index (floor(world / (16 * voxel_length))) of the box's low corner, x y z */
    int32_t unit_dims[3];    /* units per axis */
} sgam_tsdf_grid;
/* (ABI v10) One source frame of a step: depth [H][W] fp32 and (with brick_color) rgb_u8 [H][W][3] DEVICE pointers; cam2world /
 * world2cam row-major 4x4 by value.  The array of n_src <= 8 of them is a HOST array, copied into the kernel arguments. */
typedef struct sgam_tsdf_src {
    const float *depth;
    const uint8_t *rgb_u8;
    float cam2world[16], world2cam[16];
} sgam_tsdf_src;
/* Integrates the n_src source frames of ONE step of the scene loop (reference :757-790 calls volume.integrate once per source)
 * in ONE pass over the union of the units they open: a voxel is loaded once, takes the sources' updates in array order and is
 * stored once — the same values as n_src single-source calls in that order.  step_id in 1 .. 2^23 - 1, distinct per call
 * (unit_stamp holds (step_id << 8) | mask of the step's sources that opened the unit). */
int sgam_tsdf_integrate_srcs_f32(const sgam_tsdf_grid *grid, const sgam_tsdf_src *srcs, int32_t n_src, int32_t H, int32_t W, float fx,
                                 float fy, float cx, float cy, float depth_trunc, int32_t step_id, int32_t *unit_table,
                                 int32_t *unit_stamp, int32_t *counters, int32_t *brick_list, int32_t max_list, float *brick_tsdf,
                                 float *brick_weight, int32_t max_bricks, float *brick_color, const float *ray_mult, void *stream);
/* ray_mult [H][W] fp32 = sqrt(((u - cx) / fx)^2 + ((v - cy) / fy)^2 + 1), the rule's depth -> camera-distance multiplier of a
 * pixel: tabulated once per (intrinsics, size) by this call, gathered by the integration beside the depth. */
int sgam_tsdf_ray_mult_f32(int32_t H, int32_t W, float fx, float fy, float cx, float cy, float *ray_mult, void *stream);
int sgam_tsdf_raycast_depth_f32(const sgam_tsdf_grid *grid, int32_t H, int32_t W, float fx, float fy, float cx, float cy,
                                const float *cam2world, float z_near, float z_far, const int32_t *unit_table,
                                const float *brick_tsdf, float *depth_out, const float *brick_color, float *color_out,
                                void *stream);

/* (ABI v6) Zero-crossing point extraction from the fused bricks: `volume.extract_point_cloud()` of the reference's run tail
 * (sgam/inference_pipeline.py:446-450 -> rgbd_integrated_mesh.ply), Open3D's published ScalableTSDFVolume::ExtractPointCloud
 * rule: per observed voxel with |tsdf| < 0.98 and each +x / +y / +z neighbour (same test) with the opposite sign, one point on
 * the edge at the |tsdf|-weighted position; colour weighted the same way (0..255, needs brick_color); normal = normalised
 * central difference of the trilinear field, one voxel each way.
 *   counter: device uint64, zero on entry; holds the number of points FOUND on return (a first call with points = NULL only
 *   counts).  points / normals / colors [max_points][3] fp32, keys [max_points] int64 = ((unit slot * 4096 + voxel) * 3 + axis)
 *   — the points arrive in no particular order, sorting by key gives a run-independent one.  Parity unpinned vs Open3D. */
int sgam_tsdf_extract_points_f32(const sgam_tsdf_grid *grid, const int32_t *unit_table, const float *brick_tsdf,
                                 const float *brick_color, uint64_t *counter, int64_t max_points, float *points,
                                 float *normals, float *colors, int64_t *keys, void *stream);

/* ------------------------------------------------------------------------------------------
 * f4 — backward / optimiser kernels of the training step: VQModel.training_step
 * (sgam/generative_sensing_module/model.py:271-345) with VQLPIPSWithDiscriminator.forward(optimizer_idx = 0)
 * (modules/losses/vqperceptual.py:77-110) at perceptual_weight = 0 and global_step < disc_start, i.e.
 * loss = mean|x - xrec| + codebook_weight * qloss, torch.optim.Adam(betas = (0.5, 0.9)) (model.py:414-428).
 * Every product of the backward pass runs on the forward's MFMA GEMM (sgam_conv2d_gn_nhwc_f32, fp32-in mode):
 *   data gradient    dcol[M][KH*KW*cin_pad] = dy[M][Cout] . W[Cout][KH*KW*cin_pad],   dx = sgam_col2im_gather_f32(dcol)
 *   weight gradient  dW[Cout][KH*KW*cin_pad] = dy^T[Cout][M] . col^T[KH*KW*cin_pad][M]^T,   col^T = sgam_im2col_t_f32(x)
 *                    (rows ld_m >= M apart: M is rounded up to the GEMM's K granule with a zero tail)
 * for every convolution of the model through ONE pair of index kernels driven by the forward's descriptor (3x3 / 1x1,
 * stride 1, Downsample's stride 2 with (0,1,0,1) padding, Upsample's nearest-2x folded into the conv); k = tap * cin_pad
 * + channel as in sgam_pack_conv_weight.  The PatchGAN and LPIPS pieces follow below.
 * ------------------------------------------------------------------------------------------ */
int sgam_im2col_t_f32(const sgam_conv_desc *d, const float *x, float *col_t, int32_t cin_pad, int64_t ld_m, void *stream);
int sgam_col2im_gather_f32(const sgam_conv_desc *d, const float *dcol, float *dx, int32_t cin_pad, void *stream);
int sgam_unpack_conv_weight_grad_f32(const float *grad_packed, int32_t ld, float *grad_oihw, int32_t Cout, int32_t Cin,
                                     int32_t KH, int32_t KW, int32_t Cin_pad, void *stream);
/* bias gradient: out[N] = column sums of a[M][N] (two launches, fixed order) */
int64_t sgam_colsum_workspace_bytes(int32_t M, int32_t N);
int sgam_colsum_f32(const float *a, int32_t lda, float *out, int32_t M, int32_t N, void *workspace, int64_t workspace_bytes,
                    void *stream);
/* GroupNorm(32, eps)(+swish) backward (Normalize + nonlinearity, diffusionmodules/model.py:30-40): x = the layer's input,
 * dy = gradient of its output, mean_rstd [B][groups][2] from sgam_groupnorm_stats_*; dx like x; dgamma_b / dbeta_b [B][C]
 * per-image sums (the caller adds the images); group_means [B][groups][2] scratch */
int64_t sgam_groupnorm_bwd_workspace_bytes(int32_t B, int32_t HW, int32_t C);
int sgam_groupnorm_bwd_nhwc_f32(const float *x, const float *dy, const float *mean_rstd, const float *gamma, const float *beta,
                                int32_t swish, float *dx, float *dgamma_b, float *dbeta_b, float *group_means, int32_t B,
                                int32_t HW, int32_t C, int32_t groups, void *workspace, int64_t workspace_bytes, void *stream);
/* p = softmax(scale * s) over rows (AttnBlock, model.py:176-181): ds = scale * p * (dp - sum_j dp_j p_j) */
int sgam_softmax_bwd_rows_f32(const float *p, const float *dp, float *ds, int32_t rows, int32_t cols, int32_t ld, float scale,
                              void *stream);
/* rec_loss = |inputs - reconstructions| (vqperceptual.py:79): grad[rows][ld_grad] = sign(rec - target) * grad_scale on the C
 * real columns (0 on the padding), partial[ceil(rows * ld_grad / 256)] doubles = sums of |rec - target| */
int sgam_l1_loss_grad_f32(const float *rec, const float *target, float *grad, double *partial, int64_t rows, int32_t C,
                          int32_t ld_rec, int32_t ld_grad, float grad_scale, void *stream);
/* VectorQuantizer2 backward (quantize.py:296-304, legacy form): dz = dzq + two_c * (z - zq) with two_c = 2 * codebook_weight /
 * numel; d_codebook[k] = two_c_beta * sum_{t: indices[t] = k} (zq_t - z_t) */
int sgam_vq_bwd_f32(const float *dzq, const float *z, const float *zq, float *dz, int64_t n, float two_c, void *stream);
int sgam_vq_codebook_grad_f32(const int64_t *indices, const float *z, const float *zq, float *d_codebook, int32_t T, int32_t n_e,
                              int32_t D, float two_c_beta, void *stream);
/* out = alpha * a + beta * b (b may be NULL): gradient fan-in of the residual connections */
int sgam_axpby_f32(const float *a, const float *b, float *out, int64_t n, float alpha, float beta, void *stream);
/* torch.optim.Adam step (no weight decay / amsgrad) on one tensor; step = 1, 2, ... */
int sgam_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                       float beta2, float eps, int32_t step, void *stream);
/* The same step for a whole parameter set in ONE launch (`opt.step()` over the 159 / 345 tensors of a phase).  All seven arrays are
 * DEVICE arrays: per tensor t the four pointers and numels[t]; per workgroup b the tensor block_tensor[b] it works on and the
 * first element block_off[b] of its 4096-element chunk (the host lists ceil(numel / 4096) workgroups per tensor, n_blocks in all). */
int sgam_adam_multi_step_f32(float *const *params, const float *const *grads, float *const *exp_avgs, float *const *exp_avg_sqs,
                             const int64_t *numels, const int32_t *block_tensor, const int64_t *block_off, int32_t n_blocks, float lr,
                             float beta1, float beta2, float eps, int32_t step, void *stream);

/* PatchGAN discriminator pieces (modules/discriminator/model.py:17-67; hinge loss and adaptive weight,
 * modules/losses/vqperceptual.py:17-21, 63-75): nn.BatchNorm2d in training mode over [rows = B*H*W][C] NHWC matrices
 * (mean_rstd [C][2] from the batch, biased variance; running_mean / running_var updated with the unbiased one), fused with
 * LeakyReLU; mean_rstd = NULL: LeakyReLU alone (the first layer).  Workspace: sgam_batchnorm_workspace_bytes. */
int64_t sgam_batchnorm_workspace_bytes(int32_t rows, int32_t C);
int sgam_batchnorm_stats_f32(const float *x, float *mean_rstd, float *running_mean, float *running_var, int32_t rows, int32_t C,
                             float eps, float momentum, void *workspace, int64_t workspace_bytes, void *stream);
int sgam_bn_lrelu_fwd_f32(const float *x, const float *mean_rstd, const float *gamma, const float *beta, float *y, int32_t rows,
                          int32_t C, float slope, void *stream);
int sgam_bn_lrelu_bwd_f32(const float *x, const float *dy, const float *mean_rstd, const float *gamma, const float *beta, float *dx,
                          float *dgamma, float *dbeta, float *gbuf, float *means, int32_t rows, int32_t C, float slope,
                          void *workspace, int64_t workspace_bytes, void *stream);
/* mode +1: relu(1 + l), -1: relu(1 - l) (hinge_d_loss); +2: softplus(l), -2: softplus(-l) (vanilla_d_loss, vqperceptual.py:17-28); 0: l;
 * grad = d(term)/dl * grad_scale, partial[ceil(n / 256)] = sums of the terms */
int sgam_hinge_terms_f32(const float *logits, float *grad, double *partial, int64_t n, int32_t mode, float grad_scale, void *stream);
int sgam_sumsq_partial_f32(const float *a, double *partial, int64_t n, void *stream);

/* LPIPS pieces (modules/losses/lpips.py:10-123): MaxPool2d(2, 2) of the VGG16 trunk and its backward (gradient to the first
 * maximum of the window), ScalingLayer ((x - shift) / scale on the RGB channels into a zero-padded NHWC tensor; with shift 0 the
 * same kernel is its backward), and one feature level of the metric: val_b = mean_p sum_c w_c (f0/(|f0|+eps) - f1/(|f1|+eps))^2
 * (normalize_tensor -> squared difference -> NetLinLayer -> spatial_average) with its gradient w.r.t. f0. */
int sgam_maxpool2x2_f32(const float *x, float *y, int32_t B, int32_t H, int32_t W, int32_t C, void *stream);
int sgam_maxpool2x2_bwd_f32(const float *x, const float *dy, float *dx, int32_t B, int32_t H, int32_t W, int32_t C, void *stream);
int sgam_channel_affine_f32(const float *x, int32_t ldx, float *y, int32_t ldy, int64_t rows, int32_t n, const float *shift4,
                            const float *inv_scale4, void *stream);
int sgam_lpips_level_f32(const float *f0, const float *f1, const float *lin_w, double *partial, float *df0, int32_t B, int32_t HW,
                         int32_t C, float eps, float grad_scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SGAM_HIP_H */
