"""ctypes binding of libsgam_hip.so (include/sgam_hip.h).  There is NO fallback: if the library is
missing or a call fails, the product path raises — results never come from a CPU/torch substitute."""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# SGAM_HIP_LIB: an alternative build of the same library (kernel A/B experiments, scripts/exp_*.sh); default = the in-tree build
LIB_PATH = os.environ.get("SGAM_HIP_LIB") or os.path.join(_PKG, "lib", "libsgam_hip.so")

c_i32, c_i64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    """mirror of struct sgam_conv_desc"""
    _fields_ = [(n, c_i32) for n in (
        "B", "Hi", "Wi", "Cin", "Ho", "Wo", "N", "KH", "KW", "stride", "pad_t", "pad_l", "upsample2x",
        "lda", "ldb", "ldc", "ldr", "n_valid", "bias_per_row", "plan_bm", "plan_bn", "plan_ksplit")]


class TsdfGrid(ctypes.Structure):
    """mirror of struct sgam_tsdf_grid"""
    _fields_ = [("voxel_length", c_f32), ("sdf_trunc", c_f32), ("unit_base", c_i32 * 3), ("unit_dims", c_i32 * 3)]


class TsdfSrc(ctypes.Structure):
    """mirror of struct sgam_tsdf_src"""
    _fields_ = [("depth", c_vp), ("rgb_u8", c_vp), ("cam2world", c_f32 * 16), ("world2cam", c_f32 * 16)]


# name -> (restype, argtypes); must list every symbol include/sgam_hip.h declares
ABI_VERSION = 10     # include/sgam_hip.h: sgam_abi_version() of the library these prototypes were written against

PROTOTYPES = {
    "sgam_abi_version": (c_i32, []),
    "sgam_build_info": (ctypes.c_char_p, []),
    "sgam_build_commit": (ctypes.c_char_p, []),
    "sgam_build_digest": (ctypes.c_char_p, []),
    "sgam_prof_enable": (c_i32, [c_i32]),
    "sgam_prof_mark_empty": (c_i32, [c_vp]),
    "sgam_prof_count": (c_i32, []),
    "sgam_prof_get": (c_i32, [c_i32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p),
                              ctypes.POINTER(c_f32), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "sgam_prof_get_shape": (c_i32, [c_i32, ctypes.POINTER(c_i32)]),
    "sgam_conv2d_workspace_bytes": (c_i64, [ctypes.POINTER(ConvDesc)]),
    "sgam_conv2d_plan": (c_i32, [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                                 ctypes.POINTER(c_i32)]),
    "sgam_conv2d_nhwc_f32": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "sgam_conv2d_gn_nhwc_f32": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                                        c_vp]),
    "sgam_pack_conv_weight": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sgam_f32x_set_range_flag": (c_i32, [c_vp]),
    "sgam_conv2d_f32x_workspace_bytes": (c_i64, [ctypes.POINTER(ConvDesc)]),
    "sgam_conv2d_f32x_plan": (c_i32, [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                                      ctypes.POINTER(c_i32)]),
    "sgam_conv2d_nhwc_f32x": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_f32, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_i64,
                                      c_vp]),
    "sgam_conv2d_f32x_stats_chunks": (c_i32, [ctypes.POINTER(ConvDesc)]),
    "sgam_conv2d_f32x_stats_mode": (c_i32, [ctypes.POINTER(ConvDesc)]),
    "sgam_conv2d_stats_nhwc_f32x": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_f32, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                            c_i64, c_vp]),
    "sgam_groupnorm_from_partials_f32": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32,
                                                 c_i32, c_vp, c_i64, c_vp]),
    "sgam_conv2d_f32x_gn_fusable": (c_i32, [ctypes.POINTER(ConvDesc)]),
    "sgam_conv2d_f32x_uses_halo": (c_i32, [ctypes.POINTER(ConvDesc)]),
    "sgam_conv2d_gn_nhwc_f32x": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_f32, c_vp, c_vp,
                                         c_vp, c_vp, c_vp, c_i64, c_vp]),
    "sgam_conv2d_f32x_gn_foldable": (c_i32, [ctypes.POINTER(ConvDesc), c_i32]),
    "sgam_conv2d_gnp_nhwc_f32x": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_i32, c_f32, c_vp, c_vp, c_i32, c_vp, c_f32, c_vp,
                                          c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "sgam_groupnorm_stats_from_partials_f32": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "sgam_groupnorm_meanrstd_nhwc_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_i64, c_vp]),
    "sgam_tsdf_integrate_srcs_f32": (c_i32, [ctypes.POINTER(TsdfGrid), ctypes.POINTER(TsdfSrc), c_i32, c_i32, c_i32, c_f32, c_f32, c_f32,
                                             c_f32, c_f32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "sgam_tsdf_ray_mult_f32": (c_i32, [c_i32, c_i32, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp]),
    "sgam_tsdf_raycast_depth_f32": (c_i32, [ctypes.POINTER(TsdfGrid), c_i32, c_i32, c_f32, c_f32, c_f32, c_f32, c_vp, c_f32,
                                            c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "sgam_tsdf_extract_points_f32": (c_i32, [ctypes.POINTER(TsdfGrid), c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "sgam_gemm_gn_f32x_fits": (c_i32, [c_i32, c_i32, c_i32, c_i32]),
    "sgam_gemm_panel_f32x": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32,
                                     c_i32, c_vp]),
    "sgam_gemm_gn_f32x": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sgam_im2col_t_f32": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_i32, c_i64, c_vp]),
    "sgam_col2im_gather_f32": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_i32, c_vp]),
    "sgam_unpack_conv_weight_grad_f32": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sgam_colsum_workspace_bytes": (c_i64, [c_i32, c_i32]),
    "sgam_colsum_f32": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "sgam_groupnorm_bwd_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "sgam_groupnorm_bwd_nhwc_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                                            c_vp, c_i64, c_vp]),
    "sgam_softmax_bwd_rows_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "sgam_l1_loss_grad_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "sgam_vq_bwd_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_vp]),
    "sgam_vq_codebook_grad_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "sgam_axpby_f32": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_vp]),
    "sgam_adam_step_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_i32, c_vp]),
    "sgam_adam_multi_step_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_f32, c_f32, c_f32, c_i32, c_vp]),
    "sgam_batchnorm_workspace_bytes": (c_i64, [c_i32, c_i32]),
    "sgam_batchnorm_stats_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_vp, c_i64, c_vp]),
    "sgam_bn_lrelu_fwd_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_vp]),
    "sgam_bn_lrelu_bwd_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_vp, c_i64,
                                      c_vp]),
    "sgam_hinge_terms_f32": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_vp]),
    "sgam_sumsq_partial_f32": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "sgam_maxpool2x2_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sgam_maxpool2x2_bwd_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sgam_channel_affine_f32": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "sgam_lpips_level_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_vp]),
    "sgam_pack_conv_weight_f32x": (c_i32, [c_vp, c_vp, c_f32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sgam_split_rows_f32x": (c_i32, [c_vp, c_vp, c_f32, c_i32, c_i32, c_i32, c_vp]),
    "sgam_conv2d_h16_workspace_bytes": (c_i64, [ctypes.POINTER(ConvDesc)]),
    "sgam_conv2d_h16_plan": (c_i32, [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                                     ctypes.POINTER(c_i32)]),
    "sgam_conv2d_nhwc_h16": (c_i32, [ctypes.POINTER(ConvDesc), c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i64,
                                     c_vp]),
    "sgam_conv2d_h16_generic_stats_chunks": (c_i32, [ctypes.POINTER(ConvDesc)]),
    "sgam_conv2d_stats_nhwc_h16": (c_i32, [ctypes.POINTER(ConvDesc), c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i64,
                                           c_vp]),
    "sgam_conv2d_h16_uses_halo": (c_i32, [ctypes.POINTER(ConvDesc)]),
    "sgam_conv2d_h16_stats_chunks": (c_i32, [ctypes.POINTER(ConvDesc)]),
    "sgam_pack_conv_weight_h16_frag": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sgam_conv2d_halo_h16_workspace_bytes": (c_i64, [ctypes.POINTER(ConvDesc)]),
    "sgam_conv2d_halo_nhwc_h16": (c_i32, [ctypes.POINTER(ConvDesc), c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp,
                                          c_i32, c_vp, c_vp, c_i64, c_vp]),
    "sgam_conv2d_h16_gn_foldable": (c_i32, [ctypes.POINTER(ConvDesc), c_i32]),
    "sgam_conv2d_halo_gnp_nhwc_h16": (c_i32, [ctypes.POINTER(ConvDesc), c_i32, c_vp, c_vp, c_i32, c_f32, c_vp, c_vp, c_i32, c_vp, c_vp,
                                              c_vp, c_vp, c_i32, c_vp, c_vp, c_i64, c_vp]),
    "sgam_groupnorm_from_partials_h16": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32,
                                                 c_i32, c_vp, c_i64, c_vp]),
    "sgam_groupnorm_meanrstd_nhwc_h16": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_i64, c_vp]),
    "sgam_pack_conv_weight_h16": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sgam_cast_f32_h16": (c_i32, [c_vp, c_vp, c_i32, c_i64, c_vp]),
    "sgam_cast_h16_f32": (c_i32, [c_vp, c_vp, c_i32, c_i64, c_vp]),
    "sgam_groupnorm_h16_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "sgam_groupnorm_nhwc_h16": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp,
                                        c_i64, c_vp]),
    "sgam_softmax_rows_h16": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "sgam_softmax_rows_blockdiag_h16": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "sgam_encode_head_h16": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sgam_transpose_h16": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sgam_groupnorm_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "sgam_groupnorm_nhwc_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp,
                                        c_i64, c_vp]),
    "sgam_groupnorm_stats_nhwc_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_i64,
                                              c_vp]),
    "sgam_softmax_rows_f32": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "sgam_softmax_rows_blockdiag_f32": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "sgam_attention_f32x_workspace_bytes": (c_i64, [c_i32, c_i32]),
    "sgam_attention_f32x": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, c_i32, c_vp, c_i64, c_vp]),
    "sgam_attention_h16_workspace_bytes": (c_i64, [c_i32, c_i32]),
    "sgam_attention_h16": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_i32, c_vp, c_i64, c_vp]),
    "sgam_attention_f32x_batched_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "sgam_attention_f32x_batched": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_i32, c_vp, c_i64, c_vp]),
    "sgam_attention_proj_f32x_batched": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_f32, c_vp, c_vp, c_i32, c_vp,
                                                 c_i32, c_vp, c_vp, c_i64, c_vp]),
    "sgam_attention_h16_batched_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "sgam_attention_h16_batched": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_i32, c_vp, c_i64, c_vp]),
    "sgam_attn_block_f32x_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "sgam_attn_block_f32x": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, c_f32, c_vp, c_vp, c_i32,
                                     c_vp, c_vp, c_i64, c_vp]),
    "sgam_attention_small_f32x_fits": (c_i32, [c_i32, c_i32, c_i32]),
    "sgam_attention_small_f32x": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_i32, c_vp]),
    "sgam_attention_small_h16": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_i32, c_vp]),
    "sgam_groupnorm_table_from_partials": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "sgam_pack_qkv_weight_h16": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "sgam_pack_weight_tp_h16": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "sgam_attn_block_proj_h16": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp,
                                         c_vp, c_vp, c_i32, c_vp, c_i64, c_vp]),
    "sgam_attn_block_h16_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "sgam_attn_block_h16": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_i32,
                                    c_vp, c_i64, c_vp]),
    "sgam_row_sumsq_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "sgam_vq_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "sgam_vq_nearest_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp,
                                    c_i64, c_vp]),
    "sgam_vq_commit_loss_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "sgam_vq_gather_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "sgam_vq_topk_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "sgam_nchw_to_nhwc_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sgam_nhwc_to_nchw_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sgam_encode_head_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "sgam_forward_splat_f32": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                                       ctypes.POINTER(c_f32), c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                       c_vp, c_vp, c_vp]),
    "sgam_forward_splat_srcs_f32": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                                            ctypes.POINTER(c_f32), c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                            c_vp, c_vp, c_vp]),
    "sgam_forward_splat_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "sgam_forward_splat_workspace_zero_bytes": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "sgam_forward_splat_tiled_f32": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                                             ctypes.POINTER(c_f32), c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "sgam_forward_splat_tiled_srcs_f32": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                                                  ctypes.POINTER(c_f32), c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                  c_vp]),
    "sgam_depth_normalise_f32": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_i32, c_i64, c_vp]),
    "sgam_inverse_warp_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp,
                                      c_vp]),
    "sgam_inverse_warp_srcs_f32": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                                           c_vp, c_vp, c_vp]),
    "sgam_rgb_u8_to_f32": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "sgam_frame_feedback_f32": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
}

_ERRORS = {-1: "SGAM_EINVAL (bad shape / unsupported size)", -2: "SGAM_EALIGN (pointer/stride alignment)",
           -3: "SGAM_EWORKSPACE (workspace missing or too small)"}


class SgamHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load libsgam_hip.so; raise loudly if it has not been built (python -m sgam_neurips22_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  It must be in the
    # process BEFORE this library so that our DT_NEEDED libamdhip64.so.7 binds to that same runtime instance;
    # loading /opt/rocm's copy first gives two runtimes in one process (hipErrorNoDevice on the second).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise SgamHipError(
            f"{LIB_PATH} is missing: the HIP backend has not been built. Run "
            "`python -m sgam_neurips22_amd.build` (needs hipcc). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.sgam_abi_version() != ABI_VERSION:
        raise SgamHipError(f"{LIB_PATH} speaks ABI v{lib.sgam_abi_version()}, these bindings v{ABI_VERSION}: rebuild "
                           "(`python -m sgam_neurips22_amd.build`)")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = _ERRORS.get(rc, f"hipError_t {rc}" if rc > 0 else f"error {rc}")
        raise SgamHipError(f"{what} failed: {msg}")
