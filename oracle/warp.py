"""ORACLE (test infrastructure only) — ctypes front-end of warp_oracle.c plus the
element-wise depth (de)normalisation of the reference, restated with torch-CPU
fp32 ops in the reference's exact expression order.

Reference: sgam/point_rendering/warp.py:193-286, sgam/inference_pipeline.py:662-743,
sgam/generative_sensing_module/model.py:179-269, sgam/inference_pipeline.py:898-911.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libwarp_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(t):
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(t, dtype=np.float32)


def forward_splat(src_features, src_depths, tgt_intrinsic, src_intrinsics, src2tgt_transform,
                  depth_range=None, want_extras=False):
    """render_projection_from_srcs_fast (warp.py:193-286), largest-point-index-wins.

    src_features (B,N,3,H,W), src_depths (B,N,H,W), tgt_intrinsic (B,3,3),
    src_intrinsics (B,N,3,3), src2tgt_transform (B,N,4,4).  Returns dict of numpy arrays.
    """
    f = _f32(src_features); d = _f32(src_depths)
    B, N, H, W = d.shape
    Kt = _f32(tgt_intrinsic).reshape(B, 3, 3)
    # the reference inverts the source intrinsics with torch (warp.py:210)
    Kinv = _f32(torch.from_numpy(_f32(src_intrinsics).reshape(B * N, 3, 3)).inverse())
    T = _f32(src2tgt_transform).reshape(B * N, 4, 4)
    md = np.empty((B, 1, H, W), np.float32); mf = np.empty((B, 3, H, W), np.float32)
    em = np.empty((B, 1, H, W), np.uint8)
    inb = np.empty((B * N * H * W,), np.uint8) if want_extras else None
    pf = np.empty((B, 3, H, W), np.float32) if want_extras else None
    pd = np.empty((B, 1, H, W), np.float32) if want_extras else None
    idx = np.empty((B * N * H * W, 3), np.int64) if want_extras else None
    n_idx = ctypes.c_int64(0)
    dr = None if depth_range is None else np.asarray(depth_range, np.float32)
    rc = _lib().oracle_forward_splat(_p(f), _p(d), _p(Kt), _p(Kinv), _p(T), B, N, H, W, _p(dr),
                                     _p(md), _p(mf), _p(em), _p(inb), _p(pf), _p(pd), _p(idx),
                                     ctypes.byref(n_idx))
    assert rc == 0
    out = {"merge_depths": md, "merge_feats": mf, "extrapolation_mask": em.astype(bool)}
    if want_extras:
        out.update(mask=inb.astype(bool), projected_features=pf, projected_depth=pd,
                   idx=idx[: n_idx.value].copy())
    return out


def inverse_warp(src_imgs, src_depths, tgt_depth, src_intrinsics, tgt_intrinsic, T_tgt2srcs):
    """InfiniteSceneGeneration.inverse_warping (inference_pipeline.py:662-743).

    src_imgs (B,N,3,H,W), src_depths (B,N,H,W), tgt_depth (B,H,W), src_intrinsics (B,N,3,3),
    tgt_intrinsic (B,3,3), T_tgt2srcs (B,N,4,4).  Returns warped (B,3,H,W) float32 (the
    reference returns item 0 of it).
    """
    im = _f32(src_imgs); d = _f32(src_depths); td = _f32(tgt_depth)
    B, N, _, H, W = im.shape
    K = _f32(src_intrinsics).reshape(B * N, 3, 3)
    Kinv = _f32(torch.from_numpy(_f32(tgt_intrinsic).reshape(B, 3, 3)).inverse())
    T = _f32(T_tgt2srcs).reshape(B * N, 4, 4)
    out = np.empty((B, 3, H, W), np.float32)
    rc = _lib().oracle_inverse_warp(_p(im), _p(d), _p(td), _p(K), _p(Kinv), _p(T), B, N, H, W,
                                    _p(out), None)
    assert rc == 0
    return out


# ---- depth <-> normalised inverse depth (model.py:210-229, inference_pipeline.py:906-911) ----
def normalise_depth(warped_depth, extrapolation_mask, dataset):
    """warped_depth (B,1,H,W) float32 tensor, mask bool tensor -> normalised inverse depth, holes=-2."""
    wd = torch.as_tensor(warped_depth, dtype=torch.float32)
    em = torch.as_tensor(extrapolation_mask, dtype=torch.bool)
    if dataset == "google_earth":
        wd = 1 / (wd + 10)
        wd = (wd - 1 / 14.765625) / (1 / 10.099975586 - 1 / 14.765625)
    elif dataset == "clevr-infinite":
        wd = 1 / torch.clip(wd, 1e-7)
        wd = (wd - 1 / 16) / (1 / 7 - 1 / 16)
    else:
        raise NotImplementedError
    wd = 2 * wd - 1
    return wd * ~em + torch.ones_like(wd) * (-2) * em


def denormalise_depth(x3, dataset):
    """channel 3 of the decoder output -> metric depth (inference_pipeline.py:906-911)."""
    x3 = torch.as_tensor(x3, dtype=torch.float32)
    if dataset == "clevr-infinite":
        return 1 / ((x3 + 1) / 2 * (1 / 7 - 1 / 16) + 1 / 16)
    if dataset == "google_earth":
        return 1 / ((x3 + 1) / 2 * (1 / 10.099975586 - 1 / 14.765625) + 1 / 14.765625) - 10
    raise NotImplementedError


def rgb_to_uint8(x_rgb):
    """(3,H,W) in [-1,1] -> (H,W,3) uint8 by truncation (inference_pipeline.py:898-901)."""
    x = torch.as_tensor(x_rgb, dtype=torch.float32)
    rgb = np.clip(((x + 1) / 2 * 255.).permute(1, 2, 0).numpy(), 0, 255)
    return rgb.astype(np.uint8)
