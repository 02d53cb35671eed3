"""Forward-splat conditioning warp on the HIP backend.

``render_projection_from_srcs_fast`` keeps the reference's signature and 7-tuple return
(sgam/point_rendering/warp.py:193-286).  The scatter is deterministic by construction — the largest
linear point index wins, i.e. the reference's sequential ``parallel=False`` semantics, which is also what
its ``parallel=True`` path yields under ``torch.use_deterministic_algorithms(True)``;
``parallel`` is therefore accepted and ignored.

Host-side parameter prep: the 3x3 source intrinsics are inverted with ``torch.inverse`` on the host in
fp32, exactly as the reference does (warp.py:210) — 9 floats per source, not part of the device hot path.
"""
import torch

from .. import ops

_KINV_CACHE = {}


def _host_inverse(K):
    """fp32 LAPACK inverse of (…,3,3) intrinsics on the host, memoised on the bytes of K."""
    Kc = K.detach().to("cpu", torch.float32).reshape(-1, 3, 3).contiguous()
    key = Kc.numpy().tobytes()
    inv = _KINV_CACHE.get(key)
    if inv is None:
        if len(_KINV_CACHE) > 64:
            _KINV_CACHE.clear()
        inv = torch.inverse(Kc)
        _KINV_CACHE[key] = inv
    return inv


def _kinv_on(device, K):
    return _host_inverse(K).to(device)


@torch.no_grad()
def render_projection_from_srcs_fast(src_features, src_depths, tgt_intrinsic, src_intrinsics, src2tgt_transform,
                                     src_num, dynamic_masks=None, depth_range=None, parallel=False):
    """src_features (B,N,3,H,W); src_depths (B,N,H,W); tgt_intrinsic (B,3,3); src_intrinsics (B,N,3,3);
    src2tgt_transform (B,N,4,4).  Returns (merge_depths (B,1,H,W), merge_feats (B,3,H,W),
    extrapolation_mask bool (B,1,H,W), in-bounds mask (B*N*H*W,) bool, fused feats (B,N*H*W,3),
    idx (M,3) int64 rows [b,x,y], projected_features (B,3,H,W))."""
    if dynamic_masks is not None:
        raise NotImplementedError("dynamic_masks is never passed on the SGAM path (model.py:206)")
    B, N, H, W = src_depths.shape
    dev = src_depths.device
    Kinv = _kinv_on(dev, src_intrinsics)
    T = src2tgt_transform.reshape(B * N, 4, 4)
    o = ops.forward_splat(src_features, src_depths, tgt_intrinsic.reshape(B, 3, 3), Kinv, T, depth_range=depth_range,
                          want=("merge_depths", "merge_feats", "extrap", "proj_feats", "inb_mask", "pix_xy"))
    mask = o["inb_mask"].bool()
    # by-products in the reference's layouts (index plumbing only: boolean compaction keeps point order)
    bidx = torch.arange(B, device=dev).view(B, 1).expand(B, H * W * N).reshape(-1, 1)
    idx = torch.cat([bidx, o["pix_xy"].long()], 1)[mask]
    fused = src_features.reshape(B, N, 3, H * W).permute(0, 3, 1, 2).reshape(B, H * W * N, 3)
    return o["merge_depths"], o["merge_feats"], o["extrap"].bool(), mask, fused, idx, o["proj_feats"]


@torch.no_grad()
def splat_to_model_input(batch, dataset, depth_range=None):
    """The get_x hot path (model.py:184-229 without the by-products): batch -> (x (B,4,H,W) =
    cat(warped rgb, normalised inverse depth with holes=-2), extrapolation mask bool, normalised depth)."""
    if "_src_list" in batch:           # the scene loop: per-frame tensors read in place through a pointer table
        feats, depths = batch["_src_list"]
        B = batch["R_rels"].shape[0]
        T = batch["_T_src2tgt"].reshape(len(feats), 4, 4)
        o = ops.forward_splat_srcs(feats, depths, batch["Ks"][:, 0], batch["_src_Kinv"], T, B=B, depth_range=depth_range,
                                   dataset=dataset, want=("x", "extrap"), extrap_bool=True, out=batch.get("_warp_out"))
        return o["x"], o["extrap"], o["x"][:, 3:4]
    src = batch["src_imgs"]            # (B,N,H,W,3) channels-last, read in place by the kernel
    dep = batch["src_depths"]          # (B,N,H,W,1) or (B,N,H,W)
    if dep.dim() == 5:
        dep = dep[..., 0]
    B, N, H, W = dep.shape
    dev = dep.device
    Ks = batch["Ks"]
    Kinv = batch["_src_Kinv"] if "_src_Kinv" in batch else _kinv_on(dev, Ks)
    # T_src2tgt = [R | t; 0 0 0 1]  (model.py:190-194) — data movement only
    if "_T_src2tgt" in batch:          # the scene loop assembles it on the host and uploads it with the poses
        T = batch["_T_src2tgt"].reshape(B * N, 4, 4)
    else:
        T = torch.zeros((B * N, 4, 4), device=dev, dtype=torch.float32)
        T[:, :3, :3] = batch["R_rels"].reshape(B * N, 3, 3)
        T[:, :3, 3] = batch["t_rels"].reshape(B * N, 3)
        T[:, 3, 3] = 1.0
    o = ops.forward_splat(src, dep, Ks[:, 0].to(dev), Kinv.to(dev), T, channels_last=True, depth_range=depth_range,
                          dataset=dataset, want=("x", "extrap"), extrap_bool=True)
    x = o["x"]
    return x, o["extrap"], x[:, 3:4]
