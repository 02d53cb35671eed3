"""TSDF fusion of the generated RGB-D frames and the depth render at the next target pose — the
``rgbd_integration`` conditioning branch (reference sgam/inference_pipeline.py:119-133 and 745-838, which
delegates to Open3D 0.15.2).  Device side: csrc/tsdf.hip; this file sizes and owns the state.

State for one scene, all in HBM and allocated once (no growth, no host round trip per frame):
  unit table   int32 [dz][dy][dx]   direct-mapped over the scene's bounding box (units of 16 voxels)
  brick pool   fp32 tsdf + fp32 weight, 16^3 voxels per brick, bump-allocated on the device
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib, ops
from ._lib import TsdfGrid, check

UNIT = 16
# (voxel_length, sdf_trunc) per dataset: reference inference_pipeline.py:119-133
VOLUME_PARAMS = {"clevr-infinite": (0.05, 0.5), "google_earth": (0.01, 0.03)}
DEPTH_TRUNC = 20.0          # RGBDImage.create_from_color_and_depth(depth_trunc=20), :772


def frustum_bounds(K, poses_w2c, H, W, z_far, margin):
    """Axis-aligned world box of the camera frusta (apex + far-plane corners) of all poses, grown by `margin`."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    corners = np.array([[(u - cx) / fx * z_far, (v - cy) / fy * z_far, z_far, 1.0]
                        for u in (0.0, W - 1.0) for v in (0.0, H - 1.0)] + [[0.0, 0.0, 0.0, 1.0]])
    pts = []
    for T in poses_w2c:
        pts.append((np.linalg.inv(np.asarray(T, dtype=np.float64)) @ corners.T).T[:, :3])
    pts = np.concatenate(pts)
    return pts.min(0) - margin, pts.max(0) + margin


class TsdfVolume:
    def __init__(self, voxel_length, sdf_trunc, lo, hi, device, max_bricks=None, memory_budget_bytes=None, color=False):
        """lo / hi: world-space box the scene can occupy (see frustum_bounds).  max_bricks defaults to what
        `memory_budget_bytes` of brick pool holds (32 KB per brick, 80 KB with colour), capped at the number of units in
        the box; the budget defaults to a quarter of the device memory that is free right now (not more than 48 GiB).
        color: also fuse RGB8 colour (TSDFVolumeColorType.RGB8, reference :123-131)."""
        self.voxel_length, self.sdf_trunc = float(voxel_length), float(sdf_trunc)
        unit_len = np.float32(voxel_length) * np.float32(UNIT)
        base = np.floor(np.asarray(lo, dtype=np.float64) / float(unit_len)).astype(np.int64)
        top = np.floor(np.asarray(hi, dtype=np.float64) / float(unit_len)).astype(np.int64)
        dims = top - base + 1
        n_units = int(dims.prod())
        if n_units >= 2 ** 31:
            raise ops.SgamHipError(f"TSDF box of {tuple(dims)} units does not fit the int32 unit table; shrink the scene box")
        if memory_budget_bytes is None:
            free = torch.cuda.mem_get_info(device)[0] if torch.device(device).type == "cuda" else 1 << 30
            memory_budget_bytes = min(48 << 30, free // 4)
        if max_bricks is None:
            max_bricks = max(1, min(n_units, memory_budget_bytes // (UNIT ** 3 * (20 if color else 8))))
        self.base, self.dims, self.max_bricks, self.device = base, dims, int(max_bricks), device
        self.grid = TsdfGrid(np.float32(voxel_length), np.float32(sdf_trunc), (ctypes.c_int32 * 3)(*map(int, base)),
                             (ctypes.c_int32 * 3)(*map(int, dims)))
        self.unit_table = torch.full((n_units,), -1, dtype=torch.int32, device=device)
        self.unit_stamp = torch.zeros((n_units,), dtype=torch.int32, device=device)
        self.counters = torch.zeros((4 * 32,), dtype=torch.int32, device=device)        # counter k at index 32 k (a cache line each)
        self.brick_tsdf = torch.full((self.max_bricks, UNIT ** 3), 2.0, dtype=torch.float32, device=device)   # 2 = unobserved
        self.brick_weight = torch.zeros((self.max_bricks, UNIT ** 3), dtype=torch.float32, device=device)
        self.brick_color = torch.zeros((self.max_bricks, UNIT ** 3, 3), dtype=torch.float32, device=device) if color else None
        self.max_list = int(min(n_units, 1 << 22))
        self.brick_list = torch.empty((self.max_list,), dtype=torch.int32, device=device)
        self.frame_id = 0
        self._ray_mult = {}      # (H, W, fx, fy, cx, cy) -> (H,W) table of the rule's depth -> camera-distance multiplier

    @staticmethod
    def _k4(K):
        K = np.asarray(K, dtype=np.float32)
        return float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])

    def integrate(self, depth, K, T_w2c, rgb_u8=None):
        """depth (H,W) fp32 device tensor, K 3x3, T_w2c 4x4 world->camera (the reference's extrinsic [R|t]); rgb_u8 (H,W,3)
        uint8 device tensor: the frame's colour, fused when the volume was built with color=True."""
        self.integrate_many([depth], K, [T_w2c], None if rgb_u8 is None else [rgb_u8])

    def integrate_many(self, depths, K, Ts_w2c, rgbs_u8=None, Ts_c2w=None):
        """The source frames of ONE step (reference :757-790: one volume.integrate per source) in one pass over the union of
        the units they open — the same voxel values as integrate() per source in list order.  Ts_c2w: the inverses, when the
        caller holds them already (float64 4x4s)."""
        n = len(depths)
        if not 0 < n <= 8 or len(Ts_w2c) != n:
            raise ops.SgamHipError("TsdfVolume.integrate_many: 1 .. 8 source frames, one pose each")
        H, W = depths[0].shape
        srcs = (_lib.TsdfSrc * n)()
        keep = []
        for k in range(n):
            d = depths[k]
            ops._need_cuda(d)
            if tuple(d.shape) != (H, W) or d.dtype != torch.float32:
                raise ops.SgamHipError("TsdfVolume.integrate_many: depth maps must be (H,W) fp32 of one size")
            d = d.contiguous()
            rgb = None
            if self.brick_color is not None:
                rgb = None if rgbs_u8 is None else rgbs_u8[k]
                if rgb is None or rgb.dtype != torch.uint8 or tuple(rgb.shape) != (H, W, 3):
                    raise ops.SgamHipError("TsdfVolume(color=True).integrate needs the frame's (H,W,3) uint8 colour")
                rgb = rgb.contiguous()
            keep.append((d, rgb))
            T = np.asarray(Ts_w2c[k], dtype=np.float64)
            Ti = np.linalg.inv(T) if Ts_c2w is None else np.asarray(Ts_c2w[k], dtype=np.float64)
            srcs[k].depth, srcs[k].rgb_u8 = ops._p(d), ops._p(rgb)
            srcs[k].cam2world[:] = Ti.astype(np.float32).ravel().tolist()       # host 4x4s: passed by value to the kernels
            srcs[k].world2cam[:] = T.astype(np.float32).ravel().tolist()
        self.frame_id += 1
        fx, fy, cx, cy = self._k4(K)
        key = (H, W, fx, fy, cx, cy)
        if key not in self._ray_mult:
            rm = torch.empty((H, W), dtype=torch.float32, device=self.device)
            check(_lib.load().sgam_tsdf_ray_mult_f32(H, W, fx, fy, cx, cy, ops._p(rm), ops._stream()), "sgam_tsdf_ray_mult_f32")
            self._ray_mult[key] = rm
        check(_lib.load().sgam_tsdf_integrate_srcs_f32(
            ctypes.byref(self.grid), srcs, n, H, W, fx, fy, cx, cy, DEPTH_TRUNC, self.frame_id,
            ops._p(self.unit_table), ops._p(self.unit_stamp), ops._p(self.counters), ops._p(self.brick_list), self.max_list,
            ops._p(self.brick_tsdf), ops._p(self.brick_weight), self.max_bricks, ops._p(self.brick_color),
            ops._p(self._ray_mult[key]), ops._stream()), "sgam_tsdf_integrate_srcs_f32")

    def render_depth(self, K, T_w2c, H, W, z_near, z_far, want_color=False, T_c2w=None, out=None):
        """View-space z of the fused surface at the pose, (H,W) fp32, 0 where nothing is hit; with want_color also the
        fused colour at the hit, (H,W,3) fp32 in 0..255.  T_c2w: the inverse pose when the caller holds it; out: destination."""
        T = np.asarray(T_w2c, dtype=np.float64)
        c2w = np.ascontiguousarray(np.linalg.inv(T) if T_c2w is None else T_c2w, dtype=np.float32)
        if out is None:
            out = torch.empty((H, W), dtype=torch.float32, device=self.device)
        assert out.shape == (H, W) and out.dtype == torch.float32 and out.is_contiguous()
        fx, fy, cx, cy = self._k4(K)
        check(_lib.load().sgam_tsdf_raycast_depth_f32(
            ctypes.byref(self.grid), H, W, fx, fy, cx, cy, c2w.ctypes.data, float(z_near), float(z_far), ops._p(self.unit_table),
            ops._p(self.brick_tsdf), ops._p(out), ops._p(self.brick_color if want_color else None),
            ops._p(col := (torch.empty((H, W, 3), dtype=torch.float32, device=self.device) if want_color else None)), ops._stream()),
            "sgam_tsdf_raycast_depth_f32")
        return (out, col) if want_color else out

    def extract_point_cloud(self):
        """`volume.extract_point_cloud()` of the reference's run tail (inference_pipeline.py:446-450): the zero crossings of the
        fused TSDF as points (float32 (n,3) world coordinates), normals (n,3) and — when colour was fused — colours (n,3) in
        0..1, in a run-independent order (sorted by (unit, voxel, axis)).  Two launches (count, then fill) and one host sync:
        an export step after the run, not part of the loop."""
        lib = _lib.load()
        counter = torch.zeros((1,), dtype=torch.int64, device=self.device)
        check(lib.sgam_tsdf_extract_points_f32(ctypes.byref(self.grid), ops._p(self.unit_table), ops._p(self.brick_tsdf), None,
                                               ops._p(counter), 0, None, None, None, None, ops._stream()), "sgam_tsdf_extract_points_f32")
        n = int(counter.item())
        pts = torch.empty((max(n, 1), 3), dtype=torch.float32, device=self.device)
        nrm = torch.empty_like(pts)
        col = torch.empty_like(pts) if self.brick_color is not None else None
        keys = torch.empty((max(n, 1),), dtype=torch.int64, device=self.device)
        counter.zero_()
        check(lib.sgam_tsdf_extract_points_f32(ctypes.byref(self.grid), ops._p(self.unit_table), ops._p(self.brick_tsdf),
                                               ops._p(self.brick_color), ops._p(counter), n, ops._p(pts), ops._p(nrm), ops._p(col),
                                               ops._p(keys), ops._stream()), "sgam_tsdf_extract_points_f32")
        if int(counter.item()) != n:
            raise ops.SgamHipError("TSDF point extraction: the volume changed between the counting and the filling pass")
        order = torch.argsort(keys[:n])
        out = {"points": pts[:n][order].cpu().numpy(), "normals": nrm[:n][order].cpu().numpy(), "keys": keys[:n][order].cpu().numpy()}
        if col is not None:
            out["colors"] = (col[:n][order] / 255.0).cpu().numpy()
        return out

    def stats(self):
        """(bricks allocated, last frame's brick count, samples outside the box, pool overflows) — host sync."""
        return tuple(int(v) for v in self.counters[::32].cpu())

    def check(self):
        """Raise when the fusion silently lost geometry: units that could not be opened because the brick pool was
        exhausted, or depth samples that fell outside the scene box (the rendered target depth then has holes there).
        One host sync: call at checkpoints of the loop, not per frame."""
        bricks, _, outside, overflow = self.stats()
        if overflow > 0:
            raise ops.SgamHipError(f"TSDF brick pool exhausted: {overflow} unit openings dropped (pool of {self.max_bricks} "
                                   f"bricks, {min(bricks, self.max_bricks)} used); raise memory_budget_bytes / max_bricks")
        if outside > 0:
            import warnings
            warnings.warn(f"TSDF: {outside} depth samples fell outside the scene box and were not fused", RuntimeWarning)
        return bricks
