"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy, fp32, same expression order) of csrc/tsdf.hip: Open3D 0.15.2's
published TSDF integration rule (ScalableTSDFVolume::Integrate / UniformTSDFVolume::IntegrateWith
DepthToCameraDistanceMultiplier: 16^3-voxel units, stride-4 unit opening, running weighted mean of min(1, sdf/trunc))
and the direct ray cast of the fused surface.  The reference calls Open3D for this branch
(sgam/inference_pipeline.py:119-133, 745-838); Open3D is neither vendored nor installable here, so this oracle is
**parity unpinned** at the Open3D boundary: it pins the HIP kernels to the rule as restated, and analytic scenes pin
the rule to geometry.  Imported by tests/ only.
"""
import numpy as np

F = np.float32
UR = 16


class TsdfOracle:
    def __init__(self, voxel_length, sdf_trunc):
        self.voxel, self.trunc = F(voxel_length), F(sdf_trunc)
        self.unit_len = F(self.voxel * F(UR))
        self.units = {}          # (ux, uy, uz) -> [tsdf (16,16,16) indexed [z][y][x], weight]
        self.near = set()        # units the ray cast marches (everything else is crossed like empty space)

    def integrate(self, depth, K, T_w2c, depth_trunc=20.0, stride=4, rgb_u8=None):
        """rgb_u8 (H,W,3): also fuse colour, color <- (color * w + rgb) / (w + 1) (TSDFVolumeColorType::RGB8); the colour
        brick is the third entry of self.units[key]"""
        depth = np.asarray(depth, dtype=F)
        H, W = depth.shape
        fx, fy, cx, cy = F(K[0][0]), F(K[1][1]), F(K[0][2]), F(K[1][2])
        w2c = np.asarray(T_w2c, dtype=np.float64)
        c2w = np.linalg.inv(w2c).astype(F)
        w2c = w2c.astype(F)
        touched = set()
        vs, us = np.meshgrid(np.arange(0, H, stride), np.arange(0, W, stride), indexing="ij")
        d = depth[vs, us]
        ok = (d > 0) & (d <= F(depth_trunc))
        xc = ((us.astype(F) - cx) / fx) * d
        yc = ((vs.astype(F) - cy) / fy) * d
        p = [(((c2w[r, 0] * xc + c2w[r, 1] * yc) + c2w[r, 2] * d) + c2w[r, 3]).astype(F) for r in range(3)]
        lo = [np.floor((p[r] - self.trunc) / self.unit_len).astype(np.int64) for r in range(3)]
        hi = [np.floor((p[r] + self.trunc) / self.unit_len).astype(np.int64) for r in range(3)]
        for idx in zip(*np.nonzero(ok)):
            for uz in range(lo[2][idx], hi[2][idx] + 1):
                for uy in range(lo[1][idx], hi[1][idx] + 1):
                    for ux in range(lo[0][idx], hi[0][idx] + 1):
                        touched.add((ux, uy, uz))
        inv_trunc = F(1.0) / self.trunc
        safe_w, safe_h = F(W) - F(0.0001), F(H) - F(0.0001)
        ii = (np.arange(UR, dtype=F) + F(0.5)) * self.voxel
        for key in touched:
            ux, uy, uz = key
            if key not in self.units:
                self.units[key] = [np.full((UR, UR, UR), F(2.0), F), np.zeros((UR, UR, UR), F),    # 2 = unobserved
                                   np.zeros((UR, UR, UR, 3), F)]
            t, w, col = self.units[key]
            px = (F(ux) * self.unit_len + ii)[None, None, :]
            py = (F(uy) * self.unit_len + ii)[None, :, None]
            pz = (F(uz) * self.unit_len + ii)[:, None, None]
            c = [(((w2c[r, 0] * px + w2c[r, 1] * py) + w2c[r, 2] * pz) + w2c[r, 3]).astype(F) for r in range(3)]
            with np.errstate(divide="ignore", invalid="ignore"):
                uf = ((c[0] * fx) / c[2] + cx) + F(0.5)
                vf = ((c[1] * fy) / c[2] + cy) + F(0.5)
            m = (c[2] > 0) & (uf >= F(0.0001)) & (uf < safe_w) & (vf >= F(0.0001)) & (vf < safe_h)
            u = np.where(m, uf, 0).astype(np.int64)
            v = np.where(m, vf, 0).astype(np.int64)
            dd = depth[v, u]
            m &= (dd > 0) & (dd <= F(depth_trunc))
            rx = (u.astype(F) - cx) / fx
            ry = (v.astype(F) - cy) / fy
            mult = np.sqrt(((rx * rx + ry * ry) + F(1.0)).astype(F)).astype(F)
            sdf = ((dd - c[2]) * mult).astype(F)
            m &= sdf > -self.trunc
            tv = np.minimum(F(1.0), (sdf * inv_trunc).astype(F))
            new_t = ((t * w + tv) / (w + F(1.0))).astype(F)
            if rgb_u8 is not None:
                px = np.asarray(rgb_u8)[v, u].astype(F)                                       # (16,16,16,3)
                new_c = ((col * w[..., None] + px) / (w[..., None] + F(1.0))).astype(F)
                col[m] = new_c[m]
            t[m] = new_t[m]
            w[m] = (w + F(1.0))[m]
            if (new_t[m] < F(1.0)).any():
                self.near.add(key)     # holds an observed value inside the truncation band

    def _lattice(self, ix, iy, iz):
        u = self.units.get((ix >> 4, iy >> 4, iz >> 4))
        if u is None:
            return None
        x, y, z = ix & 15, iy & 15, iz & 15
        if not u[0][z, y, x] <= 1:
            return None
        return u[0][z, y, x]

    def _sample(self, p, inv_voxel):
        f, i0 = [], []
        for r in range(3):
            t = F(F(p[r] * inv_voxel) - F(0.5))
            fl = np.floor(t)
            i0.append(int(fl))
            f.append(F(t - fl))
        c = []
        for k in range(8):
            val = self._lattice(i0[0] + (k & 1), i0[1] + ((k >> 1) & 1), i0[2] + (k >> 2))
            if val is None:      # incomplete cell: nearest lattice point, if observed
                return self._lattice(i0[0] + (1 if f[0] >= F(0.5) else 0), i0[1] + (1 if f[1] >= F(0.5) else 0),
                                     i0[2] + (1 if f[2] >= F(0.5) else 0))
            c.append(F(val))
        c00 = F(c[0] + F(f[0] * F(c[1] - c[0])))
        c10 = F(c[2] + F(f[0] * F(c[3] - c[2])))
        c01 = F(c[4] + F(f[0] * F(c[5] - c[4])))
        c11 = F(c[6] + F(f[0] * F(c[7] - c[6])))
        c0 = F(c00 + F(f[1] * F(c10 - c00)))
        c1 = F(c01 + F(f[1] * F(c11 - c01)))
        return F(c0 + F(f[2] * F(c1 - c0)))

    def render_depth(self, K, T_w2c, H, W, z_near, z_far, pixels=None):
        """Ray cast; `pixels` = iterable of (v, u) restricts the (slow, pure-Python) march to those pixels."""
        fx, fy, cx, cy = F(K[0][0]), F(K[1][1]), F(K[0][2]), F(K[1][2])
        c2w = np.linalg.inv(np.asarray(T_w2c, dtype=np.float64)).astype(F)
        out = np.zeros((H, W), F)
        inv_voxel = F(1.0) / self.voxel
        fine, eps = F(F(0.5) * self.voxel), F(F(0.25) * self.voxel)
        todo = pixels if pixels is not None else [(v, u) for v in range(H) for u in range(W)]
        for v, u in todo:
            rx, ry = F((F(u) - cx) / fx), F((F(v) - cy) / fy)
            o = [c2w[r, 3] for r in range(3)]
            d = [F(F(F(c2w[r, 0] * rx) + F(c2w[r, 1] * ry)) + c2w[r, 2]) for r in range(3)]
            inv_unit = F(F(1.0) / self.unit_len)                      # the march multiplies by reciprocals fixed per ray
            with np.errstate(divide="ignore"):
                inv_d = [F(F(1.0) / d[r]) for r in range(3)]
            RS = 8
            seg_len = F(F(F(z_far) - F(z_near)) / F(RS))
            best = None
            for seg in range(RS):           # the device marches the RS segments on adjacent lanes; nearest hit wins
                t_begin = F(F(z_near) + F(F(seg) * seg_len))
                t_end = min(F(z_far), F(F(t_begin + seg_len) + F(F(2.0) * self.voxel)))
                t, prev_t, prev_val, prev_ok = t_begin, F(0), F(0), False
                while t < t_end:
                    p = [F(o[r] + F(d[r] * t)) for r in range(3)]
                    uf = [np.floor(F(p[r] * inv_unit)) for r in range(3)]
                    key = tuple(int(x) for x in uf)
                    is_open = key in self.near
                    coarse = F(z_far)
                    for r in range(3):
                        if d[r] > 0:
                            coarse = min(coarse, F(F(F(F(uf[r] + F(1)) * self.unit_len) - p[r]) * inv_d[r]))
                        elif d[r] < 0:
                            coarse = min(coarse, F(F(F(F(uf[r]) * self.unit_len) - p[r]) * inv_d[r]))
                    coarse = F(max(coarse, F(0)) + eps)
                    val = self._sample(p, inv_voxel) if is_open else None
                    ok = val is not None
                    if ok and prev_ok and prev_val > 0 and val <= 0:
                        hit = F(prev_t + F(F(t - prev_t) * F(prev_val / F(prev_val - val))))
                        if hit > 0 and (best is None or hit < best):
                            best = hit
                        break
                    prev_ok, prev_val, prev_t = ok, (val if ok else F(0)), t
                    stride = (max(fine, F(F(F(0.8) * val) * self.trunc)) if val > 0 else fine) if ok else self.voxel
                    t = F(t + (stride if is_open else coarse))
            if best is not None:
                out[v, u] = best
        return out
