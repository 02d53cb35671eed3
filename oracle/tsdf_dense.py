"""TEST INFRASTRUCTURE ONLY — an INDEPENDENT reference for the TSDF branch, deliberately NOT a twin of
csrc/tsdf.hip / oracle/tsdf.py: a dense float64 voxel grid over a box (no 16^3 units, no stride-4 unit opening, no
brick pool, no fp32 expression order), fused with Open3D 0.15.2's published per-voxel rule
(UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier, the rule ScalableTSDFVolume applies inside each
volume unit) and rendered with a different algorithm (uniform fine march + bisection on the trilinear field, vectorised
over all pixels).  Open3D itself is absent, so the branch stays "parity unpinned" at that boundary; what this file adds
is evidence that the HIP path computes the RULE and the SURFACE, not merely the same bits as its own restatement:
tests/test_gpu_tsdf.py states tolerances against it (TSDF values where both observed: 1e-5; rendered depth: a fraction
of a voxel; fused colour: 1 / 255 levels).  Imported by tests/ only.
"""
import numpy as np


class DenseTsdf:
    def __init__(self, voxel_length, sdf_trunc, lo, hi):
        self.voxel, self.trunc = float(voxel_length), float(sdf_trunc)
        self.i0 = np.floor(np.asarray(lo, np.float64) / self.voxel).astype(np.int64)
        self.n = np.floor(np.asarray(hi, np.float64) / self.voxel).astype(np.int64) - self.i0 + 1     # voxels per axis (x, y, z)
        shape = (int(self.n[2]), int(self.n[1]), int(self.n[0]))
        self.tsdf = np.zeros(shape)
        self.weight = np.zeros(shape)
        self.color = np.zeros(shape + (3,))

    def centres(self):
        ax = [(self.i0[r] + np.arange(self.n[r]) + 0.5) * self.voxel for r in range(3)]
        return ax[0][None, None, :], ax[1][None, :, None], ax[2][:, None, None]

    def integrate(self, depth, K, T_w2c, rgb_u8=None, depth_trunc=20.0):
        depth = np.asarray(depth, np.float64)
        H, W = depth.shape
        fx, fy, cx, cy = K[0][0], K[1][1], K[0][2], K[1][2]
        T = np.asarray(T_w2c, np.float64)
        X, Y, Z = self.centres()
        xc = T[0, 0] * X + T[0, 1] * Y + T[0, 2] * Z + T[0, 3]
        yc = T[1, 0] * X + T[1, 1] * Y + T[1, 2] * Z + T[1, 3]
        zc = T[2, 0] * X + T[2, 1] * Y + T[2, 2] * Z + T[2, 3]
        with np.errstate(divide="ignore", invalid="ignore"):
            uf = xc * fx / zc + cx + 0.5
            vf = yc * fy / zc + cy + 0.5
        ok = (zc > 0) & (uf >= 0.0001) & (uf < W - 0.0001) & (vf >= 0.0001) & (vf < H - 0.0001)
        u = np.where(ok, uf, 0).astype(np.int64)
        v = np.where(ok, vf, 0).astype(np.int64)
        d = depth[v, u]
        ok &= (d > 0) & (d <= depth_trunc)
        sdf = (d - zc) * np.sqrt(((u - cx) / fx) ** 2 + ((v - cy) / fy) ** 2 + 1.0)
        ok &= sdf > -self.trunc
        tv = np.minimum(1.0, sdf / self.trunc)
        w = self.weight
        self.tsdf = np.where(ok, (self.tsdf * w + tv) / (w + 1.0), self.tsdf)
        if rgb_u8 is not None:
            px = np.asarray(rgb_u8, np.float64)[v, u]
            self.color = np.where(ok[..., None], (self.color * w[..., None] + px) / (w[..., None] + 1.0), self.color)
        self.weight = np.where(ok, w + 1.0, w)

    def _field(self, p):
        """trilinear TSDF at world points p (...,3); NaN where a corner of the cell is unobserved or outside the box"""
        t = p / self.voxel - 0.5 - self.i0
        f0 = np.floor(t)
        fr = t - f0
        i = f0.astype(np.int64)
        inside = np.all((i >= 0) & (i < (self.n - 1)), axis=-1)
        i = np.where(inside[..., None], i, 0)
        out = np.zeros(p.shape[:-1])
        good = inside.copy()
        for k in range(8):
            dx, dy, dz = k & 1, (k >> 1) & 1, k >> 2
            zz, yy, xx = i[..., 2] + dz, i[..., 1] + dy, i[..., 0] + dx
            wgt = (fr[..., 0] if dx else 1 - fr[..., 0]) * (fr[..., 1] if dy else 1 - fr[..., 1]) * (fr[..., 2] if dz else 1 - fr[..., 2])
            out += wgt * self.tsdf[zz, yy, xx]
            good &= self.weight[zz, yy, xx] > 0
        return np.where(good, out, np.nan)

    def render(self, K, T_w2c, H, W, z_near, z_far, step_voxels=0.25):
        """view-space z of the first +/- crossing along every pixel ray (0 = none) and the colour of the voxel there"""
        fx, fy, cx, cy = K[0][0], K[1][1], K[0][2], K[1][2]
        c2w = np.linalg.inv(np.asarray(T_w2c, np.float64))
        v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        dirc = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones((H, W))], -1) @ c2w[:3, :3].T      # per unit of view-space z
        o = c2w[:3, 3]
        depth = np.zeros((H, W))
        done = np.zeros((H, W), bool)
        dt = step_voxels * self.voxel
        prev = np.full((H, W), np.nan)
        t = z_near
        while t < z_far and not done.all():
            val = self._field(o + dirc * t)
            hit = ~done & (prev > 0) & (val <= 0)
            if hit.any():
                lo_t, hi_t = np.full((H, W), t - dt), np.full((H, W), t)
                for _ in range(20):                      # bisection on the trilinear field
                    mid = 0.5 * (lo_t + hi_t)
                    mv = self._field(o + dirc * mid[..., None])
                    pos = mv > 0
                    lo_t = np.where(hit & pos, mid, lo_t)
                    hi_t = np.where(hit & ~pos, mid, hi_t)
                depth = np.where(hit, 0.5 * (lo_t + hi_t), depth)
                done |= hit
            prev = val
            t += dt
        pts = o + dirc * depth[..., None]
        idx = np.floor(pts / self.voxel).astype(np.int64) - self.i0
        okc = done & np.all((idx >= 0) & (idx < self.n), axis=-1)
        idx = np.where(okc[..., None], idx, 0)
        col = np.where(okc[..., None], self.color[idx[..., 2], idx[..., 1], idx[..., 0]], 0.0)
        return depth, col

    def extract_points(self):
        """zero-crossing points of the dense field by Open3D's published ExtractPointCloud rule (per observed voxel with
        |tsdf| < 0.98 and each +x / +y / +z neighbour passing the same test with the opposite sign: one point on the edge at
        the |tsdf|-weighted position, colour weighted the same way), in float64.  Returns dict(points (n,3), colors (n,3) in
        0..255, voxel (n,3) global voxel index of the edge's first end, axis (n,))."""
        t, w = self.tsdf, self.weight
        ok = (w > 0) & (t < 0.98) & (t >= -0.98)
        X, Y, Z = self.centres()
        cen = [np.broadcast_to(X, t.shape), np.broadcast_to(Y, t.shape), np.broadcast_to(Z, t.shape)]
        pts, cols, vox, axes = [], [], [], []
        for a in range(3):                      # axis 0 = x = last array dimension
            dim = 2 - a
            s0 = [slice(None)] * 3
            s1 = [slice(None)] * 3
            s0[dim], s1[dim] = slice(0, -1), slice(1, None)
            s0, s1 = tuple(s0), tuple(s1)
            f0, f1 = t[s0], t[s1]
            hit = ok[s0] & ok[s1] & (f0 * f1 < 0)
            zi, yi, xi = np.nonzero(hit)
            r0, r1 = np.abs(f0[hit]), np.abs(f1[hit])
            p = np.stack([cen[0][s0][hit], cen[1][s0][hit], cen[2][s0][hit]], 1)
            p[:, a] = (p[:, a] * r1 + (p[:, a] + self.voxel) * r0) / (r0 + r1)
            pts.append(p)
            cols.append((self.color[s0][hit] * r1[:, None] + self.color[s1][hit] * r0[:, None]) / (r0 + r1)[:, None])
            vox.append(np.stack([xi, yi, zi], 1) + self.i0)
            axes.append(np.full(len(zi), a))
        return {"points": np.concatenate(pts), "colors": np.concatenate(cols), "voxel": np.concatenate(vox), "axis": np.concatenate(axes)}
