"""oracle/tsdf.py against analytic geometry (no GPU): the restated Open3D integration rule + ray cast recover a
known surface to sub-voxel accuracy.  The Open3D boundary itself is unpinned."""
import numpy as np

from oracle.tsdf import TsdfOracle


def _K(f, c):
    return np.array([[f, 0, c], [0, f, c], [0, 0, 1]], dtype=np.float64)


def _pose(tx=0.0, ty=0.0, tz=0.0, yaw=0.0):
    """world -> camera"""
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = -R @ np.array([tx, ty, tz])
    return T


def plane_depth(K, T_w2c, H, W, z_plane):
    """view-space z of the world plane z = z_plane seen from the pose (analytic)"""
    c2w = np.linalg.inv(T_w2c)
    v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    d = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u, dtype=np.float64)], -1) @ c2w[:3, :3].T
    t = (z_plane - c2w[2, 3]) / d[..., 2]
    return np.where(t > 0, t, 0).astype(np.float32)


def test_plane_is_recovered_to_sub_voxel_accuracy():
    H = W = 48
    K = _K(60.0, 23.5)
    vol = TsdfOracle(0.05, 0.5)
    poses = [_pose(), _pose(tx=0.3, yaw=0.1), _pose(tx=-0.2, ty=0.1, yaw=-0.08)]
    for T in poses:
        vol.integrate(plane_depth(K, T, H, W, 8.0), K, T)
    assert len(vol.units) > 10
    w = np.concatenate([u[1].ravel() for u in vol.units.values()])
    assert w.max() == len(poses) and (w >= 0).all()
    T_new = _pose(tx=0.1, ty=-0.05, yaw=0.04)
    px = [(v, u) for v in (6, 17, 24, 40) for u in (5, 20, 31, 42)]
    got = vol.render_depth(K, T_new, H, W, 1.0, 16.5, pixels=px)
    want = plane_depth(K, T_new, H, W, 8.0)
    for v, u in px:
        assert got[v, u] > 0, (v, u)
        assert abs(got[v, u] - want[v, u]) < 0.5 * 0.05, (v, u, got[v, u], want[v, u])


def test_unobserved_space_renders_zero_and_truncation_is_respected():
    H = W = 32
    K = _K(40.0, 15.5)
    vol = TsdfOracle(0.05, 0.5)
    d = plane_depth(K, _pose(), H, W, 8.0)
    d[:, : W // 2] = 0.0                                   # half the view has no depth
    vol.integrate(d, K, _pose())
    out = vol.render_depth(K, _pose(), H, W, 1.0, 16.5, pixels=[(16, 4), (16, 28)])
    assert out[16, 4] == 0.0 and abs(out[16, 28] - 8.0) < 0.025
    for t, w, _col in vol.units.values():
        assert (t[w == 0] == 2.0).all() and (not (w > 0).any() or (t[w > 0].max() <= 1.0 and t[w > 0].min() > -1.0))
