"""Point-cloud artefacts of the reference's run tail (sgam/inference_pipeline.py:441-450, 1014-1063): the per-view
unprojection merged into `merged_pcds.ply` and — on the rgbd_integration branch — the fused volume's zero-crossing points in
`rgbd_integrated_mesh.ply`.  Host-side export after the run (the reference's is numpy + Open3D I/O too); the TSDF extraction
itself runs on the device bricks (csrc/tsdf.hip: tsdf_extract_kernel).

File format: what Open3D 0.15.2's `o3d.io.write_point_cloud(path, pcd)` writes by default — binary little-endian PLY,
`double x y z` [`double nx ny nz`] [`uchar red green blue`], colours as round(clamp(c, 0, 1) * 255).  Open3D is absent here,
so the byte layout is restated from its published writer, not pinned against its output."""
import numpy as np


def unproject_frame(depth, rgb_u8, K, Rt):
    """One generated frame as a coloured point cloud in world coordinates — `prepare_pcd` (inference_pipeline.py:1014-1039) in
    its own float64 expression order: pixel grid (x, y, 1) -> inv(K) @ . -> * depth -> homogeneous -> inv(Rt) @ . ; colours
    uint8 / 255.  Returns (points (H*W, 3) float64, colors (H*W, 3) float64), row-major over (y, x) like the reference."""
    depth = np.asarray(depth)
    h, w = depth.shape
    xs, ys = np.meshgrid(np.linspace(0, w - 1, w), np.linspace(0, h - 1, h))
    homo = np.ones([h, w, 3])
    homo[:, :, 0], homo[:, :, 1] = xs, ys
    tgt_2d = homo.reshape([w * h, 3]).T
    d = depth.reshape([h * w, 1]).T
    cam = np.linalg.inv(K) @ tgt_2d
    cam = np.multiply(d.repeat(3, 0), cam)
    cam_h = np.ones([4, cam.shape[1]])
    cam_h[:3, :] = cam
    world = (np.linalg.inv(Rt) @ cam_h)[:3]
    colors = np.asarray(rgb_u8).reshape([h * w, 3]).T / 255.
    return world.T, colors.T


def write_ply(path, points, colors=None, normals=None):
    """binary little-endian PLY in Open3D's vertex layout (module docstring)"""
    points = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    n = points.shape[0]
    fields = [("x", "<f8"), ("y", "<f8"), ("z", "<f8")]
    header = ["ply", "format binary_little_endian 1.0", "comment Created by Open3D", f"element vertex {n}",
              "property double x", "property double y", "property double z"]
    if normals is not None:
        fields += [("nx", "<f8"), ("ny", "<f8"), ("nz", "<f8")]
        header += ["property double nx", "property double ny", "property double nz"]
    if colors is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
        header += ["property uchar red", "property uchar green", "property uchar blue"]
    header.append("end_header")
    rec = np.zeros((n,), dtype=np.dtype(fields))
    rec["x"], rec["y"], rec["z"] = points[:, 0], points[:, 1], points[:, 2]
    if normals is not None:
        nr = np.asarray(normals, dtype=np.float64).reshape(-1, 3)
        rec["nx"], rec["ny"], rec["nz"] = nr[:, 0], nr[:, 1], nr[:, 2]
    if colors is not None:
        c8 = np.round(np.clip(np.asarray(colors, dtype=np.float64).reshape(-1, 3), 0.0, 1.0) * 255.0).astype(np.uint8)
        rec["red"], rec["green"], rec["blue"] = c8[:, 0], c8[:, 1], c8[:, 2]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(rec.tobytes())
    return n


def read_ply(path):
    """the inverse of write_ply (tests and downstream tooling): dict of arrays"""
    with open(path, "rb") as f:
        raw = f.read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    lines = raw[:end].decode("ascii").splitlines()
    n = int(next(ln for ln in lines if ln.startswith("element vertex")).split()[-1])
    kinds = {"double": "<f8", "float": "<f4", "uchar": "u1"}
    fields = [(ln.split()[2], kinds[ln.split()[1]]) for ln in lines if ln.startswith("property")]
    rec = np.frombuffer(raw[end:], dtype=np.dtype(fields), count=n)
    out = {"points": np.stack([rec["x"], rec["y"], rec["z"]], 1)}
    if "nx" in rec.dtype.names:
        out["normals"] = np.stack([rec["nx"], rec["ny"], rec["nz"]], 1)
    if "red" in rec.dtype.names:
        out["colors_u8"] = np.stack([rec["red"], rec["green"], rec["blue"]], 1)
    return out
