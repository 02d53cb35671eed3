"""InfiniteSceneGeneration — the frame-autoregressive scene-expansion loop on the HIP backend.

Counterpart of the reference harness ``sgam/inference_pipeline.py`` (constructor :23-142, prepare_grid
:157-204, zig_zag_order :452-474, get_src_grid_coords :507-531, prepare_batch_data :533-609,
inverse_warping :662-743, one_step_prediction :860-926, save_to_disk :928-959), rebuilt around an
IN-MEMORY frame store: generated frames never leave HBM between steps.  What the reference does through
its PNG / NPY round trip is reproduced bit-for-bit on the GPU by ``ops.frame_feedback``:
RGB is quantised to uint8 by truncation and re-expanded as float32(u/127.5-1), depth is de-normalised in
the reference's fp32 expression order.  Files are written only by ``export_to_disk`` after the run.

Pose grid, zig-zag order, source selection (radius 0.3 / 1.0, nearest ``num_src``) and
``T_rel = T_tgt @ inv(T_src)`` are the reference's float64 numpy formulas (host logic, a few 4x4s per
step).  ``use_rgbd_integration=True`` (reference :745-838: Open3D TSDF fusion of the source frames, mesh
extraction and an off-screen depth render at the target pose) runs on the device: ``tsdf.TsdfVolume``
integrates the source depth maps with Open3D's published rule and ray-casts the fused surface
(csrc/tsdf.hip; SURVEY §8 f1 — parity pinned to oracle/tsdf.py and analytic scenes, not to Open3D itself);
a ``tgt_depth_provider`` callback can still replace it.  ``inverse_warping`` is a HIP kernel.
"""
import os
from pathlib import Path

import numpy as np
import torch

from . import ops
from .generative_sensing_module.model import VQModel

_GL2CV = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]], dtype=np.float64)

_START = {
    "google_earth": (np.array([[1., 0., 0., -3.], [0., 0.86602527, -0.50000024, -6.],
                               [0., 0.50000024, 0.86602527, 2.], [0., 0., 0., 1.]]),
                     np.array([0., 0.11878788, 0.]), np.array([0.12, 0, 0.])),
    "clevr-infinite": (np.array([[1., 0., 0., -20.], [0., 0.95533651, -0.29552022, -20.],
                                 [0., 0.29552022, 0.95533651, 0.], [0., 0., 0., 1.]]),
                       np.array([0, 0.81632614, 0.]), np.array([0.81632614, 0, 0.])),
}


def intrinsics(data, image_resolution=(256, 256)):
    """reference :61-65 (CLEVR) and :83-89 (GoogleEarth, rows scaled by res/512)."""
    if data == "clevr-infinite":
        return np.array([[355.5555, 0, 128], [0, 355.5555, 128], [0, 0, 1]], dtype=np.float64)
    if data == "google_earth":
        K = np.array([[497.77774, 0, 256], [0, 497.77774, 256], [0, 0, 1]], dtype=np.float64)
        K[0] = K[0] * image_resolution[1] / 512
        K[1] = K[1] * image_resolution[0] / 512
        return K
    raise NotImplementedError(data)


def ray_to_z_depth(depth, K):
    """CLEVR ray-length -> z-depth conversion (reference :71-79 / :582-590), float64 numpy."""
    h, w = depth.shape[:2]
    xs, ys = np.meshgrid(np.linspace(0, w - 1, w), np.linspace(0, h - 1, h))
    return depth * K[0][0] / np.sqrt(K[0][0] ** 2 + (K[0][2] - ys - 0.5) ** 2 + (K[1][2] - xs - 0.5) ** 2)


def synthetic_seed_frame(data, seed_index=0, res=256):
    """Seeded smooth RGB-D seed frame for boxes without the reference's templates/ (bench, GPU tests):
    RGB uint8, depth fp32 in the template range ([1.4,3.4] GoogleEarth, [10.3,15.5] CLEVR)."""
    rs = np.random.RandomState(1000 + int(seed_index))        # main_scene_generation passes seed_index through argparse
    yy, xx = np.meshgrid(np.linspace(0, 1, res), np.linspace(0, 1, res), indexing="ij")
    acc = np.zeros((res, res, 4))
    for _ in range(12):
        fx, fy = rs.uniform(0.5, 6, 2)
        ph = rs.uniform(0, 2 * np.pi, 4)
        amp = rs.uniform(0.2, 1.0, 4)
        acc += amp * np.sin(2 * np.pi * (fx * xx + fy * yy)[..., None] + ph)
    acc = (acc - acc.min((0, 1))) / (acc.max((0, 1)) - acc.min((0, 1)))
    rgb = (acc[..., :3] * 255).astype(np.uint8)
    lo, hi = (1.4, 3.4) if data == "google_earth" else (10.3, 15.5)
    depth = (lo + (hi - lo) * acc[..., 3]).astype(np.float32)
    return rgb, depth


def load_template_seed(data, seed_index, image_resolution, templates_root="templates"):
    """The reference's seed frame exactly as prepare_batch_data reads it (:534-537): PIL LANCZOS resize of the
    PNG, nearest resize of the depth map.  Host codec boundary."""
    from PIL import Image
    import torch.nn.functional as F
    if data == "google_earth":
        d = Path(templates_root) / "google_earth" / f"seed{seed_index}"
        img_fn = sorted(d.glob("im*"))[0]
        dm_fn = Path(str(img_fn).replace("im", "dm").replace(".png", ".npy"))
    else:
        d = Path(templates_root) / "clevr-infinite"
        img_fn, dm_fn = d / "im_00000_00_00.png", d / "dm_00000_00_00.npy"
    rgb = np.array(Image.open(img_fn).convert("RGB").resize((image_resolution[1], image_resolution[0]),
                                                            resample=Image.LANCZOS))
    depth = np.load(dm_fn)
    if data == "clevr-infinite":  # the reference rewrites the template depth at construction (:71-79), in float64
        depth = ray_to_z_depth(depth, intrinsics(data))
    depth = F.interpolate(torch.from_numpy(depth[None, None]), size=image_resolution)[0][0].numpy().squeeze()
    # CLEVR: stays float64 — the reference keeps the seed depth in float64 through BOTH ray->z conversions (the .npy it
    # rewrites at construction, :79, and the per-load re-conversion, :582-590) and rounds to fp32 once (:607)
    return rgb, (depth if data == "clevr-infinite" else depth.astype(np.float32))


class _Lazy(dict):
    """dict whose listed keys are built on first access: the reference's batch / result dictionaries carry stacked
    copies of the source frames (`src_imgs`, `src_depths`, ...) that the device path itself never reads — they are
    materialised only for a caller that asks for them"""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._makers = {}

    def lazy(self, key, fn):
        self._makers[key] = fn
        return self

    def __missing__(self, key):
        if key in self._makers:
            self[key] = v = self._makers.pop(key)()
            return v
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._makers


class InfiniteSceneGeneration:
    def __init__(self, dynamic_model, data, topk=1, step_size_denom=2, use_rgbd_integration=False,
                 use_discriminator_loss=False, discriminator_loss_weight=0, recon_on_visible=False,
                 offscreen_rendering=True, output_dim=None, seed_index=0, num_src=None, seed_frame=None,
                 templates_root="templates", tgt_depth_provider=None, image_resolution=(256, 256),
                 trajectory_shape="grid", grid_transform_path=None):
        """`trajectory_shape` / `grid_transform_path` are this backend's spelling of what the reference hard-codes: its
        constructor sets trajectory_shape = 'grid' (:67, :82) and fills `grid_res/<data>_seed<k>` with the seed frame, so its
        'spiral' / 'cylinder' / 'trajectory' pose sets (:206-421) and the known-frame map (:144-155) are reachable only by
        editing it.  Here they are arguments: the shape selects the pose set, and a `grid_transform_path` folder holding
        `dm_<frame>_<ii>_<jj>.npy` + `im_<frame>_<ii>_<jj>.png` pairs (the layout export_to_disk writes) marks those poses
        visited and loads them into the frame store as KNOWN frames: a 'trajectory' run (which also reads `cam0_to_world.txt`
        there) warps from them; the grid / spiral / cylinder loops regenerate every pose from index 1, like the reference's —
        known frames there only seed `anchor_poses`, nothing is resumed."""
        if data not in _START:
            raise NotImplementedError(data)
        if trajectory_shape not in ("grid", "spiral", "cylinder", "trajectory"):
            raise NotImplementedError(trajectory_shape)
        self.dynamic_model, self.data, self.topk = dynamic_model, data, topk
        self.seed_index, self.step_size_denom = seed_index, step_size_denom
        self.use_rgbd_integration = use_rgbd_integration
        self.tgt_depth_provider = tgt_depth_provider
        self.image_resolution = tuple(image_resolution)
        self.output_dim = output_dim if output_dim is not None else ((20, 20) if data == "clevr-infinite" else (100, 1))
        self.K = intrinsics(data, self.image_resolution)
        self.K_inv = np.linalg.inv(self.K)
        is_vq = isinstance(dynamic_model, VQModel)
        default_src = 5 if data == "clevr-infinite" else 3
        self.num_src = (default_src if num_src is None else num_src) if is_vq else 1
        self.curr = 1
        self.trajectory_shape = trajectory_shape
        self.grid_transform_path = None if grid_transform_path is None else Path(grid_transform_path)
        self.anchor_poses = {}
        self.device = dynamic_model.device
        if seed_frame is None:
            if os.path.isdir(os.path.join(templates_root, data)):
                seed_frame = load_template_seed(data, seed_index, self.image_resolution, templates_root)
            else:
                import warnings
                warnings.warn(f"{templates_root}/{data} not found: starting from a SYNTHETIC seed frame, not the reference's "
                              "template", RuntimeWarning)
                seed_frame = synthetic_seed_frame(data, seed_index, self.image_resolution[0])
        self.frames = {}        # grid coord -> dict(rgb_f (H,W,3) fp32, depth (H,W) fp32, rgb_u8, index)
        known_map = self.get_known_map()
        if trajectory_shape == "grid":                      # reference :104-116
            self.prepare_grid(self.output_dim, known_map)
            self._ordered_grid_coords = self.zig_zag_order()
        elif trajectory_shape == "spiral":
            self.prepare_spiral(self.output_dim, known_map)
            self._ordered_grid_coords = self.zig_zag_order()
        elif trajectory_shape == "cylinder":
            self.prepare_ring(self.output_dim, known_map, horizontal_offset=0.002)
            self._ordered_grid_coords = self.zig_zag_order()
        else:
            if self.grid_transform_path is None:
                raise ValueError("trajectory_shape='trajectory' reads <grid_transform_path>/cam0_to_world.txt")
            self._ordered_grid_coords = self.prepare_trajectory(self.output_dim[0], known_map,
                                                                pose_path=self.grid_transform_path / "cam0_to_world.txt")
        self._store_seed(seed_frame)
        self._load_known_frames(known_map)
        self.dynamic_model.use_rgbd_integration = use_rgbd_integration
        self.volume = None
        self._tsdf_log = []      # the source coordinates of every integration step, in order (colour_volume replays them)
        if use_rgbd_integration and tgt_depth_provider is None:
            self.volume = self._make_volume()
        K32 = torch.from_numpy(self.K.astype(np.float32))
        # host-side fp32 inverse like the reference (warp.py:210 / inference_pipeline.py:694), done once
        self._K_dev = K32.to(self.device)
        self._Kinv_dev = torch.inverse(K32).to(self.device)
        self._Kinv_n = {}        # n sources -> (n,3,3) contiguous copy of the inverse intrinsics
        self._K_n = {}           # n sources -> (n,3,3) contiguous copy of the intrinsics
        # pinned staging ring for the per-step pose upload (see _upload)
        self._stage = [torch.empty(4096, dtype=torch.float32).pin_memory() for _ in range(8)] \
            if self.device.type == "cuda" else None
        self._stage_done = [None] * 8
        self._stage_i = 0
        H, W = self.image_resolution
        # the target view is unknown at inference: the reference feeds zeros (:560-561); constant, so made once
        self._dst_img = torch.zeros((1, H, W, 3), device=self.device)
        self._dst_depth = torch.zeros((1, H, W), device=self.device)
        self._x_dst = None       # get_x's x_dst of that constant target, kept after the first step
        # persistent outputs of the conditioning warp = the model's graph inputs (captured by address: no copy per step)
        self._warp_out = {"x": torch.empty((1, 4, H, W), device=self.device),
                          "extrap": torch.empty((1, 1, H, W), device=self.device, dtype=torch.bool),
                          "winner": torch.empty((1, H * W), device=self.device, dtype=torch.int32)} \
            if self.device.type == "cuda" else None
        if self._warp_out is not None:
            self._warp_out["x"]._sgam_persistent = self._warp_out["extrap"]._sgam_persistent = True

    # ---------------------------------------------------------------- TSDF fusion (reference :119-133, 745-838)
    # view-space z range of valid depths per dataset: the inverse-depth codec's bounds (model.py:210-229)
    _Z_RANGE = {"google_earth": (0.05, 4.8), "clevr-infinite": (1.0, 16.5)}

    def _make_volume(self, color=False):
        from .tsdf import VOLUME_PARAMS, TsdfVolume, frustum_bounds, UNIT
        voxel, trunc = VOLUME_PARAMS[self.data]
        poses = [node["T"] for row in self.transform_grid for node in row]
        H, W = self.image_resolution
        lo, hi = frustum_bounds(self.K, poses, H, W, self._Z_RANGE[self.data][1], margin=trunc + voxel * UNIT)
        # The loop's volume fuses GEOMETRY only: the conditioning path consumes nothing but the rendered depth, and colour
        # is 60 % of a voxel's bytes.  The reference's RGB8 colour (:123-131) is fused when the run's tail asks for it:
        # export_point_clouds replays the logged integrations into a colour volume (same kernels, same order).
        return TsdfVolume(voxel, trunc, lo, hi, self.device, color=color)

    def rgbd_integration(self, src_nodes, tgt_node):
        """reference :745-838: integrate every source frame of this step (again — the volume is cumulative, like the
        reference's self.volume), then render the fused surface's depth at the target pose.  (H,W) device fp32.
        One pass over the union of the units the sources open (TsdfVolume.integrate_many); poses are the nodes' own 4x4s."""
        # the reference fuses the depth as loaded (:570-574) — for the CLEVR seed that is the ONCE-converted map;
        # its second ray->z conversion (:582-590) only touches batch['src_depths'], after the fusion
        coords = [s["grid_coord"] for s in src_nodes]
        self.volume.integrate_many([self.frames[c]["depth"] for c in coords], self.K, [s["T"] for s in src_nodes],
                                   Ts_c2w=[s["T_inv"] for s in src_nodes])
        self._tsdf_log.append(coords)
        H, W = self.image_resolution
        z0, z1 = self._Z_RANGE[self.data]
        return self.volume.render_depth(self.K, tgt_node["T"], H, W, z0, z1, T_c2w=tgt_node["T_inv"])

    def colour_volume(self):
        """The fused volume WITH the reference's RGB8 colour (:123-131, 777-790): the logged integrations of the run replayed
        in order — every source frame is still in the frame store — through the colour-fusing form of the same kernels."""
        vol = self._make_volume(color=True)
        for coords in self._tsdf_log:
            nodes = [self.transform_grid[c[0]][c[1]] for c in coords]
            vol.integrate_many([self.frames[c]["depth"] for c in coords], self.K, [n["T"] for n in nodes],
                               rgbs_u8=[self.frames[c]["rgb_u8"] for c in coords], Ts_c2w=[n["T_inv"] for n in nodes])
        return vol

    # ---------------------------------------------------------------- grid / order / source choice
    def get_known_map(self):
        """reference :144-155: the frames a result folder already holds, keyed by grid coordinate — file names
        `dm_<frame index>_<ii>_<jj>.npy` (+ the `im_*.png` beside each).  No folder: nothing is known but the seed."""
        known = {}
        if self.grid_transform_path is None:
            return known
        for f in Path(self.grid_transform_path).glob("dm*"):
            idx, gi, gj = (int(v) for v in f.name[3:-4].split("_")[:3])
            # (the sibling file by NAME: the reference's str.replace("dm", "im") rewrites those letters anywhere in the path)
            known[(gi, gj)] = {"rgb_path": str(f.with_name("im" + f.name[2:-3] + "png")), "depth_path": str(f),
                               "orig_frame_idx": idx}
        return known

    def _node(self, R, t, coord, known_map):
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t
        # "T" = [R | t] world -> camera and its inverse, made once per pose (the step assembles nothing)
        node = {"R": R, "t": t, "K": self.K, "position": -R.T @ t, "visited": coord in known_map, "grid_coord": coord,
                "T": T, "T_inv": np.linalg.inv(T)}
        if coord in known_map:
            node["rgb_path"], node["depth_path"] = known_map[coord]["rgb_path"], known_map[coord]["depth_path"]
            self.anchor_poses[coord] = node
        return node

    def _load_known_frames(self, known_map):
        """the known frames of the folder enter the frame store exactly as the reference reads them back in
        prepare_batch_data (:534-537, 570-574): PIL LANCZOS resize of the PNG, nearest resize of the depth map, RGB through
        the uint8 codec.  The seed pose keeps the seed frame handed to the constructor."""
        if not known_map:
            return
        from PIL import Image
        import torch.nn.functional as F
        H, W = self.image_resolution
        for coord in sorted(known_map, key=lambda c: known_map[c]["orig_frame_idx"]):
            if coord == self._ordered_grid_coords[0] and coord in self.frames:
                continue
            if coord[0] >= len(self.transform_grid) or coord[1] >= len(self.transform_grid[coord[0]]):
                continue                                     # a frame outside this run's pose set
            rgb = np.array(Image.open(known_map[coord]["rgb_path"]).convert("RGB").resize((W, H), resample=Image.LANCZOS))
            depth = np.load(known_map[coord]["depth_path"])
            depth = F.interpolate(torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float32)[None, None]), size=(H, W))[0][0]
            u8 = torch.from_numpy(np.ascontiguousarray(rgb)).to(self.device)
            self.frames[coord] = {"rgb_u8": u8, "rgb_f": ops.rgb_u8_to_f32(u8), "depth": depth.contiguous().to(self.device),
                                  "index": known_map[coord]["orig_frame_idx"]}
            self.transform_grid[coord[0]][coord[1]]["visited"] = True

    def prepare_grid(self, grid_size, known_map=None):
        known_map = known_map or {}
        start, step_i, step_j = _START[self.data]
        step_i, step_j = step_i / self.step_size_denom, step_j / self.step_size_denom
        self.transform_grid = []
        self.anchor_poses = {}
        for i in range(grid_size[0]):
            row = []
            for j in range(grid_size[1]):
                c2w = np.eye(4)
                c2w[:3, :3] = start[:3, :3]
                c2w[:3, 3] = start[:3, 3] + step_unit(step_j, j) + step_unit(step_i, i)
                w2c = np.linalg.inv(c2w @ _GL2CV)
                row.append(self._node(w2c[:3, :3], w2c[:3, 3], (i, j), known_map))
            self.transform_grid.append(row)

    def prepare_spiral(self, grid_size, known_map=None):
        """reference :206-288: an Archimedean spiral of grid_size[0] poses in the seed camera's plane (arc 1, separation 1,
        positions theta (cos theta, sin theta) / 10 from the seed position), each camera turned about z by the reference's
        `90 - theta` — radians, as written there — one pose per row of the grid."""
        known_map = known_map or {}
        start = _START[self.data][0]
        self.transform_grid, self.anchor_poses = [], {}
        w2c = np.linalg.inv(start @ _GL2CV)
        R, t = w2c[:3, :3], w2c[:3, 3]
        origin = -R.T @ t
        arc, separation = 1, 1
        r = arc
        b = separation / (2 * np.pi)
        theta = float(r) / b
        for i in range(grid_size[0]):
            a = 90 - theta
            c2w = np.eye(4)
            c2w[:3, 3] = origin
            c2w[0, 3] += theta * np.cos(theta) / 10
            c2w[1, 3] += theta * np.sin(theta) / 10
            c2w[:3, :3] = np.array([[np.cos(a), np.sin(a), 0], [-np.sin(a), np.cos(a), 0], [0, 0, 1]])
            w2c = np.linalg.inv(c2w)
            theta += float(arc) / r
            r = b * theta
            self.transform_grid.append([self._node(w2c[:3, :3], w2c[:3, 3], (i, 0), known_map)])

    def prepare_ring(self, grid_size, known_map=None, horizontal_offset=0):
        """reference :290-360 (the 'cylinder' shape): every pose is the previous one turned by pi / 80 about the camera's x
        axis and moved by -step_i (x component replaced by `horizontal_offset`) in ITS OWN frame — a ring that closes after
        160 poses.  The reference appends every pose to ONE row while naming them (i, 0), so its own loop could not index
        them (transform_grid[i][0]); here pose i is row i, like the spiral — same poses, steppable."""
        known_map = known_map or {}
        start, step_i, _ = _START[self.data]
        step_i = step_i / self.step_size_denom
        if self.data != "google_earth":
            step_i = -step_i                                  # the reference's CLEVR branch of this shape negates it (:309)
        self.transform_grid, self.anchor_poses = [], {}
        curr = start @ _GL2CV
        theta = np.pi / 80
        rot = np.eye(4)
        rot[:3, :3] = np.array([[1, 0, 0], [0, np.cos(theta), np.sin(theta)], [0, -np.sin(theta), np.cos(theta)]])
        for i in range(grid_size[0]):
            T = np.eye(4)
            T[:3, 3] = -step_i
            T[0, 3] = horizontal_offset
            w2c = T @ rot @ np.linalg.inv(curr)
            curr = np.linalg.inv(w2c)
            self.transform_grid.append([self._node(w2c[:3, :3], w2c[:3, 3], (i, 0), known_map)])

    @staticmethod
    def load_poses(pose_file):
        """reference :362-368: one pose per line — frame index, then the row-major 4 x 4 camera-to-world matrix"""
        poses = np.loadtxt(pose_file)
        frames = poses[:, 0].astype(int)
        mats = np.reshape(poses[:, 1:], (-1, 4, 4))
        return {int(k): {"frame_idx": int(k), "pose": v} for k, v in zip(frames, mats)}

    def prepare_trajectory(self, trajectory_length, known_map, pose_path):
        """reference :370-421: `trajectory_length` consecutive poses of the file, starting at the frame index of the first
        known frame; returns the visiting order (i, 0)."""
        self.transform_grid, self.anchor_poses = [], {}
        poses = self.load_poses(pose_path)
        if not known_map:
            raise ValueError("a 'trajectory' run starts from a known frame: none found in " + str(self.grid_transform_path))
        start_idx = known_map[sorted(known_map.keys())[0]]["orig_frame_idx"]
        assert start_idx in poses
        order_idx = sorted(poses.keys())
        ptr = order_idx.index(start_idx)
        assert ptr + trajectory_length < len(order_idx)
        ordered = []
        for i in range(trajectory_length):
            w2c = np.linalg.inv(poses[order_idx[ptr + i]]["pose"])
            self.transform_grid.append([self._node(w2c[:3, :3], w2c[:3, 3], (i, 0), known_map)])
            ordered.append((i, 0))
        return ordered

    def get_closest_anchor(self, curr_node):
        """reference :423-431: the known pose nearest to a node"""
        best, res = 99999999, None
        for k in self.anchor_poses:
            d = np.linalg.norm(self.anchor_poses[k]["position"] - curr_node["position"])
            if d < best:
                best, res = d, self.anchor_poses[k]
        return res

    def zig_zag_order(self):
        rows, cols = self.output_dim
        diag = [[] for _ in range(rows + cols - 1)]
        for i in range(rows):
            for j in range(cols):
                if (i + j) % 2 == 0:
                    diag[i + j].insert(0, (i, j))
                else:
                    diag[i + j].append((i, j))
        order = [c for d in diag for c in d]
        self.transform_grid[order[0][0]][order[0][1]]["visited"] = True
        return order

    def row_major_order(self):
        """reference :477-488: boustrophedon over the rows"""
        rows, cols = self.output_dim
        order = [(i, j if i % 2 == 0 else cols - j - 1) for i in range(rows) for j in range(cols)]
        self.transform_grid[order[0][0]][order[0][1]]["visited"] = True
        return order

    def column_major_order(self):
        """reference :490-501: boustrophedon over the columns"""
        rows, cols = self.output_dim
        order = [(i if j % 2 == 0 else rows - i - 1, j) for j in range(cols) for i in range(rows)]
        self.transform_grid[order[0][0]][order[0][1]]["visited"] = True
        return order

    def next_pose(self, curr):
        return self._ordered_grid_coords[curr]

    def get_src_grid_coords(self, tgt_grid_coord):
        if getattr(self, "trajectory_shape", "grid") == "trajectory":           # reference :531: the num_src poses just behind the target
            # (the reference indexes tgt - i - 1 unchecked: close to the start that wraps to the END of the trajectory; here the
            # sources are the poses that exist behind the target — one_step_prediction checks that they hold a frame)
            return [(tgt_grid_coord[0] - i - 1, 0) for i in range(self.num_src) if tgt_grid_coord[0] - i - 1 >= 0], None
        tgt = self.transform_grid[tgt_grid_coord[0]][tgt_grid_coord[1]]
        radius = 0.3 if self.data != "clevr-infinite" else 1
        found = []
        for i in range(self.curr):
            cc = self._ordered_grid_coords[i]
            cand = self.transform_grid[cc[0]][cc[1]]
            dist = np.linalg.norm(cand["position"] - tgt["position"])
            if cand["visited"] and dist <= radius:
                found.append((cc, dist))
        found = sorted(found, key=lambda x: x[1])[: self.num_src]
        return [c for c, _ in found], None

    # ---------------------------------------------------------------- frame store
    def _store_seed(self, seed_frame):
        rgb_u8, depth = seed_frame
        u8 = torch.from_numpy(np.ascontiguousarray(rgb_u8)).to(self.device)
        fr = {"rgb_u8": u8, "rgb_f": ops.rgb_u8_to_f32(u8),  # table lookup = the PNG re-read
              "depth": torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float32)).to(self.device), "index": 0}
        if self.data == "clevr-infinite":
            # the reference converts the seed's ray depth to z AGAIN on every load (:582-590), still in float64 (the
            # once-converted map was saved as float64, :79) and rounds to fp32 only at :607.  Both fp32 maps are derived
            # from the float64 once-converted depth: `depth` feeds rgbd_integration / inverse_warping (:570-580),
            # `depth_reconv` the forward splat (batch['src_depths']).
            d64 = np.asarray(depth, dtype=np.float64)
            fr["depth_reconv"] = torch.from_numpy(ray_to_z_depth(d64, self.K).astype(np.float32)).to(self.device)
        self.frames[(0, 0)] = fr
        self.transform_grid[0][0]["visited"] = True

    def _src_depth(self, coord):
        """the depth map prepare_batch_data puts into batch['src_depths'] (the forward splat's source depth)"""
        fr = self.frames[coord]
        return fr.get("depth_reconv", fr["depth"])

    # ---------------------------------------------------------------- batch assembly
    def relative_poses(self, tgt_node, src_nodes):
        T_tgt = tgt_node["T"]
        R_rels, t_rels, T_tgt2srcs = [], [], []
        for s in src_nodes:
            T_rel = T_tgt @ np.linalg.inv(s["T"])
            T_tgt2srcs.append(np.linalg.inv(T_rel))
            R_rels.append(T_rel[:3, :3])
            t_rels.append(T_rel[:3, 3])
        return np.stack(R_rels), np.stack(t_rels), np.stack(T_tgt2srcs)

    def _upload(self, *arrays):
        """Host fp32 arrays -> device tensors through ONE asynchronous copy out of a pinned staging slot.  A pageable
        `.to(device)` makes the host wait for everything queued on the stream — i.e. for the previous frame — and the
        GPU then idles while the host enqueues the next frame's conditioning launches."""
        flat = np.concatenate([np.asarray(a, dtype=np.float32).ravel() for a in arrays])
        if self._stage is None or flat.size > self._stage[0].numel():
            devt = torch.from_numpy(flat).to(self.device)
        else:
            i = self._stage_i
            self._stage_i = (i + 1) % len(self._stage)
            if self._stage_done[i] is not None:
                self._stage_done[i].synchronize()        # only ever waits when the host is a whole ring ahead
            self._stage[i][:flat.size].copy_(torch.from_numpy(flat))
            devt = torch.empty(flat.size, dtype=torch.float32, device=self.device)
            devt.copy_(self._stage[i][:flat.size], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._stage_done[i] = ev
        outs, o = [], 0
        for a in arrays:
            outs.append(devt[o:o + a.size].view(a.shape))
            o += a.size
        return outs

    def prepare_batch_data(self, tgt_node, src_nodes, num_src):
        coords = [s["grid_coord"] for s in src_nodes]
        n = len(coords)
        R_rels, t_rels, T_tgt2srcs = self.relative_poses(tgt_node, src_nodes)
        # the sources stay where the frame store holds them: the warps read them through a pointer table
        feats = [self.frames[c]["rgb_f"] for c in coords]                                  # N x (H,W,3)
        depths = [self._src_depth(c) for c in coords]                                      # N x (H,W)
        # T_src2tgt = [R | t; 0 0 0 1] (model.py:190-194): the same fp32 values the device-side assembly would hold
        T = np.zeros((n, 4, 4), dtype=np.float32)
        T[:, :3, :3], T[:, :3, 3], T[:, 3, 3] = R_rels, t_rels, 1.0
        T_dev, T_t2s_dev, R_dev, t_dev = self._upload(T, T_tgt2srcs, R_rels, t_rels)
        if n not in self._Kinv_n:
            self._Kinv_n[n] = self._Kinv_dev.expand(n, 3, 3).contiguous()
        batch = _Lazy({
            "Ks": self._K_dev.expand(1, n, 3, 3),
            "_src_Kinv": self._Kinv_n[n], "_T_src2tgt": T_dev,
            "R_rels": R_dev[None], "t_rels": t_dev[None],
            "dst_img": self._dst_img, "dst_depth": self._dst_depth,
            "_src_list": (feats, depths), "_warp_out": self._warp_out,
        })
        batch.lazy("src_imgs", lambda: torch.stack(feats)[None])                           # (1,N,H,W,3)
        batch.lazy("src_depths", lambda: torch.stack(depths)[None])                        # (1,N,H,W)
        if self.use_rgbd_integration:
            if self.tgt_depth_provider is not None:
                tgt_depth = self.tgt_depth_provider(self, tgt_node, src_nodes, batch)      # (H,W) device fp32
            else:
                tgt_depth = self.rgbd_integration(src_nodes, tgt_node)
            # inverse_warping sees the depths as loaded (:575-580) — for the CLEVR seed the once-converted map
            if n not in self._K_n:
                self._K_n[n] = self._K_dev.expand(n, 3, 3).contiguous()
            # written straight into the rgb planes of the persistent model input (B = 1: a contiguous view)
            dst = self._warp_out["x"][:, :3] if self._warp_out is not None else None
            warped = ops.inverse_warp_srcs(feats, [self.frames[c]["depth"] for c in coords], tgt_depth[None],
                                           self._K_n[n], self._Kinv_dev[None], T_t2s_dev, out=dst)
            batch["warped_tgt_features"] = warped
            batch["warped_tgt_depth"] = tgt_depth[None]
        return batch

    def inverse_warping(self, src_imgs, src_depths, tgt_depth, src_intrinsics, tgt_intrinsic, T_tgt2srcs,
                        padding_mode='zeros', depth_threshold=100, as_numpy=True, tgt_Kinv=None):
        """reference :662-743; returns item 0 of the (B,3,H,W) result ((3,H,W) numpy like the reference, or a
        device tensor with as_numpy=False)."""
        B, N = src_imgs.shape[:2]
        dev = src_imgs.device
        # fp32 inverse on the host like the reference (:694); callers that hold it already (the scene loop: one fixed K)
        # pass tgt_Kinv and skip the device->host round trip, which would otherwise serialise host and GPU every step
        Kinv = tgt_Kinv if tgt_Kinv is not None else \
            torch.inverse(tgt_intrinsic.detach().to("cpu", torch.float32).reshape(B, 3, 3)).to(dev)
        out = ops.inverse_warp(src_imgs, src_depths, tgt_depth, src_intrinsics.reshape(B * N, 3, 3), Kinv,
                               T_tgt2srcs.reshape(B * N, 4, 4))
        return out.cpu().numpy()[0] if as_numpy else out[0]

    # ---------------------------------------------------------------- the step
    @torch.no_grad()
    def one_step_prediction(self, tgt_pose_grid_coord, save_res_to_disk=True, keep_results=False):
        """reference :860-926.  ALIASING (differs from the reference, which returns fresh tensors): `x`,
        `extrapolation_mask`, `warped_depth` are views of this scene's persistent model-input buffers and `rgbd` /
        `feature` / `pre_quantized_features` are the static outputs of the captured HIP graph — the NEXT step overwrites
        them in place.  The frame store (`self.frames`) always holds its own tensors.  A caller that keeps a result
        dictionary across steps passes keep_results=True (six device copies per step) or clones what it keeps."""
        src_coords, _ = self.get_src_grid_coords(tgt_pose_grid_coord)
        if not src_coords:       # the reference dies in np.stack([]) here (:596); e.g. the GoogleEarth spiral: poses 0.7 apart, radius 0.3
            raise ValueError(f"no visited pose within the source radius of {tuple(tgt_pose_grid_coord)}: nothing to warp from")
        missing = [c for c in src_coords if c not in self.frames]
        if missing:          # (a 'trajectory' run started without the frame of the pose behind its first target)
            raise ValueError(f"source pose(s) {missing} of target {tuple(tgt_pose_grid_coord)} hold no frame: nothing to warp from")
        tgt_meta = self.transform_grid[tgt_pose_grid_coord[0]][tgt_pose_grid_coord[1]]
        src_metas = [self.transform_grid[c[0]][c[1]] for c in src_coords]
        batch = self.prepare_batch_data(tgt_meta, src_metas, self.num_src)
        depths = batch["_src_list"][1]
        batch.lazy("src_depths", lambda: torch.stack(depths)[None][..., None])             # (1,N,H,W,1) like :868
        if self._x_dst is not None:
            batch["_x_dst"] = self._x_dst
        x, x_dst, extrapolation_mask, warped_depth = self.dynamic_model.get_x(
            batch, self.data, return_extrapolation_mask=True, no_depth_range=True, parallel=True)
        self._x_dst = x_dst
        x_sample_dets, _, pre_q, quant = self.dynamic_model(
            x, topk=self.topk, extrapolation_mask=extrapolation_mask, get_pre_quantized_feature=True,
            get_quantized_feature=True, sample_number=1)
        x_sample_det = x_sample_dets[0][0]                      # (B,4,H,W); sample number is 1
        rgb_f, depth, rgb_u8 = ops.frame_feedback(x_sample_det, self.data, want_u8=True)
        if save_res_to_disk:  # name kept from the reference; here "disk" is the in-HBM frame store
            self.save_to_store(tgt_pose_grid_coord, rgb_u8[0], rgb_f[0], depth[0])
        own = (lambda t: t.clone()) if keep_results else (lambda t: t)
        res = _Lazy({
            "rgbd": own(x_sample_dets[0].squeeze().detach()), "feature": own(quant.squeeze().detach()),
            "pre_quantized_features": own(pre_q.squeeze().detach()), "fixed": False, "x": own(x.detach()),
            "batch_R_rels": batch['R_rels'], "batch_t_rels": batch['t_rels'], "warped_depth": own(warped_depth),
            "extrapolation_mask": own(extrapolation_mask), "src_coords": src_coords,
        })
        res.lazy("batch_src_imgs", lambda: batch['src_imgs'])          # stacked copies only if the caller reads them
        res.lazy("batch_src_depths", lambda: batch['src_depths'])
        return res

    def save_to_store(self, coord, rgb_u8, rgb_f, depth):
        self.frames[coord] = {"rgb_u8": rgb_u8, "rgb_f": rgb_f, "depth": depth, "index": self.curr}
        self.transform_grid[coord[0]][coord[1]]["visited"] = True

    def _range_checkpoint(self, verified_curr):
        """split-fp32 range guard at a synchronisation point of the loop: when a kernel reported a non-finite output since
        the last checkpoint, the frames generated since then are dropped, the process switches to the fp32-in MFMA path
        (no range precondition) and the loop resumes from the last verified frame.  Returns the new `curr`."""
        if not (self.device.type == "cuda" and ops.F32_MODE == "split" and ops.f32x_range_tripped()):
            return self.curr
        import warnings
        warnings.warn(f"sgam split-fp32 path: an activation left fp16's range; regenerating frames {verified_curr}.."
                      f"{self.curr - 1} on the fp32-in MFMA path (SGAM_F32_MODE -> 'mfma')", RuntimeWarning)
        ops.set_f32_mode("mfma")
        self.dynamic_model._graphs = {}
        return self._rewind_to(verified_curr)

    def _rewind_to(self, verified_curr):
        """drop the frames generated since `verified_curr` (they may hold non-finite values); returns the new `curr`"""
        for c in self._ordered_grid_coords[verified_curr:self.curr]:
            self.frames.pop(c, None)
            self.transform_grid[c[0]][c[1]]["visited"] = False
        if self.volume is not None:            # the fused volume saw the invalid frames: start it again (every step
            self.volume = self._make_volume()  # re-integrates its own sources, :757-790)
            self._tsdf_log = []
        return verified_curr

    def scene_expansion(self, return_hs=False, range_check_every=8):
        total = self.output_dim[0] * self.output_dim[1]
        if self.device.type == "cuda":
            ops.range_flag(self.device)
        verified = self.curr
        while self.curr < total:
            self.one_step_prediction(self.next_pose(self.curr))
            self.curr += 1
            if self.curr == total or (self.curr - verified) >= range_check_every:
                self.curr = self._range_checkpoint(verified)
                verified = self.curr
                if self.volume is not None:
                    self.volume.check()            # pool overflow / samples outside the box: do not lose geometry silently
        return self.frames

    # ---------------------------------------------------------------- export (after the run)
    def export_to_disk(self, out_dir):
        """Write im_XXXXX_ii_jj.png / dm_*.npy / R_*.npy / t_*.npy like the reference's save_to_disk (:928-942)."""
        from PIL import Image
        os.makedirs(out_dir, exist_ok=True)
        for (i, j), fr in self.frames.items():
            suffix = f"_{i:02d}_{j:02d}"
            node = self.transform_grid[i][j]
            np.save(os.path.join(out_dir, f"R_{fr['index']:05d}{suffix}.npy"), node["R"])
            np.save(os.path.join(out_dir, f"t_{fr['index']:05d}{suffix}.npy"), node["t"])
            np.save(os.path.join(out_dir, f"dm_{fr['index']:05d}{suffix}.npy"), fr["depth"].cpu().numpy())
            Image.fromarray(fr["rgb_u8"].cpu().numpy()).save(os.path.join(out_dir, f"im_{fr['index']:05d}{suffix}.png"))
        return self.export_point_clouds(out_dir)


    def export_point_clouds(self, out_dir):
        """the two artefacts the reference's scene_expansion leaves behind the frames (inference_pipeline.py:441-450):
        `merged_pcds.ply` — every stored frame unprojected with its own depth, colour and pose, merged in the order of the
        frame index (the reference globs its R_<index>_*.npy files sorted: the same order) — and, on the rgbd_integration
        branch, `rgbd_integrated_mesh.ply` — the zero-crossing points of the fused volume (extracted on the device).  Returns
        {file name: number of points}."""
        from . import pointcloud
        os.makedirs(out_dir, exist_ok=True)
        pts, cols = [], []
        for (i, j), fr in sorted(self.frames.items(), key=lambda kv: (kv[1]["index"], kv[0])):
            node = self.transform_grid[i][j]
            Rt = np.eye(4)
            Rt[:3, :3], Rt[:3, 3] = node["R"], np.asarray(node["t"]).reshape(3)
            p, c = pointcloud.unproject_frame(fr["depth"].cpu().numpy(), fr["rgb_u8"].cpu().numpy(), self.K, Rt)
            pts.append(p)
            cols.append(c)
        out = {"merged_pcds.ply": pointcloud.write_ply(os.path.join(out_dir, "merged_pcds.ply"), np.concatenate(pts), np.concatenate(cols))}
        if self.use_rgbd_integration and self.volume is not None:
            pc = self.colour_volume().extract_point_cloud()
            out["rgbd_integrated_mesh.ply"] = pointcloud.write_ply(os.path.join(out_dir, "rgbd_integrated_mesh.ply"), pc["points"],
                                                                   pc.get("colors"), pc["normals"])
        return out


def step_unit(step, k):
    return step * k
