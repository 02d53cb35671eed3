"""march statistics of the TSDF ray cast on the bench's rgbd scene: the loop runs on the product library; the state it leaves is
then ray-cast through an INSTRUMENTED build of the same library (SGAM_TSDF_DEBUG_STEPS=1 -> ablib/dbg, loaded beside the
product one) whose output is 10000 x coarse + fine steps per pixel instead of the depth."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_model, DATASET
from sgam_neurips22_amd import ops, _lib
from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration, synthetic_seed_frame
dev = torch.device("cuda", 0)
model, sd, p = build_model(dev)
model.enable_hip_graph(True)
sc = InfiniteSceneGeneration(model, DATASET, seed_index=0, output_dim=(30, 1), seed_frame=synthetic_seed_frame(DATASET, 0), use_rgbd_integration=True)
for _ in range(20):
    sc.one_step_prediction(sc.next_pose(sc.curr)); sc.curr += 1
torch.cuda.synchronize()
dbg = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ablib", "dbg", "libsgam_hip.so"))
fn = dbg.sgam_tsdf_raycast_depth_f32
fn.restype, fn.argtypes = _lib.PROTOTYPES["sgam_tsdf_raycast_depth_f32"]
vol = sc.volume
node = sc.transform_grid[sc.next_pose(sc.curr)[0]][0]
c2w = np.ascontiguousarray(node["T_inv"], dtype=np.float32)
out = torch.empty((256, 256), dtype=torch.float32, device=dev)
fx, fy, cx, cy = vol._k4(sc.K)
rc = fn(ctypes.byref(vol.grid), 256, 256, fx, fy, cx, cy, c2w.ctypes.data, 0.05, 4.8, ops._p(vol.unit_table), ops._p(vol.brick_tsdf), ops._p(out), None, None, None)
torch.cuda.synchronize()
o = out.cpu().numpy().astype(np.int64)
coarse, fine = o // 10000, o % 10000
d = vol.render_depth(sc.K, node["T"], 256, 256, 0.05, 4.8, T_c2w=node["T_inv"]).cpu().numpy()
last = sc.frames[sc._ordered_grid_coords[sc.curr - 1]]["depth"].cpu().numpy()
print("rc", rc, "bricks", vol.stats())
print(f"per pixel (all 8 segments): coarse steps mean {coarse.mean():.1f} max {coarse.max()}, fine steps mean {fine.mean():.1f} p50 {np.median(fine):.0f} p99 {np.percentile(fine, 99):.0f} max {fine.max()}")
print(f"rays that hit: {(d > 0).mean():.3f}; hit depth mean {d[d > 0].mean():.3f}; generated depth of the last frame: min {last.min():.3f} mean {last.mean():.3f} max {last.max():.3f} std {last.std():.3f}; |d/dx| mean {np.abs(np.diff(last, axis=1)).mean():.4f}")
