"""the rgbd_integration branch of the scene loop alone (BASELINE configs[2] with --use_rgbd_integration): STEPS frames after WARMUP,
wall clock per frame; run under rocprofv3 --kernel-trace --stats for the per-kernel record (scripts/prof_rgbd.sh)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model, DATASET
from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration, synthetic_seed_frame
steps, warmup = int(os.environ.get("STEPS", 31)), int(os.environ.get("WARMUP", 3))
dev = torch.device("cuda", 0)
model, sd, p = build_model(dev)
model.enable_hip_graph(os.environ.get("NO_GRAPH", "0") != "1")
if os.environ.get("SPLAT_TOO", "1") == "1":      # the forward-splat branch on the same box, for the difference
    s0 = InfiniteSceneGeneration(model, DATASET, seed_index=0, output_dim=(steps + warmup + 4, 1), seed_frame=synthetic_seed_frame(DATASET, 0))
    for _ in range(warmup):
        s0.one_step_prediction(s0.next_pose(s0.curr)); s0.curr += 1
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        s0.one_step_prediction(s0.next_pose(s0.curr)); s0.curr += 1
    torch.cuda.synchronize()
    dt0 = time.perf_counter() - t
    print(f"splat loop: {steps / dt0:.1f} frames/s, {1e3 * dt0 / steps:.3f} ms/frame")
    del s0
sc = InfiniteSceneGeneration(model, DATASET, seed_index=0, output_dim=(steps + warmup + 4, 1), seed_frame=synthetic_seed_frame(DATASET, 0),
                             use_rgbd_integration=True)
for _ in range(warmup):
    sc.one_step_prediction(sc.next_pose(sc.curr)); sc.curr += 1
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    sc.one_step_prediction(sc.next_pose(sc.curr)); sc.curr += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"rgbd loop: {steps / dt:.1f} frames/s, {1e3 * dt / steps:.3f} ms/frame, tsdf stats {sc.volume.stats()}")
