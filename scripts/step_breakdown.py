"""Where one splat-branch step goes: graph-replayed VQGAN forward vs everything around it (host + small kernels)."""
import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from sgam_neurips22_amd import testing, ops
from sgam_neurips22_amd.config import default_params
from sgam_neurips22_amd.generative_sensing_module.model import VQModel
from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration, synthetic_seed_frame
p = default_params("google_earth"); m = VQModel(**p)
sd = testing.synthetic_state_dict(m.state_dict(), seed=0)
sd["quantize.embedding.weight"] = testing.codebook_from_stats(0.0, 0.5, p["n_embed"], 256, 1)
m.load_state_dict(sd); m = m.cuda().eval(); m.enable_hip_graph(True)
sc = InfiniteSceneGeneration(m, "google_earth", output_dim=(60, 1), seed_frame=synthetic_seed_frame("google_earth", 0, 256))
for _ in range(5):
    sc.one_step_prediction(sc.next_pose(sc.curr)); sc.curr += 1
def T(f, n=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
tgt = sc.next_pose(sc.curr)
src_coords, _ = sc.get_src_grid_coords(tgt)
tm = sc.transform_grid[tgt[0]][tgt[1]]; sm = [sc.transform_grid[c[0]][c[1]] for c in src_coords]
def prep():
    b = sc.prepare_batch_data(tm, sm, sc.num_src); b['src_depths'] = b['src_depths'][..., None]
    return sc.dynamic_model.get_x(b, sc.data, return_extrapolation_mask=True, no_depth_range=True, parallel=True)
print("prepare+get_x (splat) ms", T(prep))
x, x_dst, em, wd = prep()
f = lambda: sc.dynamic_model(x, topk=1, extrapolation_mask=em, get_pre_quantized_feature=True, get_quantized_feature=True, sample_number=1)
print("model (graph) ms", T(f))
out = f()
print("feedback ms", T(lambda: ops.frame_feedback(out[0][0][0], sc.data, want_u8=True)))
print("full step ms", T(lambda: sc.one_step_prediction(tgt, save_res_to_disk=False)))
# host-only cost of a step: same calls with the GPU idle between (sync each)
t = time.perf_counter()
for _ in range(20):
    sc.one_step_prediction(tgt, save_res_to_disk=False)
print("host enqueue per step ms (no sync)", (time.perf_counter() - t) / 20 * 1e3); torch.cuda.synchronize()
# host cost of hipGraphLaunch itself: replay the captured forward back to back
ent = list(m._graphs.values())[0]
g = ent[0]
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): g.replay()
t_host = (time.perf_counter() - t) / 20 * 1e3
torch.cuda.synchronize(); t_all = (time.perf_counter() - t) / 20 * 1e3
print("graph.replay(): host ms per call", t_host, " wall ms per call incl. GPU", t_all)
