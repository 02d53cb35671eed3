import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgam_neurips22_amd import testing, ops
from sgam_neurips22_amd.config import default_params
from sgam_neurips22_amd.generative_sensing_module.model import VQModel
from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration, synthetic_seed_frame
p = default_params("google_earth"); m = VQModel(**p)
sd = testing.synthetic_state_dict(m.state_dict(), seed=0)
sd["quantize.embedding.weight"] = testing.codebook_from_stats(0.0, 0.5, p["n_embed"], 256, 1)
m.load_state_dict(sd); m = m.cuda().eval(); m.enable_hip_graph(True)
if os.environ.get("PLANE") == "1":          # consistent geometry (bench.PlaneDepthScene) instead of the generated noise depths
    from bench import PlaneDepthScene as Scene
else:
    Scene = InfiniteSceneGeneration
sc = Scene(m, "google_earth", output_dim=(40, 1), seed_frame=synthetic_seed_frame("google_earth", 0, 256), use_rgbd_integration=True)
if os.environ.get("PLANE") == "1":
    sc.prepare_planes()
for _ in range(5):
    sc.one_step_prediction(sc.next_pose(sc.curr)); sc.curr += 1
def T(f, n=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
tgt = sc.next_pose(sc.curr)
src_coords, _ = sc.get_src_grid_coords(tgt)
tm = sc.transform_grid[tgt[0]][tgt[1]]; sm = [sc.transform_grid[c[0]][c[1]] for c in src_coords]
print("n src", len(sm))
print("rgbd_integration ms", T(lambda: sc.rgbd_integration(sm, tm)))
Tm = tm["T"]
print("  raycast ms", T(lambda: sc.volume.render_depth(sc.K, Tm, 256, 256, 0.05, 4.8)))
print("  integrate x1 ms", T(lambda: sc.volume.integrate(sc.frames[sm[0]["grid_coord"]]["depth"], sc.K, sm[0]["T"])))
print("  integrate_many ms", T(lambda: sc.volume.integrate_many([sc.frames[s["grid_coord"]]["depth"] for s in sm], sc.K, [s["T"] for s in sm])))
print("prepare_batch_data ms", T(lambda: sc.prepare_batch_data(tm, sm, sc.num_src)))
def step():
    b = sc.prepare_batch_data(tm, sm, sc.num_src); b['src_depths'] = b['src_depths'][..., None]
    return sc.dynamic_model.get_x(b, sc.data, return_extrapolation_mask=True, no_depth_range=True, parallel=True)
print("prepare+get_x ms", T(step))
x, x_dst, em, wd = step()
print("model ms", T(lambda: sc.dynamic_model(x, topk=1, extrapolation_mask=em, get_pre_quantized_feature=True, get_quantized_feature=True, sample_number=1)))
print("full step ms", T(lambda: sc.one_step_prediction(tgt, save_res_to_disk=False)))
print(sc.volume.stats())
# per-kernel durations of the conditioning path (HIP-event brackets of the library, 6 repetitions)
reps = 6
recs, br = ops.kernel_timeline(lambda: [sc.prepare_batch_data(tm, sm, sc.num_src) for _ in range(reps)])
per = {}
for name, ms, *_ in recs:
    per[name.split("<")[0] + ("<" + name.split("<")[1] if "integrate" in name else "")] = per.get(name.split("<")[0] + ("<" + name.split("<")[1] if "integrate" in name else ""), 0.0) + max(ms - br, 0.0)
print("conditioning kernels, us per step:", {k: round(1e3 * v / reps, 1) for k, v in per.items()}, "bracket us", round(1e3 * br, 2))
