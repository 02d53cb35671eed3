"""Experiment: two independent scenes on ONE GPU, each on its own HIP stream with its own captured graphs — how much of
the per-scene latency-bound time can a second scene fill?  (aggregate frames/s vs one scene)"""
import sys, time; sys.path.insert(0, "/root/repo")
import torch
from sgam_neurips22_amd import testing
from sgam_neurips22_amd.config import default_params
from sgam_neurips22_amd.generative_sensing_module.model import VQModel
from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration, synthetic_seed_frame

def make():
    p = default_params("google_earth"); m = VQModel(**p)
    sd = testing.synthetic_state_dict(m.state_dict(), seed=0)
    sd["quantize.embedding.weight"] = testing.codebook_from_stats(0.0, 0.5, p["n_embed"], 256, 1)
    m.load_state_dict(sd); m = m.cuda().eval(); m.enable_hip_graph(True)
    return m

NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
models = [make() for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
scenes = []
for i in range(NS):
    with torch.cuda.stream(streams[i]):
        scenes.append(InfiniteSceneGeneration(models[i], "google_earth", seed_index=i, output_dim=(80, 1),
                                              seed_frame=synthetic_seed_frame("google_earth", i, 256)))
def step_all():
    for i in range(NS):
        with torch.cuda.stream(streams[i]):
            sc = scenes[i]
            sc.one_step_prediction(sc.next_pose(sc.curr)); sc.curr += 1
for _ in range(4): step_all()
torch.cuda.synchronize(); t = time.perf_counter()
K = 30
for _ in range(K): step_all()
torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"{NS} scene(s) on {NS} stream(s): {NS * K / dt:.1f} frames/s aggregate, {dt / K * 1e3:.3f} ms per round")
