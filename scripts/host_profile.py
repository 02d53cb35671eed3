"""cProfile of the host side of the scene loop (GPU work enqueued asynchronously): where the Python time of a step goes."""
import cProfile, pstats, sys; sys.path.insert(0, "/root/repo")
import torch
from sgam_neurips22_amd import testing
from sgam_neurips22_amd.config import default_params
from sgam_neurips22_amd.generative_sensing_module.model import VQModel
from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration, synthetic_seed_frame
p = default_params("google_earth"); m = VQModel(**p)
sd = testing.synthetic_state_dict(m.state_dict(), seed=0)
sd["quantize.embedding.weight"] = testing.codebook_from_stats(0.0, 0.5, p["n_embed"], 256, 1)
m.load_state_dict(sd); m = m.cuda().eval(); m.enable_hip_graph(True)
sc = InfiniteSceneGeneration(m, "google_earth", output_dim=(80, 1), seed_frame=synthetic_seed_frame("google_earth", 0, 256))
for _ in range(6):
    sc.one_step_prediction(sc.next_pose(sc.curr)); sc.curr += 1
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(40):
    sc.one_step_prediction(sc.next_pose(sc.curr)); sc.curr += 1
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
