import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from sgam_neurips22_amd import testing
from sgam_neurips22_amd.config import default_params
from sgam_neurips22_amd.generative_sensing_module.model import VQModel
from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration, synthetic_seed_frame
p = default_params("google_earth"); m = VQModel(**p)
sd = testing.synthetic_state_dict(m.state_dict(), seed=0)
sd["quantize.embedding.weight"] = testing.codebook_from_stats(0.0, 0.5, p["n_embed"], 256, 1)
m.load_state_dict(sd); m = m.cuda().eval()
for rg in (False, True):
    sc = InfiniteSceneGeneration(m, "google_earth", output_dim=(5, 1), seed_frame=synthetic_seed_frame("google_earth", 0, 256), use_rgbd_integration=rg)
    for _ in range(4):
        out = sc.one_step_prediction(sc.next_pose(sc.curr)); sc.curr += 1
        print("rgbd" if rg else "splat", "cover", float((~out["extrapolation_mask"]).float().mean()), "depth range", float(sc.frames[sc._ordered_grid_coords[sc.curr-1]]["depth"].min()), float(sc.frames[sc._ordered_grid_coords[sc.curr-1]]["depth"].max()))
    if rg: print(sc.volume.stats(), sc.volume.dims)
