"""Build-container-only helper: import the reference (/root/reference) with sys.modules
stubs for the packages this image lacks.  Used only by
gen_golden.py to emit golden vectors; never shipped to / executed on the GPU box."""
import os
import sys
import types

import torch.nn as nn
import yaml

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "sgam"))


def install():
    # the repo root carries drop-in alias packages named like the reference's
    # (sgam/, data/); REF goes first on sys.path so the REFERENCE ones win in this process.
    for name in [m for m in sys.modules if m == "sgam" or m.startswith("sgam.") or m == "data"
                 or m.startswith("data.")]:
        del sys.modules[name]
    sys.path.insert(0, REF)
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        global_step = 0
        global_rank = 0

        @property
        def device(self):
            return next(self.parameters()).device

    pl.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = pl
    import data  # noqa: F401  (reference package)
    import data.utils  # noqa: F401
    du = types.ModuleType("data.utils.utils")
    du.instantiate_from_config = lambda cfg: nn.Identity()
    sys.modules["data.utils.utils"] = du
    for m in ["cv2", "torchvision", "open3d"]:
        sys.modules[m] = types.ModuleType(m)
    lp = types.ModuleType("sgam.generative_sensing_module.modules.losses.lpips")

    class LPIPS(nn.Module):
        def forward(self, *a):
            return 0

    lp.LPIPS = LPIPS
    sys.modules["sgam.generative_sensing_module.modules.losses.lpips"] = lp


def load_params(dataset):
    cfg = yaml.safe_load(open(f"{REF}/trained_models/{dataset}/config.yaml"))
    p = cfg["model"]["params"]
    p["data_config"] = cfg["data"]["params"]
    p["ckpt_path"] = None
    p["online_kmeans_config"]["kmean_init_codebook_path"] = None
    return p
