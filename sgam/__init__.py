"""Drop-in import paths of the reference: ``sgam.generative_sensing_module.model.VQModel``,
``sgam.inference_pipeline.InfiniteSceneGeneration``, ``sgam.point_rendering.warp.render_projection_from_srcs_fast``
… all resolve to the MI355X backend in ``sgam_neurips22_amd`` (same module objects, registered under both
names), so ``main_scene_generation.py`` runs unchanged from the repo root."""
import importlib
import sys

_IMPL = "sgam_neurips22_amd"
_ALIASES = [
    "generative_sensing_module",
    "generative_sensing_module.model",
    "generative_sensing_module.modules",
    "generative_sensing_module.modules.diffusionmodules",
    "generative_sensing_module.modules.diffusionmodules.model",
    "generative_sensing_module.modules.vqvae",
    "generative_sensing_module.modules.vqvae.quantize",
    "generative_sensing_module.modules.losses",                 # the YAML's lossconfig.target resolves
    "generative_sensing_module.modules.losses.vqperceptual",
    "generative_sensing_module.modules.losses.lpips",
    "generative_sensing_module.modules.discriminator",
    "generative_sensing_module.modules.discriminator.model",
    "point_rendering",
    "point_rendering.warp",
    "inference_pipeline",
]
for _name in _ALIASES:
    _mod = importlib.import_module(f"{_IMPL}.{_name}")
    sys.modules[f"{__name__}.{_name}"] = _mod
    if "." not in _name:
        globals()[_name] = _mod
