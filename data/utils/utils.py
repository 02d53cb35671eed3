"""The names ``main_scene_generation.py`` star-imports from ``data.utils.utils`` (reference data/utils/utils.py:
OmegaConf :17, torch :20, np :18, instantiate_from_config :178-181), without the training-only dependencies
(wandb, pytorch_lightning, torchvision, omegaconf) that the inference path never touches."""
import numpy as np  # noqa: F401
import torch  # noqa: F401

from sgam_neurips22_amd.config import OmegaConf, instantiate_from_config  # noqa: F401

__all__ = ["OmegaConf", "torch", "np", "instantiate_from_config"]
