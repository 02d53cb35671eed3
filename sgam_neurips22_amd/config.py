"""Config handling: the reference's ``trained_models/<dataset>/config.yaml`` files are consumed unchanged
(``load_config``); ``default_params`` carries the same hyper-parameters for boxes where the YAML is absent
(GPU test box, bench).  ``OmegaConf`` is a minimal stand-in for the omegaconf API that
``main_scene_generation.py`` uses (load, attribute + item access, attribute assignment, ** expansion)."""
import copy

import yaml

_DDCONFIG = {"double_z": False, "z_channels": 256, "resolution": 64, "in_channels": 4, "out_ch": 4, "ch": 128,
             "ch_mult": [1, 1, 2, 2, 4], "num_res_blocks": 2, "attn_resolutions": [16], "dropout": 0.0,
             "use_vae": False}
_N_EMBED = {"google_earth": 4096, "clevr-infinite": 16384}
_DEPTH_RANGE = {"google_earth": [0.099975586, 4.765625], "clevr-infinite": [7, 16]}


def default_params(dataset):
    """Constructor kwargs of VQModel equal to trained_models/<dataset>/config.yaml (model.params), minus the
    checkpoint path."""
    if dataset not in _N_EMBED:
        raise NotImplementedError(dataset)
    data_cfg = {"batch_size": 4, "num_workers": 0, "n_src": 1, "dataset": dataset,
                "depth_range": list(_DEPTH_RANGE[dataset]), "phase": "conditional_generation", "use_depth": True,
                "image_resolution": [256, 256]}
    return {
        "phase": "conditional_generation", "embed_dim": 256, "n_embed": _N_EMBED[dataset], "ckpt_path": None,
        "vq_step_threshold": 0, "use_extrapolation_mask": True,
        "online_kmeans_config": {"do_online_kmeans_clustering": False, "online_kmeans_word_timeout": 10,
                                 "inactive_threshold": 0.4, "train_feature_buffer_size": 1000,
                                 "kmean_init_codebook_path": None},
        "ddconfig": copy.deepcopy(_DDCONFIG), "lossconfig": {"target": None, "params": {}},
        "data_config": data_cfg,
    }


class _Node(dict):
    """dict with attribute access, recursively."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return _Node({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


class OmegaConf:
    @staticmethod
    def load(path):
        with open(path) as f:
            return _wrap(yaml.safe_load(f))

    @staticmethod
    def create(obj=None):
        return _wrap(obj or {})

    @staticmethod
    def to_container(cfg, resolve=True):
        if isinstance(cfg, dict):
            return {k: OmegaConf.to_container(v) for k, v in cfg.items()}
        if isinstance(cfg, list):
            return [OmegaConf.to_container(v) for v in cfg]
        return cfg


def load_config(path):
    """YAML -> VQModel kwargs, as main_scene_generation.prepare_vqgan does (:15-26)."""
    cfg = OmegaConf.load(path)
    params = cfg.model.params
    params.data_config = cfg.data.params
    params.setdefault("online_kmeans_config", {}).setdefault("kmean_init_codebook_path", None)
    return params


def instantiate_from_config(config):
    """data/utils/utils.py:178-181."""
    import importlib
    if "target" not in config or config["target"] is None:
        raise KeyError("Expected key `target` to instantiate.")
    module, cls = config["target"].rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)(**config.get("params", dict()))
