"""CPU: host-side logic of the drop-in surface — config handling, module tree / state_dict keys, pose grid,
zig-zag order, source selection, relative poses, seeded weights, and the product path refusing CPU tensors."""
import os

import numpy as np
import pytest
import torch

from sgam_neurips22_amd import ops, testing
from sgam_neurips22_amd.config import OmegaConf, default_params, load_config
from sgam_neurips22_amd.generative_sensing_module.model import VQModel
from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration, intrinsics, synthetic_seed_frame


@pytest.fixture(scope="module")
def model():
    return VQModel(**default_params("google_earth"))


def test_state_dict_surface(model):
    sd = model.state_dict()
    assert len(sd) == 345 and sum(v.numel() for v in sd.values()) == 68990620
    for k in ["conv_in.weight", "encoder.down.2.attn.1.q.weight", "encoder.down.3.downsample.conv.bias",
              "encoder.mid.attn_1.proj_out.weight", "decoder.up.2.attn.2.norm.weight", "decoder.up.4.upsample.conv.weight",
              "decoder.up.0.block.2.conv2.weight", "quantize.embedding.weight", "post_quant_conv.bias",
              "encoder.down.2.block.0.nin_shortcut.weight"]:
        assert k in sd, k
    assert not any("temb" in k for k in sd)
    assert sd["quantize.embedding.weight"].shape == (4096, 256)
    assert VQModel(**default_params("clevr-infinite")).state_dict()["quantize.embedding.weight"].shape == (16384, 256)


def test_checkpoint_loading_tolerates_training_keys(model, tmp_path):
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd["loss.discriminator.main.0.weight"] = torch.zeros(3)
    sd["perceptual_loss.net.slice1.0.weight"] = torch.zeros(3)
    sd["loss.logvar"] = torch.zeros(())
    path = tmp_path / "ckpt.ckpt"
    torch.save({"state_dict": sd}, path)
    p = default_params("google_earth")
    p["ckpt_path"] = str(path)
    m = VQModel(**p)
    assert torch.equal(m.state_dict()["decoder.conv_out.weight"], sd["decoder.conv_out.weight"])


def test_yaml_config_roundtrip(tmp_path):
    import yaml
    p = default_params("google_earth")
    cfg = {"model": {"target": "sgam.generative_sensing_module.model.VQModel", "params": {k: v for k, v in p.items()
                                                                                          if k != "data_config"}},
           "data": {"target": "data.utils.utils.DataModuleFromConfig", "params": p["data_config"]}}
    f = tmp_path / "config.yaml"
    f.write_text(yaml.safe_dump(cfg))
    c = OmegaConf.load(str(f))
    c.model.params.data_config = c.data.params                    # main_scene_generation.py:23-25 idioms
    assert c.model["params"]["n_embed"] == 4096 and c.model.params.ddconfig.ch_mult == [1, 1, 2, 2, 4]
    m = VQModel(**load_config(str(f)))
    assert len(m.state_dict()) == 345


def test_synthetic_weights_are_deterministic(model):
    a = testing.synthetic_state_dict(model.state_dict(), seed=0)
    b = testing.synthetic_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not torch.equal(a["conv_in.weight"], testing.synthetic_state_dict(model.state_dict(), seed=1)["conv_in.weight"])


def test_intrinsics_and_grid(model, golden):
    K = intrinsics("google_earth")
    assert np.allclose(K, [[248.88887, 0, 128], [0, 248.88887, 128], [0, 0, 1]])
    assert intrinsics("clevr-infinite")[0, 0] == 355.5555
    scene = InfiniteSceneGeneration.__new__(InfiniteSceneGeneration)
    scene.data, scene.step_size_denom, scene.K, scene.output_dim = "google_earth", 2, K, (4, 1)
    scene.prepare_grid((4, 1))
    order = scene.zig_zag_order()
    assert order == [(0, 0), (1, 0), (2, 0), (3, 0)]
    p0, p1 = scene.transform_grid[0][0]["position"], scene.transform_grid[1][0]["position"]
    assert np.allclose(p0, [-3, -6, 2]) and np.allclose(p1 - p0, [0, 0.05939394, 0])
    # relative pose of the first step equals what the reference fed its warp (golden capture)
    tr = golden("trajectory_ge.npz")
    R, t, _ = scene.relative_poses(scene.transform_grid[1][0], [scene.transform_grid[0][0]])
    assert np.array_equal(R.astype(np.float32)[None], tr["s0.R_rels"]) and np.array_equal(t.astype(np.float32)[None], tr["s0.t_rels"])


def test_zigzag_order_2d_and_source_selection():
    scene = InfiniteSceneGeneration.__new__(InfiniteSceneGeneration)
    scene.data, scene.step_size_denom, scene.K, scene.output_dim = "clevr-infinite", 2, intrinsics("clevr-infinite"), (3, 3)
    scene.prepare_grid((3, 3))
    order = scene.zig_zag_order()
    assert order == [(0, 0), (0, 1), (1, 0), (2, 0), (1, 1), (0, 2), (1, 2), (2, 1), (2, 2)]
    scene._ordered_grid_coords, scene.num_src, scene.curr = order, 5, 4
    for c in order[:4]:
        scene.transform_grid[c[0]][c[1]]["visited"] = True
    srcs, _ = scene.get_src_grid_coords((1, 1))
    # radius 1.0, grid step 0.408: (0,1),(1,0) at 0.408; (0,0) at 0.577; (2,0) at 0.577
    assert set(srcs) == {(0, 1), (1, 0), (0, 0), (2, 0)} and set(srcs[:2]) == {(0, 1), (1, 0)}


def test_synthetic_seed_frame_ranges():
    rgb, d = synthetic_seed_frame("google_earth", 0)
    assert rgb.shape == (256, 256, 3) and rgb.dtype == np.uint8 and 1.39 < d.min() and d.max() < 3.41
    rgb2, _ = synthetic_seed_frame("google_earth", 1)
    assert not np.array_equal(rgb, rgb2)


def test_product_path_refuses_cpu_tensors(model):
    x, mask = testing.rect_hole_input(1, 64, 64)
    with pytest.raises(ops.SgamHipError, match="no CPU fallback"):
        model(x, extrapolation_mask=mask)
    with pytest.raises(ops.SgamHipError):
        ops.forward_splat(torch.zeros(1, 1, 3, 8, 8), torch.ones(1, 1, 8, 8), torch.eye(3)[None], torch.eye(3)[None],
                          torch.eye(4)[None])


# ---- host codec boundary: the seed frame as the reference reads it, the frame store as it writes it --------
def _write_template(root, data, seed_index, rgb, depth):
    from PIL import Image
    d = root / ("google_earth/seed%d" % seed_index if data == "google_earth" else "clevr-infinite")
    d.mkdir(parents=True)
    stem = "00000_00_00"
    Image.fromarray(rgb).save(d / f"im_{stem}.png")
    np.save(d / f"dm_{stem}.npy", depth)


def test_load_template_seed_google_earth(tmp_path):
    """inference_pipeline.py:534-537 of the reference: PIL LANCZOS resize of the PNG, nearest resize of the depth map"""
    from PIL import Image
    from sgam_neurips22_amd.inference_pipeline import load_template_seed
    rs = np.random.RandomState(3)
    rgb = rs.randint(0, 256, (512, 512, 3), dtype=np.uint8)
    depth = rs.uniform(1.4, 3.4, (512, 512)).astype(np.float32)
    _write_template(tmp_path, "google_earth", 2, rgb, depth)
    got_rgb, got_d = load_template_seed("google_earth", 2, (256, 256), templates_root=str(tmp_path))
    assert got_rgb.dtype == np.uint8 and got_rgb.shape == (256, 256, 3)
    assert np.array_equal(got_rgb, np.array(Image.fromarray(rgb).resize((256, 256), resample=Image.LANCZOS)))
    assert got_d.dtype == np.float32 and got_d.shape == (256, 256)
    assert np.array_equal(got_d, depth[::2, ::2])          # F.interpolate default = nearest, source index floor(i * 2)


def test_load_template_seed_clevr_keeps_float64(tmp_path):
    """the CLEVR template is ray length: converted to z-depth ONCE in float64 at load (reference :71-79) and kept in
    float64 — the second conversion and the only rounding to fp32 happen in the loop (:582-590, :607)"""
    from sgam_neurips22_amd.inference_pipeline import intrinsics, load_template_seed, ray_to_z_depth
    rs = np.random.RandomState(4)
    rgb = rs.randint(0, 256, (256, 256, 3), dtype=np.uint8)
    ray = rs.uniform(10.3, 15.5, (256, 256))               # float64 on disk, like the reference's .npy
    _write_template(tmp_path, "clevr-infinite", 0, rgb, ray)
    got_rgb, got_d = load_template_seed("clevr-infinite", 0, (256, 256), templates_root=str(tmp_path))
    assert np.array_equal(got_rgb, rgb)                     # same size: LANCZOS resize is the identity
    assert got_d.dtype == np.float64
    assert np.array_equal(got_d, ray_to_z_depth(ray, intrinsics("clevr-infinite")))
    assert (got_d < ray).all() or np.isclose(got_d, ray).any()      # z-depth <= ray length, equal only on the axis


def test_export_to_disk_writes_the_reference_layout(tmp_path):
    """save_to_disk of the reference (:928-942): im_XXXXX_ii_jj.png (lossless), dm_ / R_ / t_ .npy per visited cell; what
    is written reads back bit for bit (the codec the in-HBM frame store replaces inside the loop)"""
    from PIL import Image
    from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration
    rs = np.random.RandomState(5)
    sc = object.__new__(InfiniteSceneGeneration)            # no device: the method only walks the frame store
    sc.frames, sc.transform_grid = {}, [[None] * 3 for _ in range(2)]
    for n, (i, j) in enumerate([(0, 0), (0, 1), (1, 2)]):
        sc.frames[(i, j)] = {"index": n, "rgb_u8": torch.from_numpy(rs.randint(0, 256, (64, 64, 3), dtype=np.uint8)),
                             "depth": torch.from_numpy(rs.uniform(1, 4, (64, 64)).astype(np.float32))}
        sc.transform_grid[i][j] = {"R": rs.randn(3, 3), "t": rs.randn(3)}
    sc.K = np.array([[60.0, 0, 32], [0, 60.0, 32], [0, 0, 1]])
    sc.use_rgbd_integration, sc.volume = False, None
    counts = sc.export_to_disk(str(tmp_path / "out"))
    names = sorted(p.name for p in (tmp_path / "out").iterdir())
    assert names == sorted([f"{k}_{n:05d}_{i:02d}_{j:02d}.{e}" for n, (i, j) in enumerate([(0, 0), (0, 1), (1, 2)])
                            for k, e in (("im", "png"), ("dm", "npy"), ("R", "npy"), ("t", "npy"))] + ["merged_pcds.ply"])
    # the run tail's merged per-view point cloud (inference_pipeline.py:441-445): every frame unprojected, in frame order
    from sgam_neurips22_amd import pointcloud
    assert counts == {"merged_pcds.ply": 3 * 64 * 64}
    merged = pointcloud.read_ply(tmp_path / "out" / "merged_pcds.ply")
    fr1 = sc.frames[(0, 1)]
    Rt = np.eye(4)
    Rt[:3, :3], Rt[:3, 3] = sc.transform_grid[0][1]["R"], sc.transform_grid[0][1]["t"]
    pts, _ = pointcloud.unproject_frame(fr1["depth"].numpy(), fr1["rgb_u8"].numpy(), sc.K, Rt)
    assert np.array_equal(merged["points"][4096:8192], pts) and np.array_equal(merged["colors_u8"][4096:8192], fr1["rgb_u8"].numpy().reshape(-1, 3))
    for (i, j), fr in sc.frames.items():
        sfx = f"{fr['index']:05d}_{i:02d}_{j:02d}"
        assert np.array_equal(np.array(Image.open(tmp_path / "out" / f"im_{sfx}.png")), fr["rgb_u8"].numpy())
        assert np.array_equal(np.load(tmp_path / "out" / f"dm_{sfx}.npy"), fr["depth"].numpy())
        assert np.array_equal(np.load(tmp_path / "out" / f"R_{sfx}.npy"), sc.transform_grid[i][j]["R"])
        assert np.array_equal(np.load(tmp_path / "out" / f"t_{sfx}.npy"), sc.transform_grid[i][j]["t"])


def test_online_codebook_refresh_follows_the_reference_rule():
    """model.py:274-295 / 313-323: countdown per codeword, reset by the first image's indices; dead words replaced by
    scipy's kmeans2 centres of the buffered features once the three conditions hold"""
    from scipy.cluster.vq import kmeans2
    from sgam_neurips22_amd.training import OnlineCodebookRefresh

    class _Q:
        def __init__(self):
            self.embedding = torch.nn.Embedding(8, 4)
            self.calls = []

        def update_codebook(self, feats, idx):
            self.calls.append((np.array(feats), list(idx)))
            for i, ci in enumerate(idx):
                self.embedding.weight.data[ci] = torch.from_numpy(feats[i])

    class _M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.quantize = _Q()

    m = _M()
    cfg = {"do_online_kmeans_clustering": True, "online_kmeans_word_timeout": 2, "inactive_threshold": 0.4,
           "train_feature_buffer_size": 3, "frequency": 2, "start_global_step": 1}
    r = OnlineCodebookRefresh(m, cfg)
    rs = np.random.RandomState(0)
    feats = [rs.randn(4, 2, 2).astype(np.float32) for _ in range(6)]
    assert r.before_step(0) == 0
    r.after_forward(0, np.array([[0, 1]]), feats[0])
    assert r.features == [] and all(v == 2 for v in r.countdown.values())           # before start_global_step: nothing happens
    for step in (1, 2, 3):
        assert r.before_step(step) == 0                                           # buffer too short / words still alive
        r.after_forward(step, np.array([[0, 1, 1], [7, 7, 7]]), feats[step])      # only the FIRST image's indices count
    assert [r.countdown[k] for k in range(8)] == [1, 1, -1, -1, -1, -1, -1, -1] and len(r.features) == 3
    before = m.quantize.embedding.weight.data.clone()
    np.random.seed(123)
    n = r.before_step(4)                                                           # 6/8 dead > 0.4, 3 features, 4 % 2 == 0
    assert n == 6 and m.quantize.calls[0][1] == [2, 3, 4, 5, 6, 7]
    np.random.seed(123)
    f = np.stack(feats[1:4]).transpose(0, 2, 3, 1).reshape(-1, 4)
    want = kmeans2(f, 6, minit="points")[0]
    assert np.allclose(m.quantize.calls[0][0], want.astype(np.float32))
    assert torch.equal(m.quantize.embedding.weight.data[:2], before[:2])
    assert [r.countdown[k] for k in range(2, 8)] == [2] * 6
    assert r.before_step(5) == 0


# ------------------------------------------------------------------------------------------------ the reference's own entry script
_REF_MAIN = "/root/reference/main_scene_generation.py"


@pytest.mark.skipif(not os.path.exists(_REF_MAIN), reason="the reference checkout is only present in the build container")
@pytest.mark.parametrize("dataset,n_embed", [("google_earth", 4096), ("clevr-infinite", 16384)])
def test_reference_main_script_runs_on_this_backend_up_to_the_checkpoint(dataset, n_embed, monkeypatch):
    """the upper boundary, exercised with the REFERENCE's own `main_scene_generation.py` (read in place, nothing copied):
    with this repository first on the import path its star-import of `data.utils.utils` (OmegaConf, torch), `sgam.inference_pipeline`
    and `sgam.generative_sensing_module.model` resolve to this backend; `prepare_vqgan(data)` loads the reference's YAML, assigns
    `config.model.params.data_config`, expands `**config.model['params']` into VQModel — down to the checkpoint load, which is served
    an empty Lightning dictionary here (the trained weights are not fetchable) — and the scene class it would construct takes the
    script's exact keyword arguments."""
    import inspect
    import runpy
    import sys as _sys
    from sgam_neurips22_amd.generative_sensing_module.model import VQModel as OurVQ
    from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration as OurScene
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    monkeypatch.chdir("/root/reference")                    # the script's config paths are relative to its checkout
    monkeypatch.setattr(_sys, "path", [root] + [p for p in _sys.path if p not in ("", "/root/reference")])
    loaded = []

    def fake_load(path, *a, **k):
        loaded.append(str(path))
        return {"state_dict": {}}
    monkeypatch.setattr(torch, "load", fake_load)
    ns = runpy.run_path(_REF_MAIN, run_name="reference_main")           # (not "__main__": no argparse, no cuda)
    assert ns["VQModel"] is OurVQ and ns["InfiniteSceneGeneration"] is OurScene
    assert hasattr(ns["OmegaConf"], "load") and ns["torch"] is torch
    model = ns["prepare_vqgan"](dataset)
    assert isinstance(model, OurVQ) and model.n_embed == n_embed and model.quantize.embedding.weight.shape == (n_embed, 256)
    assert len(loaded) == 1 and loaded[0].endswith(".ckpt")             # the YAML's ckpt_path reached init_from_ckpt
    assert model.phase == "conditional_generation" and model.use_extrapolation_mask is True
    dc = model.data_config
    assert (dc["dataset"] if isinstance(dc, dict) else dc.dataset) == dataset
    with pytest.raises(NotImplementedError):
        ns["prepare_vqgan"]("kitti360")
    # the constructor call of the script, verbatim keywords (argparse hands seed_index over as a STRING)
    bound = inspect.signature(OurScene.__init__).bind(None, model, dataset, seed_index="0", use_rgbd_integration=True,
                                                      offscreen_rendering=True)
    assert bound.arguments["seed_index"] == "0"
