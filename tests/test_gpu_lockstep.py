"""GPU: lock-stepped scene batching (sgam_neurips22_amd.distributed.LockstepScenes; SURVEY §8e below GPU granularity) —
S trajectories through ONE launch sequence at B = S must produce, scene by scene, what each scene produces alone."""
import numpy as np
import pytest
import torch

from sgam_neurips22_amd import ops, testing
from sgam_neurips22_amd.config import default_params
from sgam_neurips22_amd.distributed import LockstepScenes
from sgam_neurips22_amd.generative_sensing_module.model import VQModel
from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration, synthetic_seed_frame

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _maxerr(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


def _model(golden):
    g = golden("vqgan_full_ge256.npz")
    p = default_params("google_earth")
    m = VQModel(**p)
    sd = testing.synthetic_state_dict(m.state_dict(), seed=0)
    sd["quantize.embedding.weight"] = testing.codebook_from_stats(float(g["zmean"]), float(g["zstd"]), 4096, 256, int(g["cb_seed"]))
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


@pytest.mark.parametrize("S,graphed,dt", [(3, False, "f32"), (4, True, "f32"), (2, True, "fp16")],
                         ids=["S3_eager_f32", "S4_graph_f32", "S2_graph_fp16"])
def test_lockstep_scenes_reproduce_their_solo_steps(golden, S, graphed, dt):
    """Every step of every lock-stepped scene against the SAME step of a solo scene started from the same frame store
    (teacher-forced on the lock-stepped scene's frames, so that a near-tie that falls the other way at one step cannot
    excuse the following ones): model input and hole mask bit for bit (the splat does not depend on the batch), latent within
    5e-5, codebook indices equal wherever the top-2 margin is >= 1e-4, RGB-D within 5e-5 and stored uint8 within the 1-LSB
    truncation boundary when the codes agree (fp32 path; the 16-bit mode is held to its own looser budget)."""
    mL, sd = _model(golden)
    mS, _ = _model(golden)
    for m in (mL, mS):
        m.set_compute_dtype(dt)
    mL.enable_hip_graph(graphed)
    seeds = [synthetic_seed_frame("google_earth", i) for i in range(S)]
    L = LockstepScenes(mL, "google_earth", seeds, output_dim=(5, 1))
    solos = [InfiniteSceneGeneration(mS, "google_earth", seed_index=i, output_dim=(5, 1), seed_frame=seeds[i]) for i in range(S)]
    cb = sd["quantize.embedding.weight"]
    tol = 5e-5 if dt == "f32" else 3e-2
    flips = 0
    for step in range(3):
        for i, solo in enumerate(solos):           # the solo scene starts this step from the lock-stepped scene's store
            solo.frames = {c: dict(fr) for c, fr in L.scenes[i].frames.items()}
            assert solo.curr == L.scenes[i].curr
        r = L.step(keep_results=True)
        assert r["x"].shape == (S, 4, 256, 256) and r["indices"].shape[0] == S
        for i, solo in enumerate(solos):
            res = solo.one_step_prediction(tuple(r["tgt"]))
            solo.curr += 1
            assert [tuple(c) for c in res["src_coords"]] == [tuple(c) for c in r["src_coords"][i]]
            assert torch.equal(res["x"], r["x"][i:i + 1]), "the conditioning warp must not depend on the batch"
            assert torch.equal(res["extrapolation_mask"], r["extrapolation_mask"][i:i + 1])
            pre_s = res["pre_quantized_features"]
            assert _maxerr(pre_s, r["pre_quantized_features"][i]) <= tol
            idx_s = torch.cdist(res["feature"].reshape(256, -1).t().double().cpu(), cb.double()).argmin(1)
            idx_l = r["indices"][i].reshape(-1).cpu()
            differ = idx_s != idx_l
            if dt == "f32":
                gap = testing.top2_relative_gap(pre_s.reshape(256, -1).t(), cb)
                assert not bool((differ & (gap >= 1e-4)).any()), gap[differ].tolist()
            else:
                assert differ.float().mean().item() <= 0.05
            flips += int(differ.sum())
            if not bool(differ.any()):
                assert _maxerr(res["rgbd"], r["rgbd"][i]) <= tol
                a = solo.frames[tuple(r["tgt"])]
                b = L.scenes[i].frames[tuple(r["tgt"])]
                du8 = (a["rgb_u8"].cpu().numpy().astype(np.int16) - b["rgb_u8"].cpu().numpy().astype(np.int16))
                assert np.abs(du8).max() <= (1 if dt == "f32" else 8)
                if dt == "f32":
                    assert (du8 != 0).mean() < 5e-3 and _maxerr(a["depth"], b["depth"]) <= 1e-3
    print(f"lockstep S={S} {dt}: near-tie flips over 3 steps: {flips}")
    # frames are the scenes' own tensors (views of one batched allocation per step, disjoint per scene)
    for i in range(S):
        assert len(L.scenes[i].frames) == 4 and L.scenes[i].curr == 4
    mL.enable_hip_graph(False)


def test_lockstep_expand_runs_the_whole_grid(golden):
    m, _ = _model(golden)
    m.enable_hip_graph(True)
    seeds = [synthetic_seed_frame("google_earth", i) for i in range(2)]
    L = LockstepScenes(m, "google_earth", seeds, output_dim=(10, 1))
    frames = L.expand()
    assert all(len(f) == 10 for f in frames)
    assert all(torch.isfinite(fr["depth"]).all() for f in frames for fr in f.values())
    assert not torch.equal(frames[0][(9, 0)]["rgb_u8"], frames[1][(9, 0)]["rgb_u8"])
    assert len(m._graphs) == 1
    m.enable_hip_graph(False)


def test_lockstep_clevr_five_sources(golden):
    """CLEVR-Infinite in lock step (2 x 2 grid walk, up to 5 sources per scene -> a pointer table of up to 2 x 3 frames here,
    16384-code quantiser): every step against the solo scene started from the same store, as above."""
    g = golden("vqgan_full_clevr256_topk1.npz")
    p = default_params("clevr-infinite")

    def model():
        m = VQModel(**p)
        sd = testing.synthetic_state_dict(m.state_dict(), seed=0)
        sd["quantize.embedding.weight"] = testing.codebook_from_stats(float(g["zmean"]), float(g["zstd"]), 16384, 256, int(g["cb_seed"]))
        m.load_state_dict(sd)
        return m.to(DEV).eval(), sd
    (mL, sd), (mS, _) = model(), model()
    seeds = [synthetic_seed_frame("clevr-infinite", i) for i in range(2)]
    L = LockstepScenes(mL, "clevr-infinite", seeds, output_dim=(2, 2))
    solos = [InfiniteSceneGeneration(mS, "clevr-infinite", seed_index=i, output_dim=(2, 2), seed_frame=seeds[i]) for i in range(2)]
    cb = sd["quantize.embedding.weight"]
    for step in range(3):
        for i, solo in enumerate(solos):
            solo.frames = {c: dict(fr) for c, fr in L.scenes[i].frames.items()}
        r = L.step(keep_results=True)
        for i, solo in enumerate(solos):
            res = solo.one_step_prediction(tuple(r["tgt"]))
            solo.curr += 1
            assert len(res["src_coords"]) == step + 1
            assert torch.equal(res["x"], r["x"][i:i + 1]) and torch.equal(res["extrapolation_mask"], r["extrapolation_mask"][i:i + 1])
            assert _maxerr(res["pre_quantized_features"], r["pre_quantized_features"][i]) <= 5e-5
            idx_s = torch.cdist(res["feature"].reshape(256, -1).t().double().cpu(), cb.double()).argmin(1)
            differ = idx_s != r["indices"][i].reshape(-1).cpu()
            gap = testing.top2_relative_gap(res["pre_quantized_features"].reshape(256, -1).t(), cb)
            assert not bool((differ & (gap >= 1e-4)).any())
            if not bool(differ.any()):
                assert _maxerr(res["rgbd"], r["rgbd"][i]) <= 5e-5
