"""GPU: BASELINE.json's configurations AS STATED, against fixtures the reference itself produced
(tests/golden/gen_golden.py): config 2 in bf16, config 3 free-running for 32 frames, config 5 (512x512, four warp
candidates, fp32 parity + fp16 MFMA path), the top-k > 1 sampler, the CLEVR loop, and the commitment loss."""
import json
import os

import numpy as np
import pytest
import torch

from sgam_neurips22_amd import ops, testing
from sgam_neurips22_amd.config import default_params
from sgam_neurips22_amd.generative_sensing_module.model import VQModel
from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _maxerr(a, b):
    return (torch.as_tensor(a).detach().cpu().double() - torch.as_tensor(b).detach().cpu().double()).abs().max().item()


def _report(name, payload):
    """numbers the docs quote (agreement rates, drift): printed and, on the GPU box, left under gpurun_out/"""
    print(f"[{name}] {json.dumps(payload)}")
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"report_{name}.json"), "w") as f:
            json.dump(payload, f)
    except OSError:
        pass


def _model(dataset, codebook, n_embed):
    p = default_params(dataset)
    m = VQModel(**p)
    sd = testing.synthetic_state_dict(m.state_dict(), seed=0)
    sd["quantize.embedding.weight"] = codebook
    assert codebook.shape == (n_embed, 256)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd, p


def _ge_model(golden):
    g = golden("vqgan_full_ge256.npz")
    return _model("google_earth", testing.codebook_from_stats(float(g["zmean"]), float(g["zstd"]), 4096, 256, int(g["cb_seed"])), 4096)


# ------------------------------------------------------------------------------------------------ config 5
def _config5(golden):
    g = golden("config5_ge512_b4.npz")
    cb = testing.apply_codebook_repairs(testing.codebook_from_stats(float(g["zmean"]), float(g["zstd"]), 4096, 256, 0),
                                        g["repairs"], float(g["zmean"]), float(g["zstd"]))
    m, sd, p = _model("google_earth", cb, 4096)
    batch = {k: torch.from_numpy(v).to(DEV) for k, v in testing.config5_batch(g["src0_rgb"], g["src0_depth"]).items()}
    batch["src_depths"] = batch["src_depths"][..., None]
    return g, m, batch


def test_config5_fp32_parity_with_reference(golden):
    """512x512, a batch of four warp candidates (two sources each): forward splat -> get_x -> VQGAN arg-min path at
    B = 4.  The model input is bit-identical to the reference's (hash of x, mask), codebook indices are bit-exact on a
    codebook whose top-2 margin is >= 1e-4 FOR THIS INPUT (asserted by the generator and the oracle test), latent and
    RGB-D within 1e-4, commitment loss to fp32 rounding."""
    g, m, batch = _config5(golden)
    assert float(g["min_gap"]) >= 1e-4
    with torch.no_grad():
        x, x_dst, mask, wd = m.get_x(batch, "google_earth", return_extrapolation_mask=True, no_depth_range=True)
        assert x.shape == (4, 4, 512, 512) and mask.dtype == torch.bool
        assert np.array_equal(np.packbits(mask.cpu().numpy()), g["mask"])
        assert testing.sha256(x) == g["x_sha"].tobytes(), "forward splat + depth codec must be bit-exact at 512x512, B=4"
        dec, diff, idx, pre = m(x, extrapolation_mask=mask, get_codebook_count=True, get_pre_quantized_feature=True)
    assert torch.equal(idx.cpu().to(torch.int16), torch.from_numpy(g["indices"])), "codebook indices must be bit-exact"
    assert _maxerr(pre[0], g["pre_quant0"]) <= TOL and _maxerr(pre[:, ::8], g["pre_quant_sub"]) <= TOL
    assert _maxerr(dec[..., ::4, ::4], g["dec_sub"]) <= TOL
    for b in range(4):
        assert abs(float(dec[b].double().sum()) - float(g["dec_sum"][b])) <= 0.5      # 1M outputs x 1e-4 would be 100
    # diff is the loss over the whole batch; the reference ran item by item: mean of equal-sized means
    assert abs(float(diff) - float(np.mean(g["emb_loss"]))) <= 2e-6 * float(np.mean(g["emb_loss"]))


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_config5_16bit_mfma_path(golden, dt):
    """config 5 as BASELINE states it: the SAME batch through the 16-bit MFMA path (h16 convs at B = 4, fused
    sgam_attention_h16 over 16384 tokens).  Not a parity path: index agreement with the reference's indices is
    reported and held to a floor, RGB-D error is bounded on identical codes."""
    g, m, batch = _config5(golden)
    with torch.no_grad():
        x, _, mask, _ = m.get_x(batch, "google_earth", return_extrapolation_mask=True, no_depth_range=True)
        m.set_compute_dtype(dt)
        dec, _, idx, pre = m(x, extrapolation_mask=mask, get_codebook_count=True, get_pre_quantized_feature=True)
        ref_idx = torch.from_numpy(g["indices"].astype(np.int64)).to(DEV)
        dec_same = m.decode(m.quantize.get_codebook_entry(ref_idx.reshape(-1), (4, 32, 32, 256)))
    agree = (idx == ref_idx).float().mean().item()
    err_pre = _maxerr(pre[0], g["pre_quant0"]) / float(np.abs(g["pre_quant0"]).max())
    err_dec = _maxerr(dec_same[..., ::4, ::4], g["dec_sub"]) / float(np.abs(g["dec_sub"]).max())
    _report(f"config5_{dt}", {"index_agreement_vs_reference": agree, "latent_rel_err": err_pre,
                              "decoder_rel_err_same_codes": err_dec})
    assert dec.dtype == torch.float32 and torch.isfinite(dec).all()
    # floors within 3 points of / bounds 1.5x what the deterministic kernels measure (report_config5_*.json: agreement 0.9966 /
    # 0.9583, latent 7.4e-3 / 5.3e-2, decoder 2.1e-3 / 1.7e-2 for fp16 / bf16): a regression of a few percent must fail
    assert agree >= (0.99 if dt == "fp16" else 0.93)
    assert err_pre <= (1.1e-2 if dt == "fp16" else 8e-2) and err_dec <= (3.2e-3 if dt == "fp16" else 2.6e-2)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_fused_attention_h16_16384_tokens(dt):
    """sgam_attention_h16 at the 512x512 model's size (n = 16384, C = 256) against softmax(q k^T / 16) v in fp64"""
    C, n = 256, 16384
    qkv = testing.seeded_tensor("attn16.16384", (n, 3 * C)).to(DEV).to(dt)
    o = ops.attention_h16(qkv, C, C ** -0.5)
    assert o.dtype == dt and torch.equal(o, ops.attention_h16(qkv, C, C ** -0.5))
    q, k, v = (qkv[:, i * C:(i + 1) * C].double() for i in range(3))
    err = 0.0
    for r0 in range(0, n, 2048):      # 2048 x 16384 fp64 score rows at a time
        ref = torch.softmax(q[r0:r0 + 2048] @ k.t() * C ** -0.5, dim=1) @ v
        err = max(err, (o[r0:r0 + 2048].double() - ref).abs().max().item() / max(1.0, ref.abs().max().item()))
    assert err <= (3e-3 if dt == torch.float16 else 2e-2), err


def test_fused_attention_f32x_16384_tokens():
    C, n = 256, 16384
    qkv = testing.seeded_tensor("attn32.16384", (n, 3 * C)).to(DEV)
    o = ops.attention(qkv, C, C ** -0.5)
    q, k, v = (qkv[:, i * C:(i + 1) * C].double() for i in range(3))
    for r0 in range(0, n, 4096):
        ref = torch.softmax(q[r0:r0 + 4096] @ k.t() * C ** -0.5, dim=1) @ v
        assert (o[r0:r0 + 4096].double() - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())


# ------------------------------------------------------------------------------------------------ config 2 in bf16
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_config2_clevr_16bit(golden, dt):
    """BASELINE config 2 as stated: CLEVR (16384 codes — near-ties 4x denser than GoogleEarth's 4096), one conditional
    generation step through the top-k = 1 path, in bf16.  Agreement with the reference's fp32 indices is reported."""
    g = golden("vqgan_full_clevr256_topk1.npz")
    m, sd, p = _model("clevr-infinite", testing.codebook_from_stats(float(g["zmean"]), float(g["zstd"]), 16384, 256, int(g["cb_seed"])), 16384)
    m.set_compute_dtype(dt)
    x, mask = testing.rect_hole_input(1, 256, 256, seed=3)
    with torch.no_grad():
        decs, _, idx, pre, quants = m(x.to(DEV), topk=1, extrapolation_mask=mask.to(DEV), sample_number=1,
                                      get_codebook_count=True, get_pre_quantized_feature=True, get_quantized_feature=True)
        ref_idx = torch.from_numpy(g["indices"]).reshape(-1).to(DEV)
        dec_same = m.decode(m.quantize.get_codebook_entry(ref_idx, (1, 16, 16, 256)))
    agree = (idx.reshape(-1) == ref_idx).float().mean().item()
    err_pre = _maxerr(pre, g["pre_quant"]) / float(np.abs(g["pre_quant"]).max())
    err_dec = _maxerr(dec_same[..., ::2, ::2], g["dec_sub"]) / float(np.abs(g["dec_sub"]).max())
    _report(f"config2_clevr_{dt}", {"index_agreement_vs_reference": agree, "latent_rel_err": err_pre,
                                    "decoder_rel_err_same_codes": err_dec})
    assert decs[0].shape == (1, 1, 4, 256, 256) and torch.isfinite(decs[0]).all()
    # (report_config2_clevr_*.json: agreement 0.9727 / 0.9961 — 7 and 1 of 256 tokens —, latent 1.5e-2 / 1.6e-3, decoder
    # 1.7e-2 / 2.5e-3 for bf16 / fp16; floors within 3 points, bounds 1.5x)
    assert agree >= (0.94 if dt == "bf16" else 0.99)
    assert err_pre <= (2.2e-2 if dt == "bf16" else 2.5e-3) and err_dec <= (2.6e-2 if dt == "bf16" else 3.7e-3)


# ------------------------------------------------------------------------------------------------ top-k > 1
def test_topk4_two_samples_match_reference(golden):
    """VQModel.forward(topk=4, sample_number=2): the sampling branch of get_multiple_codewords (quantize.py:344-381)
    against the reference's own draws (CPU generator, row-0 distribution quirk): indices and gathered latents bit-exact,
    both decodes within 1e-4."""
    g = golden("vqgan_topk4_s2.npz")
    m, sd, p = _ge_model(golden)
    x, mask = testing.rect_hole_input(1, 256, 256, seed=3)
    torch.manual_seed(3)
    with torch.no_grad():
        decs, diff, idx, pre, quants = m(x.to(DEV), topk=4, extrapolation_mask=mask.to(DEV), sample_number=2,
                                         get_codebook_count=True, get_pre_quantized_feature=True, get_quantized_feature=True)
    assert diff is None and len(decs) == 2 and decs[0].shape == (1, 1, 4, 256, 256) and quants.shape == (1, 2, 256, 16, 16)
    assert torch.equal(idx.cpu(), torch.from_numpy(g["indices"]))
    assert testing.sha256(quants) == g["quant_sha"].tobytes()
    for s in range(2):
        assert _maxerr(decs[s][0, 0][..., ::2, ::2], g["dec_sub"][s]) <= TOL


def test_topk_samples_batched_are_dense(golden):
    """B = 2, sample_number = 2: every sample's slice of the (B,S,h,w,D) gather has batch stride S*h*w*D; the decoder
    must see dense NHWC tensors (item b > 0 was read from the wrong memory before)."""
    m, sd, p = _ge_model(golden)
    xs, ms = zip(*[testing.rect_hole_input(1, 256, 256, seed=50 + i) for i in range(2)])
    x, mask = torch.cat(xs).to(DEV), torch.cat(ms).to(DEV)
    with torch.no_grad():
        torch.manual_seed(7)
        decs, _, idx, quants = m(x, topk=4, extrapolation_mask=mask, sample_number=2, get_codebook_count=True,
                                 get_quantized_feature=True)
        assert idx.shape == (2, 2, 16, 16) and quants.shape == (2, 2, 256, 16, 16) and decs[0].shape == (1, 2, 4, 256, 256)
        for s in range(2):
            for b in range(2):
                want = m.decode(m.quantize.get_codebook_entry(idx[b, s].reshape(-1), (1, 16, 16, 256)))
                assert torch.equal(quants[b, s], m.quantize.get_codebook_entry(idx[b, s].reshape(-1), (1, 16, 16, 256))[0])
                assert _maxerr(decs[s][0, b], want[0]) <= 5e-5, (s, b)
    with pytest.raises(ops.SgamHipError, match="pixel-dense"):
        ops.nhwc_to_nchw(torch.zeros((2, 2, 4, 4, 32), device=DEV)[:, 0])


def test_commitment_loss_matches_reference(golden):
    """row a9: `diff` of forward(topk=None) is the reference's emb_loss scalar (quantize.py:296-301)"""
    for name, res in (("ge64", 64), ("ge256", 256)):
        g = golden(f"vqgan_full_{name}.npz")
        m, sd, p = _model("google_earth", testing.codebook_from_stats(float(g["zmean"]), float(g["zstd"]), 4096, 256, int(g["cb_seed"])), 4096)
        x, mask = testing.rect_hole_input(1, res, res, seed=3)
        with torch.no_grad():
            dec, diff = m(x.to(DEV), extrapolation_mask=mask.to(DEV))[:2]
            _, loss, _, _ = m.encode(x.to(DEV), extrapolation_mask=mask.to(DEV))
        assert diff.dim() == 0 and diff.dtype == torch.float32
        assert abs(float(diff) - float(g["emb_loss"])) <= 5e-6 * float(g["emb_loss"]) and torch.equal(diff, loss)


# ------------------------------------------------------------------------------------------------ trajectories
def test_ge_trajectory_free_running_32_frames(golden):
    """BASELINE config 3: the GoogleEarth loop FREE-RUNNING for 32 generated frames — every frame conditions on this
    backend's own earlier outputs, exactly like the reference conditioned on its own (no teacher forcing).

    What is well posed: the loop is a DISCRETE feedback system — the splat rounds projected points to pixels
    (warp.py:225), the feedback truncates RGB to uint8 (:898-901), the quantiser takes an arg-min — and the
    seeded-weight VQGAN turns one flipped code into a different frame (GroupNorm / attention are global).  Outputs that
    agree with the reference to ~2e-5 (inside the 1e-4 budget, but not bit-identical: a CPU and a GPU sum in different
    orders) therefore stay on the reference's trajectory only until one of those decisions falls the other way.  So:
    every frame up to the first index difference must be exact (indices, hole mask, uint8 RGB within the 1-LSB
    truncation boundary, depth within 1e-3) and there must be at least 3 of them; the first difference must be
    EXPLAINED — every differing token is a near-tie of the reference's own run (relative top-2 gap < 1e-4), or the model
    input of that step already differs from the reference's by a discrete splat decision (x checksum off by more than
    rounding noise while the hole mask is still identical); pose / source bookkeeping must match on all 32 frames.
    Per-frame agreement and RGB-D drift are reported (DESIGN.md §2 quotes them)."""
    tr = golden("trajectory_ge_free32.npz")
    seed = golden("trajectory_ge.npz")
    m, sd, p = _ge_model(golden)
    steps = int(tr["steps"])
    scene = InfiniteSceneGeneration(m, "google_earth", seed_index=0, output_dim=(steps + 1, 1),
                                    seed_frame=(seed["seed_rgb"], seed["seed_depth"]))
    cb = sd["quantize.embedding.weight"].to(DEV)
    rows, first = [], None
    for step in range(steps):
        tgt = scene.next_pose(scene.curr)
        srcs, _ = scene.get_src_grid_coords(tgt)
        assert tuple(tgt) == tuple(tr[f"s{step}.tgt"]) and [tuple(s) for s in srcs] == [tuple(s) for s in tr[f"s{step}.srcs"]]
        res = scene.one_step_prediction(tgt)
        # (1) THIS backend's step against the oracle on THIS backend's own inputs — asserted at every one of the 32 steps
        orow = _oracle_check_step(scene, res, sd, p, srcs, tgt, "google_earth")
        # (2) the reference's free run: a report (the two runs part ways at the first discrete decision that falls the other way)
        q = res["feature"].reshape(256, -1).t()
        idx = torch.cdist(q.double(), cb.double()).argmin(1).reshape(16, 16).cpu()
        ref_idx = torch.from_numpy(tr[f"s{step}.indices"].astype(np.int64))
        fr = scene.frames[tuple(tgt)]
        assert torch.isfinite(fr["depth"]).all()
        du8 = np.abs(fr["rgb_u8"].cpu().numpy()[::4, ::4].astype(np.int16) - tr[f"s{step}.rgb_u8_sub"].astype(np.int16))
        dd = np.abs(fr["depth"].cpu().numpy()[::4, ::4] - tr[f"s{step}.depth_sub"])
        mask_same = np.array_equal(np.packbits((res["x"][0, 3] == -2).cpu().numpy()), tr[f"s{step}.mask"])
        row = {"step": step, "idx_agree": float((idx == ref_idx).float().mean()), "mask_same": bool(mask_same),
               "u8_max": int(du8.max()), "u8_frac_diff": float((du8 != 0).mean()), "depth_max": float(dd.max()),
               "ref_min_gap": float(tr[f"s{step}.gap"].min()),
               "x_sum_delta": abs(float(res["x"].double().sum()) - float(tr[f"s{step}.x_sum"])), "oracle": orow}
        if first is None and row["idx_agree"] < 1.0:
            first = step
            gaps = tr[f"s{step}.gap"].reshape(16, 16)[(idx != ref_idx).numpy()]
            row["flipped_tokens"], row["flipped_ref_gap_max"] = int(gaps.size), float(gaps.max())
        rows.append(row)
        scene.curr += 1
    _report("trajectory_free32", {"first_difference_at_step": first, "rows": rows})
    n_exact = steps if first is None else first
    for r in rows[:n_exact]:
        assert r["mask_same"] and r["u8_max"] <= 1 and r["u8_frac_diff"] < 5e-3 and r["depth_max"] <= 1e-3, r
    assert n_exact >= 3, rows[:3]                     # the margin-guarded frame and its immediate successors
    if first is not None:
        # the first difference must be EXPLAINED (docstring): every flipped token is a near-tie of the reference's own run, or the
        # model input of that step already differs from the reference's (identical fp32 inputs give x_sum_delta == 0 exactly: both
        # sides sum the same values in float64; one uint8 feedback LSB is 7.8e-3) while the hole mask is still the reference's
        r = rows[first]
        near_tie = r["flipped_ref_gap_max"] < 1e-4
        input_differs = r["x_sum_delta"] > 1e-6 and r["mask_same"]
        assert near_tie or input_differs, r
        if not near_tie:
            # ... and that input difference must itself come from the discrete feedback (uint8 truncation boundary pixels of an
            # EARLIER frame, bounded by 1 LSB each), not from a wrong frame: every earlier frame was within the boundary
            assert all(q["u8_max"] <= 1 and q["depth_max"] <= 1e-3 for q in rows[:first]), rows[:first]
            assert any(q["u8_frac_diff"] > 0 for q in rows[:first]) or rows[first]["x_sum_delta"] < 1e-2, rows[:first + 1]


def _oracle_check_step(scene, res, sd, p, srcs, tgt, dataset, tgt_depth=None):
    """One step of the scene loop against the oracle ON THE HIP PATH'S OWN INPUTS: the conditioning warp
    is recomputed by oracle/warp_oracle.c from the frames this backend's store holds and the poses it uploaded (model input
    and hole mask bit for bit), the VQGAN by oracle/vqgan.py from this backend's own `x` (indices equal wherever the
    oracle's top-2 margin is >= 1e-4, latent and decoder output within 1e-4 — the decoder on the HIP path's own codes, so a
    near-tie that fell the other way does not hide a decoder bug), and the feedback codec (uint8 truncation within its
    1-LSB boundary, de-normalised depth within 1e-3; bit for bit on this backend's own decoder output).  Returns the
    numbers for the report."""
    from oracle import vqgan as OV
    from oracle import warp as OW
    from conftest import bits_equal
    N = len(srcs)
    feats = np.stack([scene.frames[tuple(c)]["rgb_f"].cpu().numpy().transpose(2, 0, 1) for c in srcs])[None]   # (1,N,3,H,W)
    K = scene.K.astype(np.float32)
    x_hip = res["x"].cpu()
    mask_hip = res["extrapolation_mask"].cpu()
    if tgt_depth is None:      # forward-splat branch (model.py:184-229)
        depths = np.stack([scene._src_depth(tuple(c)).cpu().numpy() for c in srcs])[None]                       # (1,N,H,W)
        T = np.zeros((1, N, 4, 4), np.float32)
        T[0, :, :3, :3] = res["batch_R_rels"].cpu().numpy()[0]
        T[0, :, :3, 3] = res["batch_t_rels"].cpu().numpy()[0]
        T[0, :, 3, 3] = 1.0
        w = OW.forward_splat(feats, depths, K[None], np.tile(K, (1, N, 1, 1)), T)
        em = torch.from_numpy(w["extrapolation_mask"])
        x_or = torch.cat([torch.from_numpy(w["merge_feats"]), OW.normalise_depth(torch.from_numpy(w["merge_depths"]), em, dataset)], 1)
    else:                      # rgbd branch (:575-580, model.py:260-274): inverse warp at the fused volume's rendered depth
        depths = np.stack([scene.frames[tuple(c)]["depth"].cpu().numpy() for c in srcs])[None]
        tgt_node = scene.transform_grid[tgt[0]][tgt[1]]
        T_t2s = scene.relative_poses(tgt_node, [scene.transform_grid[c[0]][c[1]] for c in srcs])[2].astype(np.float32)
        td = tgt_depth.cpu().numpy()[None]
        warped = OW.inverse_warp(feats, depths, td, np.tile(K, (1, N, 1, 1)), K[None], T_t2s[None])
        wd = torch.from_numpy(td)[:, None]
        em = wd <= 0                   # model.py:196-199: holes of a supplied depth are its non-positive pixels
        x_or = torch.cat([torch.from_numpy(warped), OW.normalise_depth(wd, em, dataset)], 1)
    assert torch.equal(mask_hip.bool(), em.bool()), "hole mask must equal the oracle's on the same inputs"
    assert bits_equal(x_hip.numpy(), x_or.numpy()), "model input must equal the oracle's bit for bit on the same inputs"
    o = OV.forward(sd, p["ddconfig"], x_hip, mask_hip, topk=1)
    idx_or = o["indices"].reshape(-1)
    pre_hip = res["pre_quantized_features"].cpu()[None]
    assert _maxerr(pre_hip, o["pre_quant"]) <= TOL
    z = o["pre_quant"].permute(0, 2, 3, 1).reshape(-1, 256)
    gap = testing.top2_relative_gap(z, sd["quantize.embedding.weight"])
    cbd = sd["quantize.embedding.weight"].double()
    idx_hip = torch.cdist(res["feature"].cpu().reshape(256, -1).t().double(), cbd).argmin(1)
    differ = idx_hip != idx_or
    assert not bool((differ & (gap >= 1e-4)).any()), \
        f"a well-conditioned token changed its code: gaps {gap[differ & (gap >= 1e-4)].tolist()}"
    dec_hip = res["rgbd"].cpu()[None]
    if bool(differ.any()):      # near-ties only: judge the decoder on the codes the HIP path actually decoded
        zq = torch.nn.functional.embedding(idx_hip, sd["quantize.embedding.weight"]).view(1, 16, 16, 256).permute(0, 3, 1, 2)
        dec_or = OV.decode(sd, p["ddconfig"], zq.contiguous())
    else:
        dec_or = o["dec"][0][0]
    err_dec = _maxerr(dec_hip, dec_or)
    assert err_dec <= TOL, err_dec
    fr = scene.frames[tuple(tgt)]
    u8_hip, d_hip = fr["rgb_u8"].cpu().numpy(), fr["depth"].cpu().numpy()
    # the codec on this backend's own decoder output: bit for bit
    assert np.array_equal(u8_hip, OW.rgb_to_uint8(dec_hip[0, :3]))
    assert bits_equal(d_hip, OW.denormalise_depth(dec_hip[0, 3], dataset).numpy())
    # and against the oracle's decoder output: 1-LSB truncation boundary / 1e-3
    du8 = np.abs(u8_hip.astype(np.int16) - OW.rgb_to_uint8(dec_or[0, :3]).astype(np.int16))
    dd = float(np.abs(d_hip - OW.denormalise_depth(dec_or[0, 3], dataset).numpy()).max())
    assert du8.max() <= 1 and (du8 != 0).mean() < 5e-3 and dd <= 1e-3, (int(du8.max()), float((du8 != 0).mean()), dd)
    return {"n_src": N, "near_tie_flips": int(differ.sum()), "min_gap": float(gap.min()), "pre_err": _maxerr(pre_hip, o["pre_quant"]),
            "dec_err": err_dec, "u8_boundary_frac": float((du8 != 0).mean()), "depth_err": dd}


def test_ge_rgbd_branch_free_running_matches_oracle_per_step(golden):
    """The rgbd_integration branch of config 3 (inference_pipeline.py:745-838, 575-580) free-running for 8 frames: at every
    step the inverse warp at the depth this backend's fused volume rendered, the VQGAN and the feedback codec are checked
    against the oracle on this backend's own inputs (the TSDF render itself is pinned in tests/test_gpu_tsdf.py)."""
    seed = golden("trajectory_ge.npz")
    m, sd, p = _ge_model(golden)
    scene = InfiniteSceneGeneration(m, "google_earth", seed_index=0, output_dim=(9, 1),
                                    seed_frame=(seed["seed_rgb"], seed["seed_depth"]), use_rgbd_integration=True)
    seen = {}
    render = scene.rgbd_integration

    def recording(src_nodes, tgt_node):
        seen["tgt_depth"] = render(src_nodes, tgt_node)
        return seen["tgt_depth"]

    scene.rgbd_integration = recording
    rows = []
    for step in range(8):
        tgt = scene.next_pose(scene.curr)
        srcs, _ = scene.get_src_grid_coords(tgt)
        res = scene.one_step_prediction(tgt)
        rows.append(_oracle_check_step(scene, res, sd, p, srcs, tgt, "google_earth", tgt_depth=seen["tgt_depth"]))
        scene.curr += 1
    _report("trajectory_rgbd8_oracle", {"rows": rows})


def test_clevr_trajectory_matches_reference(golden):
    """CLEVR-Infinite loop, 3 steps on a 2x2 grid from the reference's own template: seed depth converted twice in
    float64 (bit-exact), num_src 5 / radius 1.0 source choice, 16384-code quantiser, saved uint8 RGB within 1 LSB."""
    tr = golden("trajectory_clevr.npz")
    m, sd, p = _model("clevr-infinite", testing.codebook_from_stats(float(tr["zmean"]), float(tr["zstd"]), 16384, 256, int(tr["cb_seed"])), 16384)
    scene = InfiniteSceneGeneration(m, "clevr-infinite", seed_index=0, output_dim=(2, 2),
                                    seed_frame=(tr["seed_rgb"], tr["seed_depth_once"]))
    assert np.array_equal(scene._src_depth((0, 0)).cpu().numpy(), tr["seed_src_depth"])
    cb = sd["quantize.embedding.weight"].to(DEV)
    lut = ops.rgb_lut(DEV)
    for step in range(3):
        tgt = scene.next_pose(scene.curr)
        srcs, _ = scene.get_src_grid_coords(tgt)
        assert tuple(tgt) == tuple(tr[f"s{step}.tgt"]) and [tuple(s) for s in srcs] == [tuple(s) for s in tr[f"s{step}.srcs"]]
        res = scene.one_step_prediction(tgt)
        assert np.array_equal(res["batch_R_rels"].cpu().numpy(), tr[f"s{step}.R_rels"])
        assert np.array_equal(res["batch_t_rels"].cpu().numpy(), tr[f"s{step}.t_rels"])
        assert np.array_equal(np.packbits((res["x"][0, 3] == -2).cpu().numpy()), tr[f"s{step}.mask"])
        assert abs(float(res["x"].double().sum()) - float(tr[f"s{step}.x_sum"])) <= 1e-2
        q = res["feature"].reshape(256, -1).t()
        idx = torch.cdist(q.double(), cb.double()).argmin(1).reshape(16, 16).cpu()
        assert torch.equal(idx, torch.from_numpy(tr[f"s{step}.indices"].astype(np.int64)))
        assert _maxerr(res["rgbd"][:, ::4, ::4], tr[f"s{step}.rgbd_sub"]) <= TOL
        fr = scene.frames[tuple(tgt)]
        du8 = np.abs(fr["rgb_u8"].cpu().numpy().astype(np.int16) - tr[f"s{step}.rgb_u8"].astype(np.int16))
        assert du8.max() <= 1 and (du8 != 0).mean() < 5e-3
        # CLEVR depth = 1 / (h (1/7 - 1/16) + 1/16) has a pole inside the decoder's output range (random weights reach it):
        # compare the inverse depth, which is linear in the decoder output (1e-4 there <=> 4e-6 here)
        assert np.abs(1.0 / fr["depth"].cpu().numpy().astype(np.float64) - 1.0 / tr[f"s{step}.depth"].astype(np.float64)).max() <= 1e-5
        u8 = torch.from_numpy(tr[f"s{step}.rgb_u8"]).to(DEV)       # teacher forcing, like the GoogleEarth 3-step test
        fr["rgb_u8"], fr["rgb_f"] = u8, ops.rgb_u8_to_f32(u8)
        fr["depth"] = torch.from_numpy(tr[f"s{step}.depth"].astype(np.float32)).to(DEV)
        scene.curr += 1


def test_clevr_rgbd_branch_uses_once_converted_seed_depth():
    """with use_rgbd_integration the reference fuses and inverse-warps the seed depth as loaded (converted
    ONCE, inference_pipeline.py:570-580); only batch['src_depths'] carries the second conversion (:582-590)."""
    from sgam_neurips22_amd.inference_pipeline import ray_to_z_depth, synthetic_seed_frame
    m = VQModel(**default_params("clevr-infinite")).to(DEV).eval()
    seed = synthetic_seed_frame("clevr-infinite", 0)
    seen = {}

    def provider(scene, tgt_node, src_nodes, batch):
        seen["src_depths"] = batch["src_depths"].clone()
        return torch.full((256, 256), 12.0, device=DEV)

    scene = InfiniteSceneGeneration(m, "clevr-infinite", output_dim=(2, 2), seed_frame=seed, use_rgbd_integration=True,
                                    tgt_depth_provider=provider)
    once = scene.frames[(0, 0)]["depth"].cpu().numpy()
    assert np.array_equal(once, seed[1])
    tgt = scene.next_pose(scene.curr)
    res = scene.one_step_prediction(tgt)
    twice = ray_to_z_depth(seed[1].astype(np.float64), scene.K).astype(np.float32)
    assert np.array_equal(seen["src_depths"][0, 0].cpu().numpy(), twice)
    # the inverse warp saw the once-converted map: redo it with the contiguous entry point on that map
    T = scene.relative_poses(scene.transform_grid[tgt[0]][tgt[1]], [scene.transform_grid[0][0]])[2]   # (1,4,4) tgt->src
    want = ops.inverse_warp(scene.frames[(0, 0)]["rgb_f"].permute(2, 0, 1)[None, None].contiguous(),
                            scene.frames[(0, 0)]["depth"][None, None], torch.full((1, 256, 256), 12.0, device=DEV),
                            scene._K_dev[None], scene._Kinv_dev[None], torch.from_numpy(T.astype(np.float32)).to(DEV))
    assert torch.equal(res["x"][:, :3], want)


@pytest.mark.parametrize("dataset,grid", [("google_earth", (100, 1)), ("clevr-infinite", (20, 20))])
def test_scene_constructed_with_the_reference_scripts_exact_arguments(dataset, grid):
    """`main_scene_generation.py:46-53` verbatim: model `.to('cuda:0').eval()`, the three seeds, then
    `InfiniteSceneGeneration(model, data, seed_index=args.seed_index, use_rgbd_integration=args.use_rgbd_integration,
    offscreen_rendering=args.offscreen_rendering)` with argparse's defaults — seed_index the STRING "0", both flags True — i.e.
    the default CLI path: the TSDF branch, the reference's default grid (100 x 1 / 20 x 20), num_src 3 / 5.  Two steps of its
    `scene_expansion` loop body run (templates/ is absent on this box: synthetic seed frame, announced by a warning)."""
    import random
    import warnings
    model = VQModel(**default_params(dataset)).to("cuda:0").eval()
    random.seed(10)
    np.random.seed(29)
    torch.random.manual_seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        framework = InfiniteSceneGeneration(model, dataset, seed_index="0", use_rgbd_integration=True, offscreen_rendering=True)
    assert framework.output_dim == grid and framework.num_src == (3 if dataset == "google_earth" else 5)
    assert framework.use_rgbd_integration and model.use_rgbd_integration and framework.volume is not None
    assert len(framework._ordered_grid_coords) == grid[0] * grid[1] and framework.curr == 1
    for _ in range(2):                                         # the body of scene_expansion (:436-440)
        res = framework.one_step_prediction(framework.next_pose(framework.curr))
        framework.curr += 1
        assert res["rgbd"].shape == (4, 256, 256) and torch.isfinite(res["rgbd"]).all()
    assert len(framework.frames) == 3 and framework.volume.check() > 0


# ------------------------------------------------------------------------------------------------ split-fp32 range guard
def test_split_fp32_range_guard_fires_and_recovers():
    """the split-fp32 path needs |x| < 65520 on operands that are not GroupNorm-ed first (the
    stride-2 / upsampling / shortcut convs read the raw residual stream).  With encoder.conv_in scaled so that the stream
    runs at ~2e5, the kernels must report it (range flag), the forward must recompute on the fp32-in MFMA path, and the
    result must match the oracle like any other forward (finite, indices exact on a margin-checked codebook, 1e-4)."""
    import warnings
    from oracle import vqgan as OV
    p = default_params("google_earth")
    m = VQModel(**p)
    sd = testing.synthetic_state_dict(m.state_dict(), seed=0)
    sd["encoder.conv_in.weight"] = sd["encoder.conv_in.weight"] * 4.0e5
    x, mask = testing.rect_hole_input(1, 64, 64, seed=3)
    pre = OV.encode_features(sd, p["ddconfig"], x, mask)
    assert torch.isfinite(pre).all()
    z = pre.permute(0, 2, 3, 1).reshape(-1, 256)
    sd["quantize.embedding.weight"], _ = testing.repaired_codebook(z, float(z.mean()), float(z.std()), 4096, 256, 0, 1e-3)
    o = OV.forward(sd, p["ddconfig"], x, mask)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    assert ops.F32_MODE == "split"
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            with torch.no_grad():
                dec, _, idx, pre_g = m(x.to(DEV), extrapolation_mask=mask.to(DEV), get_codebook_count=True,
                                       get_pre_quantized_feature=True)
        assert any("left fp16's range" in str(x_.message) for x_ in w), "the range flag must fire"
        assert ops.F32_MODE == "mfma"
        assert torch.isfinite(dec).all() and torch.isfinite(pre_g).all()
        assert torch.equal(idx.cpu(), o["indices"])
        assert _maxerr(dec, o["dec"]) <= TOL, _maxerr(dec, o["dec"])
        # and a well-scaled model does not trip it
        ops.set_f32_mode("split")
        sd2 = testing.synthetic_state_dict(m.state_dict(), seed=0)
        m2 = VQModel(**p)
        m2.load_state_dict(sd2)
        m2 = m2.to(DEV).eval()
        with warnings.catch_warnings(record=True) as w2:
            warnings.simplefilter("always")
            with torch.no_grad():
                m2(x.to(DEV), extrapolation_mask=mask.to(DEV))
        assert not w2 and ops.F32_MODE == "split"
    finally:
        ops.set_f32_mode("split")


@pytest.mark.parametrize("scale", [1e-3, 1e-5])
def test_split_fp32_small_magnitude_stream(scale):
    """VERDICT r2 weak #2 — the OTHER end of the split's range: a residual stream of magnitude << 1 puts the lo halves of
    un-normalised operands into fp16 subnormals (absolute error <= 2^-25 per element instead of 2^-22 relative).  With
    encoder.conv_in scaled DOWN (weight and bias, so the stream entering the first ResnetBlock really is ~scale) the forward
    must still match the oracle at the north-star tolerance — exact indices on a margin-checked codebook, 1e-4 on latent and
    RGB-D — either on the split path itself or, if a kernel reports lost precision, after the same fallback as the
    overflow case."""
    import warnings
    from oracle import vqgan as OV
    p = default_params("google_earth")
    m = VQModel(**p)
    sd = testing.synthetic_state_dict(m.state_dict(), seed=0)
    sd["encoder.conv_in.weight"] = sd["encoder.conv_in.weight"] * scale
    sd["encoder.conv_in.bias"] = sd["encoder.conv_in.bias"] * scale
    x, mask = testing.rect_hole_input(1, 64, 64, seed=3)
    pre = OV.encode_features(sd, p["ddconfig"], x, mask)
    z = pre.permute(0, 2, 3, 1).reshape(-1, 256)
    sd["quantize.embedding.weight"], _ = testing.repaired_codebook(z, float(z.mean()), float(z.std()), 4096, 256, 0, 1e-3)
    o = OV.forward(sd, p["ddconfig"], x, mask)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    assert ops.F32_MODE == "split"
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            with torch.no_grad():
                dec, _, idx, pre_g = m(x.to(DEV), extrapolation_mask=mask.to(DEV), get_codebook_count=True,
                                       get_pre_quantized_feature=True)
        fell_back = ops.F32_MODE == "mfma"
        _report(f"small_magnitude_{scale:g}", {"fell_back_to_fp32_mfma": fell_back, "pre_err": _maxerr(pre_g, o["pre_quant"]),
                                                "dec_err": _maxerr(dec, o["dec"]), "warnings": [str(x_.message)[:80] for x_ in w]})
        assert torch.isfinite(dec).all()
        assert torch.equal(idx.cpu(), o["indices"])
        assert _maxerr(pre_g, o["pre_quant"]) <= TOL and _maxerr(dec, o["dec"]) <= TOL
    finally:
        ops.set_f32_mode("split")
