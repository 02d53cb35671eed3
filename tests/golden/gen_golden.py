"""Golden-vector generator — BUILD CONTAINER ONLY (needs /root/reference; never runs on the GPU box).

Imports the reference (yshen47/SGAM_NeurIPS22) with the sys.modules stubs of _ref_import.py, runs ITS code on
seeded inputs and writes small .npz fixtures next to this file.  The fixtures are data (inputs + the
reference's outputs); no reference source is copied.  Re-run:  python tests/golden/gen_golden.py

Fixtures
  splat_*.npz      render_projection_from_srcs_fast (warp.py:193-286), torch.use_deterministic_algorithms(True)
  invwarp_*.npz    InfiniteSceneGeneration.inverse_warping (inference_pipeline.py:662-743)
  vqgan_ops.npz    ResnetBlock / AttnBlock / Downsample / Upsample / GroupNorm+swish / VectorQuantizer2
  vqgan_full_*.npz VQModel.forward on seeded synthetic weights + margin-guarded codebook (clevr256_argmin = BASELINE config 1
                   verbatim: CLEVR, U(-1, 1) input, mask all false, topk=None, get_codebook_count=True)
  trajectory_ge.npz 3 steps of InfiniteSceneGeneration.one_step_prediction (GoogleEarth seed0)
  vqgan_topk4_s2.npz        VQModel.forward(topk=4, sample_number=2): get_multiple_codewords' sampling branch (CPU RNG)
  config5_ge512_b4.npz      BASELINE config 5: 512x512, four warp candidates (two real template sources) -> get_x -> forward
  trajectory_ge_free32.npz  32 FREE-RUNNING steps of the GoogleEarth loop (no teacher forcing; BASELINE config 3)
  trajectory_clevr.npz      3 steps of the CLEVR-Infinite loop (seed depth converted twice in float64, num_src 5)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
os.environ.setdefault("MPLBACKEND", "Agg")

import _ref_import as R  # noqa: E402

from sgam_neurips22_amd import testing  # noqa: E402  (product-side seeded weight/input helpers)

R.install()
torch.Tensor.cuda = lambda self, *a, **k: self  # the reference hard-codes .cuda() (inference_pipeline.py:608)
torch.use_deterministic_algorithms(True)

from sgam.generative_sensing_module.model import VQModel  # noqa: E402  (REFERENCE)
from sgam.generative_sensing_module.modules.diffusionmodules import model as ref_dm  # noqa: E402
from sgam.inference_pipeline import InfiniteSceneGeneration  # noqa: E402
from sgam.point_rendering import warp as ref_warp  # noqa: E402


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


def gen_splat():
    print("forward splat")
    for tag, seed, B, N, H, W, rs_, dr, bad in testing.SPLAT_CASES:
        f, d, Ks, T = testing.synth_warp_inputs(seed, B, N, H, W, rs_, bad)
        r = ref_warp.render_projection_from_srcs_fast(torch.from_numpy(f), torch.from_numpy(d), torch.from_numpy(Ks[:, 0]),
                                                      torch.from_numpy(Ks), torch.from_numpy(T), src_num=N, depth_range=dr,
                                                      parallel=True)
        save(f"splat_{tag}.npz", merge_depths=r[0], merge_feats=r[1],
             extrapolation_mask=r[2], mask=np.packbits(r[3].numpy()), idx=r[5].numpy().astype(np.int32),
             projected_features=r[6])
    # GoogleEarth seed0 template, grid step (0,0)->(1,0): the real first warp of the pipeline
    from PIL import Image
    import torch.nn.functional as F
    rgb = np.array(Image.open(f"{R.REF}/templates/google_earth/seed0/im_00000.png").resize((256, 256), resample=Image.LANCZOS))
    dm = F.interpolate(torch.from_numpy(np.load(f"{R.REF}/templates/google_earth/seed0/dm_00000.npy")[None, None]),
                       size=(256, 256))[0][0].numpy().squeeze().astype(np.float32)
    return rgb, dm


def gen_invwarp():
    print("inverse warp")
    for tag, seed, N, H, W, s_, bad in testing.INVWARP_CASES:
        im, d, td, Ks, K, T = testing.synth_invwarp_inputs(seed, N, H, W, s_, bad)
        self = InfiniteSceneGeneration.__new__(InfiniteSceneGeneration)
        out = InfiniteSceneGeneration.inverse_warping(self, torch.from_numpy(im), torch.from_numpy(d), torch.from_numpy(td),
                                                      torch.from_numpy(Ks), torch.from_numpy(K)[None], torch.from_numpy(T))
        save(f"invwarp_{tag}.npz", warped=out)


def gen_ops():
    print("per-op VQGAN fixtures")
    out = {}
    for tag, kind, kw, shape in testing.OP_CASES:
        if kind == "ResnetBlock":
            mod, extra = ref_dm.ResnetBlock(temb_channels=0, dropout=0.0, **kw), (None,)
        elif kind == "AttnBlock":
            mod, extra = ref_dm.AttnBlock(kw["in_channels"]), ()
        else:
            mod, extra = getattr(ref_dm, kind)(kw["in_channels"], kw["with_conv"]), ()
        mod.load_state_dict(testing.synthetic_state_dict(mod.state_dict(), seed=5))
        with torch.no_grad():
            out[f"{tag}.y"] = mod(testing.seeded_tensor(tag, shape), *extra)
    gn = ref_dm.Normalize(256)
    gn.load_state_dict(testing.synthetic_state_dict(gn.state_dict(), seed=5))
    x = testing.seeded_tensor("gn256", (2, 256, 12, 12), 3.0, 0.5)
    with torch.no_grad():
        out["gn256.y"], out["gn256.y_swish"] = gn(x), ref_dm.nonlinearity(gn(x))
    save("vqgan_ops.npz", **out)


def gen_full(dataset, res, tag, topk=None, plain_input=False):
    print(f"full model {dataset} {res}x{res} topk={topk}")
    p = R.load_params(dataset)
    torch.manual_seed(0)
    model = VQModel(**p).eval()
    sd = testing.synthetic_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd)
    x, mask = testing.config1_input(res) if plain_input else testing.rect_hole_input(1, res, res, seed=3)
    with torch.no_grad():
        pre = model.encode(x, extrapolation_mask=mask)[3]
    z = pre.permute(0, 2, 3, 1).reshape(-1, pre.shape[1])
    zmean, zstd = float(z.mean()), float(z.std())
    cb_seed = 0
    while True:
        cb, gap = testing.codebook_from_stats(zmean, zstd, p["n_embed"], z.shape[1], cb_seed), None
        gap = testing.top2_relative_gap(z, cb)
        if float(gap.min()) >= 1e-4:
            break
        cb_seed += 1
    print(f"  codebook seed {cb_seed}: min relative top-2 gap {float(gap.min()):.3e}")
    sd["quantize.embedding.weight"] = cb
    model.load_state_dict(sd)
    torch.manual_seed(3)
    with torch.no_grad():
        if topk is None:
            dec, diff, idx, pre, quant = model(x, extrapolation_mask=mask, get_codebook_count=True,
                                               get_pre_quantized_feature=True, get_quantized_feature=True)
        else:
            decs, diff, idx, pre, quant = model(x, topk=topk, extrapolation_mask=mask, sample_number=1,
                                                get_codebook_count=True, get_pre_quantized_feature=True,
                                                get_quantized_feature=True)
            dec = decs[0][0]
    wsum = np.array([float(sd[k].double().abs().sum()) for k in sorted(sd.keys())[:8]])
    step = 1 if res <= 64 else 2
    extra = {} if diff is None else {"emb_loss": float(diff)}
    save(f"vqgan_full_{tag}.npz", dataset=dataset, res=res, zmean=zmean, zstd=zstd, cb_seed=cb_seed, **extra,
         weight_abs_sums=wsum, dec_sub=dec[..., ::step, ::step], dec_step=step, dec_sum=float(dec.double().sum()),
         dec_abs_sum=float(dec.double().abs().sum()), indices=idx, pre_quant=pre, quant=quant,
         topk=-1 if topk is None else topk, plain_input=int(plain_input))


def gen_trajectory(rgb0, dm0):
    print("GoogleEarth 3-step trajectory")
    import tempfile
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.symlink(f"{R.REF}/templates", os.path.join(tmp, "templates"))
    os.chdir(tmp)
    try:
        p = R.load_params("google_earth")
        torch.manual_seed(0)
        model = VQModel(**p).eval()
        sd = testing.synthetic_state_dict(model.state_dict(), seed=0)
        full = np.load(os.path.join(HERE, "vqgan_full_ge256.npz"))
        sd["quantize.embedding.weight"] = testing.codebook_from_stats(float(full["zmean"]), float(full["zstd"]), 4096, 256,
                                                                      int(full["cb_seed"]))
        model.load_state_dict(sd)
        import random
        random.seed(10); np.random.seed(29); torch.random.manual_seed(3)
        fw = InfiniteSceneGeneration(model, "google_earth", seed_index=0, use_rgbd_integration=False, output_dim=(4, 1))
        out = {"seed_rgb": rgb0, "seed_depth": dm0, "zmean": full["zmean"], "zstd": full["zstd"], "cb_seed": full["cb_seed"]}
        for step in range(3):
            with torch.no_grad():
                tgt = fw.next_pose(fw.curr)
                srcs, _ = fw.get_src_grid_coords(tgt)
                res = fw.one_step_prediction(tgt)
            if step == 0:  # the pipeline's first real warp as a stand-alone splat fixture (G3)
                T = torch.eye(4)[None, None].repeat(1, 1, 1, 1)
                T[0, 0, :3, :3], T[0, 0, :3, 3] = res["batch_R_rels"][0, 0], res["batch_t_rels"][0, 0]
                K = torch.from_numpy(fw.K.astype(np.float32))[None]
                sf = res["batch_src_imgs"].permute(0, 1, 4, 2, 3).contiguous()
                r = ref_warp.render_projection_from_srcs_fast(sf, res["batch_src_depths"][..., 0], K, K[None], T, src_num=1,
                                                              parallel=True)
                mf = r[1].numpy()
                save("splat_ge_seed0.npz", K=K[0], T=T[0, 0], merge_depths=r[0],
                     merge_feats_u8=np.round((mf + 1) * 127.5).astype(np.uint8), merge_feats_zero=np.packbits(mf == 0),
                     extrapolation_mask=np.packbits(r[2].numpy()), n_inbounds=int(r[3].sum()), idx_sum=int(r[5].sum()))
            from PIL import Image
            node = fw.transform_grid[tgt[0]][tgt[1]]
            out[f"s{step}.tgt"] = np.array(tgt)
            out[f"s{step}.srcs"] = np.array(srcs)
            out[f"s{step}.rgb_u8"] = np.array(Image.open(node["rgb_path"]))
            out[f"s{step}.depth"] = np.load(node["depth_path"])
            out[f"s{step}.R_rels"] = res["batch_R_rels"]
            out[f"s{step}.t_rels"] = res["batch_t_rels"]
            out[f"s{step}.mask"] = np.packbits((res["x"][0, 3] == -2).numpy())
            out[f"s{step}.x_sum"] = float(res["x"].double().sum())
            out[f"s{step}.rgbd_sub"] = res["rgbd"][:, ::4, ::4]
            out[f"s{step}.feature_idx_check"] = float(res["feature"].double().abs().sum())
            fw.curr += 1
        save("trajectory_ge.npz", **out)
    finally:
        os.chdir(cwd)


def _ge_model(cb=None):
    """GoogleEarth reference VQModel on the seeded synthetic weights (+ the ge256 fixture's margin-guarded codebook)."""
    p = R.load_params("google_earth")
    torch.manual_seed(0)
    model = VQModel(**p).eval()
    sd = testing.synthetic_state_dict(model.state_dict(), seed=0)
    full = np.load(os.path.join(HERE, "vqgan_full_ge256.npz"))
    sd["quantize.embedding.weight"] = cb if cb is not None else testing.codebook_from_stats(
        float(full["zmean"]), float(full["zstd"]), 4096, 256, int(full["cb_seed"]))
    model.load_state_dict(sd)
    return model, sd, p, full


def gen_topk4():
    """get_multiple_codewords' sampling branch (quantize.py:344-381) through VQModel.forward: topk=4, 2 samples.  The
    reference is run on the CPU, so the multinomial draws come from the CPU generator seeded right before the call."""
    print("top-k = 4, sample_number = 2 (GoogleEarth 256x256)")
    model, sd, p, full = _ge_model()
    x, mask = testing.rect_hole_input(1, 256, 256, seed=3)
    torch.manual_seed(3)
    with torch.no_grad():
        decs, diff, idx, pre, quants = model(x, topk=4, extrapolation_mask=mask, sample_number=2, get_codebook_count=True,
                                             get_pre_quantized_feature=True, get_quantized_feature=True)
    assert diff is None and len(decs) == 2 and idx.shape == (1, 2, 16, 16) and quants.shape == (1, 2, 256, 16, 16)
    n_sampled = int((idx[0, 0] != idx[0, 1]).sum())
    print(f"  tokens whose two samples differ: {n_sampled} / 256")
    save("vqgan_topk4_s2.npz", indices=idx, quant_sha=np.frombuffer(testing.sha256(quants), np.uint8),
         dec_sub=torch.stack([d[0, 0][..., ::2, ::2] for d in decs]), dec_sum=[float(d.double().sum()) for d in decs],
         n_sampled=n_sampled)


def gen_config5():
    """BASELINE config 5: GoogleEarth 512x512, a batch of four warp candidates per step.  Sources (N = 2): the reference's
    seed0 template at its native 512x512 and a synthetic seeded frame; candidates: four target poses in the pipeline's
    own grid geometry (testing.config5_batch).  Reference get_x (forward splat at B = 4) -> forward (arg-min path) item by item
    (B = 1 calls: the 16384 x 16384 fp32 score matrix is 1 GiB per item)."""
    print("config 5: 512x512, four warp candidates")
    from PIL import Image
    d = f"{R.REF}/templates/google_earth/seed0"
    src0_rgb = np.array(Image.open(f"{d}/im_00000.png").convert("RGB"))
    src0_depth = np.load(f"{d}/dm_00000.npy").astype(np.float32)
    batch_np = testing.config5_batch(src0_rgb, src0_depth)
    model, sd, p, full = _ge_model()
    batch = {k: torch.from_numpy(v) for k, v in batch_np.items()}
    batch["src_depths"] = batch["src_depths"][..., None]
    with torch.no_grad():
        x, x_dst, mask, wd = model.get_x(batch, "google_earth", return_extrapolation_mask=True, no_depth_range=True,
                                         parallel=True)
        pres = torch.cat([model.encode(x[b:b + 1], extrapolation_mask=mask[b:b + 1])[3] for b in range(4)])
    z = pres.permute(0, 2, 3, 1).reshape(-1, 256)
    zmean, zstd = float(z.mean()), float(z.std())
    cb, repairs = testing.repaired_codebook(z, zmean, zstd, 4096, 256, seed=0, min_gap=1e-4)
    gap = float(testing.top2_relative_gap(z, cb).min())
    print(f"  codebook N({zmean:.4f}, {zstd:.4f}), {len(repairs)} repaired rows: min relative top-2 gap {gap:.3e}")
    assert gap >= 1e-4
    sd["quantize.embedding.weight"] = cb
    model.load_state_dict(sd)
    decs, idxs, losses = [], [], []
    with torch.no_grad():
        for b in range(4):
            dec, diff, idx, pre = model(x[b:b + 1], extrapolation_mask=mask[b:b + 1], get_codebook_count=True,
                                        get_pre_quantized_feature=True)
            assert torch.equal(pre, pres[b:b + 1])
            decs.append(dec); idxs.append(idx); losses.append(float(diff))
    dec, idx = torch.cat(decs), torch.cat(idxs)
    save("config5_ge512_b4.npz", src0_rgb=src0_rgb, src0_depth=src0_depth, zmean=zmean, zstd=zstd, repairs=np.array(repairs, np.int64).reshape(-1, 2), min_gap=gap,
         x_sha=np.frombuffer(testing.sha256(x), np.uint8), mask=np.packbits(mask.numpy()), x_sum=float(x.double().sum()),
         x_sub=x[..., ::8, ::8], indices=idx.to(torch.int16), pre_quant0=pres[0], pre_quant_sub=pres[:, ::8],
         dec_sub=dec[..., ::4, ::4], dec_sum=[float(d.double().sum()) for d in decs], emb_loss=losses)


def _quant_to_idx(quant, cb):
    """indices of a pure-gather quantised latent (256,16,16): exact row match against the codebook"""
    q = quant.reshape(256, -1).t()
    d = torch.cdist(q.double(), cb.double())
    idx = d.argmin(1)
    assert torch.equal(cb[idx], q)
    return idx.reshape(16, 16)


def _run_trajectory(dataset, model, cb, steps, output_dim):
    import random
    import tempfile
    from PIL import Image
    cwd, tmp = os.getcwd(), tempfile.mkdtemp()
    os.symlink(f"{R.REF}/templates", os.path.join(tmp, "templates"))
    os.chdir(tmp)
    out = {}
    try:
        random.seed(10); np.random.seed(29); torch.random.manual_seed(3)
        fw = InfiniteSceneGeneration(model, dataset, seed_index=0, use_rgbd_integration=False, output_dim=output_dim)
        for step in range(steps):
            with torch.no_grad():
                tgt = fw.next_pose(fw.curr)
                srcs, _ = fw.get_src_grid_coords(tgt)
                res = fw.one_step_prediction(tgt)
            node = fw.transform_grid[tgt[0]][tgt[1]]
            out[f"s{step}.tgt"], out[f"s{step}.srcs"] = np.array(tgt), np.array(srcs)
            out[f"s{step}.indices"] = _quant_to_idx(res["feature"], cb).to(torch.int16)
            out[f"s{step}.mask"] = np.packbits((res["x"][0, 3] == -2).numpy())
            out[f"s{step}.x_sum"] = float(res["x"].double().sum())
            # how well conditioned each token's arg-min was in the reference's own run: relative top-2 gap
            out[f"s{step}.gap"] = testing.top2_relative_gap(res["pre_quantized_features"].reshape(256, -1).t(), cb).float()
            yield step, fw, res, node, out, Image
            fw.curr += 1
    finally:
        os.chdir(cwd)


def gen_trajectory_free(steps=32):
    """BASELINE config 3: the GoogleEarth loop FREE-RUNNING for 32 generated frames (every frame conditions on the
    reference's own earlier outputs through its PNG / NPY round trip)."""
    print(f"GoogleEarth free-running trajectory, {steps} frames")
    model, sd, p, full = _ge_model()
    cb = sd["quantize.embedding.weight"]
    final = None
    for step, fw, res, node, out, Image in _run_trajectory("google_earth", model, cb, steps, (steps + 1, 1)):
        u8 = np.array(Image.open(node["rgb_path"]))
        out[f"s{step}.rgb_u8_sub"] = u8[::4, ::4]
        out[f"s{step}.rgb_u8_sum"] = int(u8.astype(np.int64).sum())
        out[f"s{step}.depth_sub"] = np.load(node["depth_path"])[::4, ::4]
        final = out
        print(f"  step {step}: srcs {out[f's{step}.srcs'].tolist()} holes {int((res['x'][0, 3] == -2).sum())}")
    save("trajectory_ge_free32.npz", steps=steps, **final)


def gen_train():
    """one autoencoder update of the REFERENCE on the small model of the training tests: VQModel.forward + VQLPIPSWithDiscriminator
    (optimizer_idx 0, perceptual_weight 0, global_step 0 < disc_start) + backward — loss terms and a few gradients"""
    from sgam.generative_sensing_module.modules.losses.vqperceptual import VQLPIPSWithDiscriminator
    print("training step (small model)")
    p = testing.small_train_params(R.load_params("google_earth"))
    p["phase"] = "codebook"
    torch.manual_seed(0)
    model = VQModel(**p).train()
    sd = testing.synthetic_state_dict(model.state_dict(), seed=11)
    model.load_state_dict(sd)
    x, mask, x_dst = testing.train_batch()
    with torch.no_grad():
        pre = model.encode(x, extrapolation_mask=mask)[3]
    z = pre.permute(0, 2, 3, 1).reshape(-1, pre.shape[1])
    zmean, zstd = float(z.mean()), float(z.std())
    cb_seed = 0
    while float(testing.top2_relative_gap(z, testing.codebook_from_stats(zmean, zstd, 64, 32, cb_seed)).min()) < 1e-4:
        cb_seed += 1
    sd["quantize.embedding.weight"] = testing.codebook_from_stats(zmean, zstd, 64, 32, cb_seed)
    model.load_state_dict(sd)
    loss_fn = VQLPIPSWithDiscriminator(disc_start=10 ** 9, codebook_weight=1.0, perceptual_weight=0.0, disc_in_channels=4,
                                       disc_weight=0.8, disc_num_layers=2, use_discriminative_loss=True)
    torch.manual_seed(1)
    xrec, qloss, idx, pre = model(x, extrapolation_mask=mask, get_codebook_count=True, get_pre_quantized_feature=True)
    aeloss, log = loss_fn(qloss, x_dst, xrec, 0, 0, last_layer=model.get_last_layer(), split="train", extrapolation_mask=mask)
    model.zero_grad()
    aeloss.backward()
    named = dict(model.named_parameters())
    keep = ["encoder.conv_in.weight", "decoder.conv_out.weight", "quantize.embedding.weight", "encoder.mid.attn_1.q.weight",
            "conv_in.weight", "encoder.down.0.downsample.conv.weight", "decoder.up.1.upsample.conv.bias", "encoder.norm_out.weight"]
    gn = {k: float(v.grad.double().norm()) for k, v in named.items() if v.grad is not None}
    save("train_step_small.npz", zmean=zmean, zstd=zstd, cb_seed=cb_seed, loss=float(aeloss), quant_loss=float(qloss),
         rec_loss=float(log["train/rec_loss"]), d_weight=float(log["train/d_weight"]), indices=idx[2] if isinstance(idx, tuple) else idx,
         grad_norm_names=np.array(sorted(gn)), grad_norms=np.array([gn[k] for k in sorted(gn)]),
         **{"grad." + k: named[k].grad for k in keep})


def gen_train_gan():
    """the reference's whole training_step AFTER disc_start (perceptual_weight 0) on the small model: autoencoder loss with the
    adaptive-weighted generator term, its gradients, then the discriminator's hinge loss and gradients, BatchNorm running
    statistics after the three discriminator forwards"""
    from sgam.generative_sensing_module.modules.losses.vqperceptual import VQLPIPSWithDiscriminator
    print("training step with the discriminator (small model)")
    g0 = np.load(os.path.join(HERE, "train_step_small.npz"))
    p = testing.small_train_params(R.load_params("google_earth"))
    p["phase"] = "codebook"
    torch.manual_seed(0)
    model = VQModel(**p).train()
    sd = testing.synthetic_state_dict(model.state_dict(), seed=11)
    sd["quantize.embedding.weight"] = testing.codebook_from_stats(float(g0["zmean"]), float(g0["zstd"]), 64, 32, int(g0["cb_seed"]))
    model.load_state_dict(sd)
    x, mask, x_dst = testing.train_batch()
    loss_fn = VQLPIPSWithDiscriminator(disc_start=0, codebook_weight=1.0, perceptual_weight=0.0, disc_in_channels=4, disc_weight=0.8,
                                       use_discriminative_loss=True).train()
    loss_fn.discriminator.load_state_dict(testing.synthetic_disc_state_dict(loss_fn.discriminator.state_dict(), seed=2))
    xrec, qloss, idx, pre = model(x, extrapolation_mask=mask, get_codebook_count=True, get_pre_quantized_feature=True)
    aeloss, log = loss_fn(qloss, x_dst, xrec, 0, 0, last_layer=model.get_last_layer(), split="train", extrapolation_mask=mask)
    model.zero_grad()
    aeloss.backward()
    named = dict(model.named_parameters())
    keep = ["encoder.conv_in.weight", "decoder.conv_out.weight", "quantize.embedding.weight", "decoder.up.1.upsample.conv.bias"]
    gn = {k: float(v.grad.double().norm()) for k, v in named.items() if v.grad is not None}
    ae_grads = {"grad." + k: named[k].grad.clone() for k in keep}
    discloss, dlog = loss_fn(qloss, x_dst, xrec, 1, 0, last_layer=model.get_last_layer(), split="train", extrapolation_mask=mask)
    loss_fn.discriminator.zero_grad()
    discloss.backward()
    dn = dict(loss_fn.discriminator.named_parameters())
    dgn = {k: float(v.grad.double().norm()) for k, v in dn.items()}
    dsd = loss_fn.discriminator.state_dict()
    save("train_step_gan_small.npz", loss=float(aeloss), d_weight=float(log["train/d_weight"]), g_loss=float(log["train/g_loss"]),
         rec_loss=float(log["train/rec_loss"]), quant_loss=float(qloss), disc_loss=float(discloss),
         logits_real=float(dlog["train/logits_real"]), logits_fake=float(dlog["train/logits_fake"]),
         grad_norm_names=np.array(sorted(gn)), grad_norms=np.array([gn[k] for k in sorted(gn)]),
         dgrad_norm_names=np.array(sorted(dgn)), dgrad_norms=np.array([dgn[k] for k in sorted(dgn)]),
         **ae_grads, **{"dgrad.main.0.weight": dn["main.0.weight"].grad, "dgrad.main.3.weight": dn["main.3.weight"].grad,
                        "dgrad.main.11.bias": dn["main.11.bias"].grad, "dgrad.main.6.bias": dn["main.6.bias"].grad},
         **{"bn." + k: v for k, v in dsd.items() if "running" in k or "num_batches" in k})


FULL_TRAIN_KEEP = ["encoder.conv_in.weight", "decoder.conv_out.weight", "conv_in.weight", "quant_conv.weight",
                   "encoder.mid.attn_1.q.weight", "encoder.down.0.block.0.conv1.weight", "decoder.up.4.block.0.norm1.weight",
                   "decoder.up.1.upsample.conv.bias", "encoder.down.2.block.0.nin_shortcut.weight"]


def gen_train_full():
    """the same autoencoder update of the REFERENCE at the REAL configuration: the 68 990 620-parameter GoogleEarth model
    (trained_models/google_earth/config.yaml), one 256 x 256 image with its hole mask — VQModel.forward + VQLPIPSWithDiscriminator
    (optimizer_idx 0, perceptual_weight 0, before disc_start) + backward: loss terms, indices, the gradient norm of EVERY
    parameter tensor and nine full gradient tensors (VERDICT r2 next #7a)"""
    from sgam.generative_sensing_module.modules.losses.vqperceptual import VQLPIPSWithDiscriminator
    print("training step (full 256x256 GoogleEarth model)")
    p = R.load_params("google_earth")
    p["phase"] = "codebook"
    torch.manual_seed(0)
    model = VQModel(**p).train()
    sd = testing.synthetic_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd)
    x, mask = testing.rect_hole_input(1, 256, 256, seed=9)
    x_dst = testing.seeded_tensor("train_full.dst", (1, 4, 256, 256), scale=0.5).clamp(-1, 1)
    with torch.no_grad():
        pre = model.encode(x, extrapolation_mask=mask)[3]
    z = pre.permute(0, 2, 3, 1).reshape(-1, pre.shape[1])
    zmean, zstd = float(z.mean()), float(z.std())
    cb, repairs = testing.repaired_codebook(z, zmean, zstd, 4096, 256, 0, 1e-4)
    sd["quantize.embedding.weight"] = cb
    model.load_state_dict(sd)
    loss_fn = VQLPIPSWithDiscriminator(disc_start=10 ** 9, codebook_weight=1.0, perceptual_weight=0.0, disc_in_channels=4,
                                       disc_weight=0.8, use_discriminative_loss=True)
    xrec, qloss, idx, pre = model(x, extrapolation_mask=mask, get_codebook_count=True, get_pre_quantized_feature=True)
    aeloss, log = loss_fn(qloss, x_dst, xrec, 0, 0, last_layer=model.get_last_layer(), split="train", extrapolation_mask=mask)
    model.zero_grad()
    aeloss.backward()
    named = dict(model.named_parameters())
    gn = {k: float(v.grad.double().norm()) for k, v in named.items() if v.grad is not None}
    save("train_step_full256.npz", zmean=zmean, zstd=zstd, repairs=np.asarray(repairs, dtype=np.int64).reshape(-1, 2),
         loss=float(aeloss), quant_loss=float(qloss), rec_loss=float(log["train/rec_loss"]),
         indices=(idx[2] if isinstance(idx, tuple) else idx).to(torch.int16), xrec_sub=xrec.detach()[..., ::4, ::4],
         grad_norm_names=np.array(sorted(gn)), grad_norms=np.array([gn[k] for k in sorted(gn)]),
         **{"grad." + k: named[k].grad for k in FULL_TRAIN_KEEP})


def gen_lpips():
    """the REFERENCE's LPIPS class (modules/losses/lpips.py) on 64 x 64 images, with its shipped `lin` weights, over a stand-in for
    `torchvision.models.vgg16`: torchvision is not installed and its ImageNet checkpoint cannot be fetched, so the trunk has the
    standard configuration "D" with SYNTHETIC weights (testing.synthetic_vgg_state_dict).  What this pins is the reference's
    slicing, scaling, normalisation, `lin` weighting and averaging — not the metric's pretrained values."""
    import importlib
    import types
    import torch.nn as nn
    print("LPIPS (reference class, stand-in VGG16 trunk)")

    def vgg16(pretrained=False, **kw):
        layers, cin = [], 3
        for v in [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        m = nn.Module()
        m.features = nn.Sequential(*layers)
        return m

    tv, tvm = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")
    tvm.vgg16 = vgg16
    tv.models = tvm
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tvm
    for m in ("requests", "tqdm"):
        if m not in sys.modules:
            try:
                importlib.import_module(m)
            except Exception:
                sys.modules[m] = types.ModuleType(m)
    name = "sgam.generative_sensing_module.modules.losses.lpips"
    stub = sys.modules.pop(name, None)
    cwd = os.getcwd()
    os.chdir(R.REF)                              # load_from_pretrained resolves the shipped lin weights relative to the cwd
    try:
        ref_lpips = importlib.import_module(name)
        torch.manual_seed(0)
        lp = ref_lpips.LPIPS().eval()
    finally:
        os.chdir(cwd)
        if stub is not None:
            sys.modules[name] = stub
    sd = testing.synthetic_vgg_state_dict(lp.state_dict(), seed=4)
    lp.load_state_dict(sd)
    a = testing.seeded_tensor("lpips.a", (2, 3, 64, 64), scale=0.5).clamp(-1, 1).requires_grad_(True)
    b = testing.seeded_tensor("lpips.b", (2, 3, 64, 64), scale=0.5).clamp(-1, 1)
    val = lp(a, b)
    val.sum().backward()
    save("lpips_small.npz", value=val.detach().reshape(-1), grad_input=a.grad,
         **{"lin." + k: v for k, v in sd.items() if k.startswith("lin")})


def gen_trajectory_clevr():
    """CLEVR-Infinite loop, 3 steps on a 2x2 grid: 16384 codes, num_src 5, the seed depth's ray->z conversion applied at
    construction (:71-79) AND again at every load (:582-590), both in float64."""
    print("CLEVR 3-step trajectory")
    p = R.load_params("clevr-infinite")
    torch.manual_seed(0)
    model = VQModel(**p).eval()
    sd = testing.synthetic_state_dict(model.state_dict(), seed=0)
    full = np.load(os.path.join(HERE, "vqgan_full_clevr256_topk1.npz"))
    cb = testing.codebook_from_stats(float(full["zmean"]), float(full["zstd"]), 16384, 256, int(full["cb_seed"]))
    sd["quantize.embedding.weight"] = cb
    model.load_state_dict(sd)
    final = None
    for step, fw, res, node, out, Image in _run_trajectory("clevr-infinite", model, cb, 3, (2, 2)):
        if step == 0:
            out["seed_rgb"] = np.array(Image.open(fw.transform_grid[0][0]["rgb_path"]).convert("RGB"))
            out["seed_depth_once"] = np.load(fw.transform_grid[0][0]["depth_path"])          # float64, converted once
            out["seed_src_depth"] = res["batch_src_depths"][0, 0, ..., 0]                     # fp32 after both conversions
        out[f"s{step}.rgb_u8"] = np.array(Image.open(node["rgb_path"]))
        out[f"s{step}.depth"] = np.load(node["depth_path"])
        out[f"s{step}.R_rels"], out[f"s{step}.t_rels"] = res["batch_R_rels"], res["batch_t_rels"]
        out[f"s{step}.rgbd_sub"] = res["rgbd"][:, ::4, ::4]
        final = out
    save("trajectory_clevr.npz", zmean=full["zmean"], zstd=full["zstd"], cb_seed=full["cb_seed"], **final)


def pcd_inputs():
    """seeded inputs of the per-view unprojection fixture (shared with tests/test_pointcloud_cpu.py through the .npz)"""
    rs = np.random.RandomState(77)
    h, w = 20, 24
    depth = rs.uniform(1.4, 3.4, (h, w)).astype(np.float32)
    color = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    K = np.array([[497.77774 * w / 512, 0, w / 2], [0, 497.77774 * h / 512, h / 2], [0, 0, 1]])
    a = 0.3
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ np.array([[1, 0, 0], [0, np.cos(0.1), -np.sin(0.1)], [0, np.sin(0.1), np.cos(0.1)]])
    Rt = np.eye(4)
    Rt[:3, :3], Rt[:3, 3] = R, [0.2, -0.1, 0.35]
    return depth, color, K, Rt


def gen_pcd():
    """the reference's own `prepare_pcd` (inference_pipeline.py:1014-1039), run with a two-attribute stand-in for the Open3D
    container it fills (PointCloud.points / .colors, Vector3dVector = the array itself): pins the float64 unprojection that
    `merged_pcds.ply` is made of"""
    print("per-view unprojection (prepare_pcd)")
    import types
    import sgam.inference_pipeline as ref_ip

    class _PC:
        points = colors = None
    ref_ip.o3d.geometry = types.SimpleNamespace(PointCloud=_PC)
    ref_ip.o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.asarray(a))
    depth, color, K, Rt = pcd_inputs()
    pcd = InfiniteSceneGeneration.prepare_pcd(None, depth, color, K, Rt)
    save("prepare_pcd.npz", depth=depth, color=color, K=K, Rt=Rt, points=np.asarray(pcd.points), colors=np.asarray(pcd.colors))


def pose_set_inputs():
    """seeded inputs of the `trajectory` pose-set fixture: a cam0_to_world.txt of 12 rigid poses (frame index + row-major 4x4 per
    line, the format load_poses reads, inference_pipeline.py:362-368) and the frame indices of two known frames"""
    rs = np.random.RandomState(5)
    rows = []
    for k in range(12):
        a, b = 0.05 * k, 0.02 * k
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = Rz @ Rx, rs.uniform(-1, 1, 3) + [0.3 * k, 0, 0]
        rows.append(np.concatenate([[100 + 3 * k], T.reshape(-1)]))
    return np.stack(rows), [(103, 0, 0), (106, 1, 0)]


def gen_pose_sets():
    """The reference's OTHER pose sets and visiting orders, run as they stand on a bare instance (their constructor hard-codes
    trajectory_shape = 'grid', inference_pipeline.py:67 / :82, so they are reachable only by editing it): prepare_spiral (:206-288),
    prepare_ring (:290-360, the 'cylinder' shape), prepare_trajectory (:370-421) + load_poses, get_known_map (:144-155),
    row_major_order / column_major_order (:477-501), the trajectory branch of get_src_grid_coords (:531).  Open3D is only used
    there to DRAW coordinate frames: stubbed to no-ops."""
    print("pose sets: spiral / cylinder / trajectory, known map, visiting orders")
    import tempfile
    import types
    import sgam.inference_pipeline as ref_ip

    class _Frame:
        def transform(self, T):
            return self
    ref_ip.o3d.geometry = types.SimpleNamespace(TriangleMesh=types.SimpleNamespace(create_coordinate_frame=lambda size=1: _Frame()))
    ref_ip.o3d.visualization = types.SimpleNamespace(draw_geometries=lambda frames: None)
    if not hasattr(np, "int"):
        np.int = int                     # load_poses uses the alias numpy 2 removed
    out = {}
    for data in ("google_earth", "clevr-infinite"):
        tag = "ge" if data == "google_earth" else "clevr"
        o = object.__new__(InfiniteSceneGeneration)
        o.data, o.step_size_denom = data, 2
        o.K = np.array([[248.88887, 0, 128], [0, 248.88887, 128], [0, 0, 1]]) if tag == "ge" else np.array([[355.5555, 0, 128], [0, 355.5555, 128], [0, 0, 1]])
        for shape, fn, kw in (("spiral", o.prepare_spiral, {}), ("cylinder", o.prepare_ring, {"horizontal_offset": 0.002})):
            fn((9, 1), {}, "out", **kw)
            nodes = [n for row in o.transform_grid for n in row]
            out[f"{tag}_{shape}_R"] = np.stack([n["R"] for n in nodes])
            out[f"{tag}_{shape}_t"] = np.stack([n["t"] for n in nodes])
            out[f"{tag}_{shape}_position"] = np.stack([n["position"] for n in nodes])
            out[f"{tag}_{shape}_rows"] = np.array([len(row) for row in o.transform_grid])
    # trajectory: poses from a file, known frames from the files of the result folder
    poses, known = pose_set_inputs()
    with tempfile.TemporaryDirectory() as d:
        np.savetxt(os.path.join(d, "cam0_to_world.txt"), poses)
        for idx, i, j in known:
            np.save(os.path.join(d, f"dm_{idx:05d}_{i:02d}_{j:02d}.npy"), np.zeros((2, 2), np.float32))
        o = object.__new__(InfiniteSceneGeneration)
        o.data, o.step_size_denom, o.K = "google_earth", 2, np.eye(3)
        o.grid_transform_path = ref_ip.Path(d)
        km = o.get_known_map()
        out["known_keys"] = np.array(sorted(km.keys()))
        out["known_orig_idx"] = np.array([km[k]["orig_frame_idx"] for k in sorted(km.keys())])
        order = o.prepare_trajectory(7, km, d, pose_path=os.path.join(d, "cam0_to_world.txt"))
        out["traj_order"] = np.array(order)
        nodes = [row[0] for row in o.transform_grid]
        out["traj_R"] = np.stack([n["R"] for n in nodes])
        out["traj_t"] = np.stack([n["t"] for n in nodes])
        out["traj_position"] = np.stack([n["position"] for n in nodes])
        out["traj_visited"] = np.array([n["visited"] for n in nodes])
        out["traj_anchor_keys"] = np.array(sorted(o.anchor_poses.keys()))
        o.trajectory_shape, o.num_src, o.curr = "trajectory", 3, 5
        out["traj_srcs_of_5"] = np.array(o.get_src_grid_coords((5, 0))[0])
        out["traj_closest_anchor_of_6"] = np.array(o.get_closest_anchor(nodes[6])["grid_coord"])
    out["poses_txt"] = poses
    out["known_files"] = np.array(known)
    # visiting orders on a 3 x 4 grid
    o = object.__new__(InfiniteSceneGeneration)
    o.output_dim = (3, 4)
    o.transform_grid = [[{"visited": False} for _ in range(4)] for _ in range(3)]
    out["row_major_3x4"] = np.array(o.row_major_order())
    out["column_major_3x4"] = np.array(o.column_major_order())
    out["zig_zag_3x4"] = np.array(o.zig_zag_order())
    save("pose_sets.npz", **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    if not only or "poses" in only:
        gen_pose_sets()
    if not only or "pcd" in only:
        gen_pcd()
    rgb0, dm0 = gen_splat() if (not only or "splat" in only or "traj" in only) else (None, None)
    if not only or "inv" in only:
        gen_invwarp()
    if not only or "ops" in only:
        gen_ops()
    if not only or "full" in only:
        gen_full("google_earth", 64, "ge64")
        gen_full("google_earth", 256, "ge256")
        gen_full("clevr-infinite", 256, "clevr256_topk1", topk=1)
    if not only or "config1" in only:
        # BASELINE config 1 verbatim: CLEVR (16 384 codes), forward(x, extrapolation_mask, get_codebook_count=True), arg-min path
        gen_full("clevr-infinite", 256, "clevr256_argmin", plain_input=True)
    if not only or "traj" in only:
        gen_trajectory(rgb0, dm0)
    if not only or "topk4" in only:
        gen_topk4()
    if not only or "config5" in only:
        gen_config5()
    if not only or "free" in only:
        gen_trajectory_free()
    if not only or "clevr" in only:
        gen_trajectory_clevr()
    if not only or "train" in only:
        gen_train()
        gen_train_gan()
    if not only or "trainfull" in only:
        gen_train_full()
    if not only or "lpips" in only:
        gen_lpips()
