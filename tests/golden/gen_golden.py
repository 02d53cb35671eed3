"""Golden-vector generator — BUILD CONTAINER ONLY (needs /root/reference; never runs on the GPU box).

Imports the reference (yshen47/SGAM_NeurIPS22) with the sys.modules stubs of _ref_import.py, runs ITS code on
seeded inputs and writes small .npz fixtures next to this file.  The fixtures are data (inputs + the
reference's outputs); no reference source is copied.  Re-run:  python tests/golden/gen_golden.py

Fixtures
  splat_*.npz      render_projection_from_srcs_fast (warp.py:193-286), torch.use_deterministic_algorithms(True)
  invwarp_*.npz    InfiniteSceneGeneration.inverse_warping (inference_pipeline.py:662-743)
  vqgan_ops.npz    ResnetBlock / AttnBlock / Downsample / Upsample / GroupNorm+swish / VectorQuantizer2
  vqgan_full_*.npz VQModel.forward on seeded synthetic weights + margin-guarded codebook
  trajectory_ge.npz 3 steps of InfiniteSceneGeneration.one_step_prediction (GoogleEarth seed0)
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


def gen_full(dataset, res, tag, topk=None):
    print(f"full model {dataset} {res}x{res} topk={topk}")
    p = R.load_params(dataset)
    torch.manual_seed(0)
    model = VQModel(**p).eval()
    sd = testing.synthetic_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd)
    x, mask = testing.rect_hole_input(1, res, res, seed=3)
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
    save(f"vqgan_full_{tag}.npz", dataset=dataset, res=res, zmean=zmean, zstd=zstd, cb_seed=cb_seed,
         weight_abs_sums=wsum, dec_sub=dec[..., ::step, ::step], dec_step=step, dec_sum=float(dec.double().sum()),
         dec_abs_sum=float(dec.double().abs().sum()), indices=idx, pre_quant=pre, quant=quant,
         topk=-1 if topk is None else topk)


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


if __name__ == "__main__":
    only = sys.argv[1:]
    rgb0, dm0 = gen_splat() if (not only or "splat" in only or "traj" in only) else (None, None)
    if not only or "inv" in only:
        gen_invwarp()
    if not only or "ops" in only:
        gen_ops()
    if not only or "full" in only:
        gen_full("google_earth", 64, "ge64")
        gen_full("google_earth", 256, "ge256")
        gen_full("clevr-infinite", 256, "clevr256_topk1", topk=1)
    if not only or "traj" in only:
        gen_trajectory(rgb0, dm0)
