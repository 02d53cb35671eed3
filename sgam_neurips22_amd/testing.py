"""Seeded synthetic weights and inputs shared by the tests, the golden-vector generator, smoke() and
bench.py (trained checkpoints live on Google Drive and cannot be fetched: SURVEY §8c)."""
import zlib

import numpy as np
import torch


def synthetic_state_dict(reference_state_dict, seed=0, codebook_sigma=1.0):
    """Deterministic fp32 weights for every tensor of `reference_state_dict` (name -> tensor or shape):
    conv/linear weights ~ U(-b, b) with b = sqrt(3 / fan_in) (unit-gain variance), biases ~ U(-0.1, 0.1),
    GroupNorm weight 1 + 0.1 N(0,1) / bias 0.1 N(0,1), codebook rows ~ N(0, codebook_sigma^2).
    One independent generator per tensor (seeded by crc32 of its name) so the result does not depend on
    module construction order."""
    out = {}
    for name in sorted(reference_state_dict.keys()):
        ref = reference_state_dict[name]
        shape = tuple(ref.shape) if hasattr(ref, "shape") else tuple(ref)
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))
        if name.endswith("embedding.weight"):
            t = torch.randn(shape, generator=g) * codebook_sigma
        elif ".norm" in name or name.startswith("norm") or "norm_out" in name:
            t = torch.randn(shape, generator=g) * 0.1
            if name.endswith("weight"):
                t = t + 1.0
        elif name.endswith("bias"):
            t = (torch.rand(shape, generator=g) * 2 - 1) * 0.1
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            b = (3.0 / fan_in) ** 0.5
            t = (torch.rand(shape, generator=g) * 2 - 1) * b
        out[name] = t.float()
    return out


def hot_path_keys(state_dict):
    """state_dict keys the inference hot path owns (drops loss.* / perceptual_loss.* of a full checkpoint)."""
    return {k: v for k, v in state_dict.items() if not (k.startswith("loss.") or k.startswith("perceptual_loss."))}


def codebook_from_stats(zmean, zstd, n_e, dim, seed=0):
    """Seeded codebook with rows ~ N(zmean, zstd^2): matched to the latent statistics so that the nearest-
    neighbour structure is well conditioned (the reference's default init U(+-1/n_e) gives exact fp32 ties,
    SURVEY D4).  Fixtures store (zmean, zstd, seed) and assert the top-2 margin."""
    g = torch.Generator().manual_seed(4242 + int(seed))
    return (torch.randn((n_e, dim), generator=g) * float(zstd) + float(zmean)).float()


def top2_relative_gap(z_tokens, codebook):
    """Relative gap between the nearest and second-nearest codeword distance of each token (float64)."""
    z, cb = z_tokens.detach().double().cpu(), codebook.detach().double().cpu()
    d = (z ** 2).sum(1, keepdim=True) + (cb ** 2).sum(1) - 2 * z @ cb.t()
    top2 = torch.topk(d, 2, dim=1, largest=False).values
    return (top2[:, 1] - top2[:, 0]) / top2[:, 0].abs().clamp_min(1e-12)


def rect_hole_input(B, H, W, seed=3, hole_frac=0.3):
    """Seeded model input: x (B,4,H,W) ~ U(-1,1) with a rectangular hole (rgb 0, depth -2, mask 1)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((B, 4, H, W), generator=g) * 2 - 1
    mask = torch.zeros((B, 1, H, W), dtype=torch.bool)
    hh = int(H * hole_frac ** 0.5)
    ww = int(W * hole_frac ** 0.5)
    y0, x0 = H // 5, W // 4
    mask[:, :, y0:y0 + hh, x0:x0 + ww] = True
    x[:, :3][mask.expand(B, 3, H, W)] = 0.0
    x[:, 3:][mask] = -2.0
    return x, mask


def config1_input(res=256):
    """BASELINE config 1: x = seeded U(-1, 1) (1, 4, res, res), extrapolation mask all false"""
    g = torch.Generator().manual_seed(1)
    return torch.rand((1, 4, res, res), generator=g) * 2 - 1, torch.zeros((1, 1, res, res), dtype=torch.bool)


def seeded_tensor(tag, shape, scale=1.0, shift=0.0):
    """Deterministic N(shift, scale^2) fp32 tensor keyed by a string tag (inputs of the per-op fixtures)."""
    g = torch.Generator().manual_seed(zlib.crc32(tag.encode()) % (2 ** 31))
    return torch.randn(tuple(shape), generator=g) * scale + shift


def _small_rotation(rs, s):
    w = rs.randn(3) * s
    th = np.linalg.norm(w)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def synth_warp_inputs(seed, B, N, H, W, rot_scale=0.05, bad_depth=False):
    """Seeded forward-splat inputs: features (B,N,3,H,W) with ~2% exact zeros, depths in the GoogleEarth
    template range (optionally 5% zeros / 2% negatives), pinhole K, random small rigid motions."""
    rs = np.random.RandomState(seed)
    f = rs.uniform(-1, 1, (B, N, 3, H, W)).astype(np.float32)
    f[rs.rand(*f.shape) < 0.02] = 0
    d = rs.uniform(1.4, 3.4, (B, N, H, W)).astype(np.float32)
    if bad_depth:
        d[rs.rand(*d.shape) < 0.05] = 0
        d[rs.rand(*d.shape) < 0.02] *= -1
    fx = W * 0.97
    K = np.array([[fx, 0, W / 2], [0, fx, H / 2], [0, 0, 1]], np.float32)
    Ks = np.tile(K, (B, N, 1, 1))
    T = np.tile(np.eye(4, dtype=np.float32), (B, N, 1, 1))
    for b in range(B):
        for n in range(N):
            T[b, n, :3, :3] = _small_rotation(rs, rot_scale)
            T[b, n, :3, 3] = rs.randn(3) * 0.1
    return f, d, Ks, T


def synth_invwarp_inputs(seed, N, H, W, rot_scale=0.05, bad=False):
    """Seeded inverse-warp inputs (B=1): src images / depths, a target depth (optionally with zeros,
    negatives and -1 pixels that defeat the `sum > 0` validity test), K, target->source motions."""
    rs = np.random.RandomState(seed)
    im = rs.uniform(-1, 1, (1, N, 3, H, W)).astype(np.float32)
    d = rs.uniform(1.4, 3.4, (1, N, H, W)).astype(np.float32)
    td = rs.uniform(1.4, 3.4, (1, H, W)).astype(np.float32)
    if bad:
        td[rs.rand(*td.shape) < 0.05] = 0
        td[rs.rand(*td.shape) < 0.03] *= -1
        im[rs.rand(*im.shape) < 0.1] = -1
    fx = W * 0.97
    K = np.array([[fx, 0, W / 2], [0, fx, H / 2], [0, 0, 1]], np.float32)
    Ks = np.tile(K, (1, N, 1, 1))
    T = np.tile(np.eye(4, dtype=np.float32), (1, N, 1, 1))
    for n in range(N):
        T[0, n, :3, :3] = _small_rotation(rs, rot_scale)
        T[0, n, :3, 3] = rs.randn(3) * 0.1
    return im, d, td, Ks, K, T


# (tag, seed, B, N, H, W, rot_scale, depth_range, bad_depth) — shared by gen_golden.py and the tests
SPLAT_CASES = [("a", 1, 1, 1, 64, 64, 0.05, None, False), ("b", 2, 1, 3, 64, 64, 0.05, None, False),
               ("c", 3, 2, 3, 48, 80, 0.05, None, False), ("d", 4, 1, 5, 64, 64, 0.2, None, False),
               ("e", 5, 1, 3, 64, 64, 0.05, (0.5, 3.0), False), ("f", 6, 2, 2, 64, 64, 0.1, None, True)]
# (tag, seed, N, H, W, rot_scale, bad)
INVWARP_CASES = [("a", 1, 1, 64, 64, 0.05, False), ("b", 2, 3, 64, 64, 0.05, False), ("c", 3, 3, 48, 80, 0.2, False),
                 ("d", 5, 3, 64, 64, 0.1, True)]
# (tag, module kind, ctor args, input shape NCHW)
OP_CASES = [("res128", "ResnetBlock", dict(in_channels=128, out_channels=128), (1, 128, 24, 20)),
            ("res128_256", "ResnetBlock", dict(in_channels=128, out_channels=256), (2, 128, 16, 16)),
            ("res512_256", "ResnetBlock", dict(in_channels=512, out_channels=256), (1, 512, 8, 8)),
            ("attn256", "AttnBlock", dict(in_channels=256), (1, 256, 16, 24)),
            ("attn512", "AttnBlock", dict(in_channels=512), (2, 512, 8, 8)),
            ("down128", "Downsample", dict(in_channels=128, with_conv=True), (1, 128, 16, 20)),
            ("up256", "Upsample", dict(in_channels=256, with_conv=True), (1, 256, 8, 12))]


def sha256(t):
    """digest of a tensor's / array's raw bytes: bit-exact outputs are pinned by hash where the data would be large"""
    import hashlib
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()


def _fresh_row(zmean, zstd, dim, row, k):
    g = torch.Generator().manual_seed(99991 + 31 * int(row) + 7 * int(k))
    return (torch.randn((dim,), generator=g) * float(zstd) + float(zmean)).float()


def apply_codebook_repairs(cb, repairs, zmean, zstd):
    """replay the (row, k) replacements recorded by `repaired_codebook`"""
    cb = cb.clone()
    for row, k in np.asarray(repairs).reshape(-1, 2).tolist():
        cb[row] = _fresh_row(zmean, zstd, cb.shape[1], row, k)
    return cb


def repaired_codebook(z_tokens, zmean, zstd, n_e, dim, seed=0, min_gap=1e-4, max_rounds=200):
    """`codebook_from_stats`, then every token whose nearest / second-nearest codeword are closer than `min_gap`
    (relative) gets its SECOND-nearest row replaced by a fresh seeded row, until the arg-min of every token is well
    conditioned.  Seed search alone cannot reach that for thousands of tokens (P(gap < 1e-4) is ~5e-3 per token).
    Returns (codebook, [(row, k), ...]) — the list replays through `apply_codebook_repairs`."""
    cb = codebook_from_stats(zmean, zstd, n_e, dim, seed)
    z = z_tokens.detach().double().cpu()
    repairs, count = [], {}
    for _ in range(max_rounds):
        c = cb.double()
        d = (z ** 2).sum(1, keepdim=True) + (c ** 2).sum(1) - 2 * z @ c.t()
        top2 = torch.topk(d, 2, dim=1, largest=False)
        gap = (top2.values[:, 1] - top2.values[:, 0]) / top2.values[:, 0].abs().clamp_min(1e-12)
        bad = torch.nonzero(gap < min_gap).reshape(-1).tolist()
        if not bad:
            return cb, repairs
        for t in bad:
            row = int(top2.indices[t, 1])
            k = count.get(row, 0)
            count[row] = k + 1
            cb[row] = _fresh_row(zmean, zstd, dim, row, k)
            repairs.append((row, k))
    raise RuntimeError("repaired_codebook did not converge")


def config5_batch(src0_rgb_u8, src0_depth, res=512):
    """BASELINE config 5 inputs (numpy fp32, keys / shapes of the reference's get_x batch, model.py:179-209): a batch
    of FOUR candidate target poses, each conditioned on the same TWO 512x512 source frames — the reference's
    GoogleEarth seed0 template at grid cell (0,0) and a synthetic seeded frame at cell (1,0) — with the pipeline's own
    pose geometry (inference_pipeline.py:157-204: start pose, half grid steps, OpenGL->CV flip, T_rel = T_tgt T_src^-1)."""
    from .inference_pipeline import _GL2CV, _START, intrinsics, synthetic_seed_frame
    start, step_i, _ = _START["google_earth"]
    K = intrinsics("google_earth", (res, res))

    def w2c(k, yaw=0.0):
        c2w = np.eye(4)
        c, s = np.cos(yaw), np.sin(yaw)
        c2w[:3, :3] = start[:3, :3] @ np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
        c2w[:3, 3] = start[:3, 3] + step_i / 2 * k
        return np.linalg.inv(c2w @ _GL2CV)

    T_srcs = [w2c(0.0), w2c(1.0)]
    T_tgts = [w2c(2.0), w2c(1.5, 0.02), w2c(0.5, -0.03), w2c(3.0)]
    rgb1, d1 = synthetic_seed_frame("google_earth", 1, res)
    lut = (np.arange(256, dtype=np.float64) / 127.5 - 1.0).astype(np.float32)
    imgs = np.stack([lut[src0_rgb_u8], lut[rgb1]])                      # (2,H,W,3)
    deps = np.stack([np.asarray(src0_depth, np.float32), d1])           # (2,H,W)
    R = np.zeros((4, 2, 3, 3))
    t = np.zeros((4, 2, 3))
    for b, Tt in enumerate(T_tgts):
        for n, Ts in enumerate(T_srcs):
            Trel = Tt @ np.linalg.inv(Ts)
            R[b, n], t[b, n] = Trel[:3, :3], Trel[:3, 3]
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    return {"Ks": f32(np.tile(K, (4, 2, 1, 1))), "R_rels": f32(R), "t_rels": f32(t),
            "src_imgs": f32(np.tile(imgs[None], (4, 1, 1, 1, 1))), "src_depths": f32(np.tile(deps[None], (4, 1, 1, 1))),
            "dst_img": np.zeros((4, res, res, 3), np.float32), "dst_depth": np.zeros((4, res, res), np.float32)}


def small_train_params(params):
    """the small conditional VQGAN of the training-step tests and of tests/golden/train_step_small.npz: 64 x 64 input, widths
    128 / 256 (the GroupNorm kernels of the forward path take multiples of 128 channels), one Downsample, 32 x 32 latent of 32
    channels, 64 codes, attention at 32 and in the middle (applied to a copy of a full params dict)"""
    import copy
    p = copy.deepcopy(params)
    p["embed_dim"], p["n_embed"] = 32, 64
    p["ddconfig"].update(ch=128, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[32], resolution=64, z_channels=32)
    return p


def train_batch():
    """(x, mask, x_dst) of that fixture: B = 2, 64 x 64"""
    x, mask = rect_hole_input(2, 64, 64, seed=5)
    x_dst = (seeded_tensor("train.dst", (2, 4, 64, 64), scale=0.5)).clamp(-1, 1)
    return x, mask, x_dst


def synthetic_disc_state_dict(reference_state_dict, seed=0):
    """deterministic PatchGAN weights in the spirit of its weights_init (Conv ~ N(0, 0.02), BatchNorm weight ~ N(1, 0.02), bias 0)
    — scaled up 5x so that the logits are not all near zero — with fresh running statistics"""
    out = {}
    for name in sorted(reference_state_dict.keys()):
        ref = reference_state_dict[name]
        g = torch.Generator().manual_seed((zlib.crc32(("disc." + name).encode()) + 7919 * seed) % (2 ** 31))
        if name.endswith("num_batches_tracked"):
            t = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_mean"):
            t = torch.zeros(tuple(ref.shape))
        elif name.endswith("running_var"):
            t = torch.ones(tuple(ref.shape))
        elif ref.dim() == 4:
            t = torch.randn(tuple(ref.shape), generator=g) * 0.1
        elif name.endswith("weight"):
            t = 1.0 + 0.02 * torch.randn(tuple(ref.shape), generator=g)
        else:
            t = 0.05 * torch.randn(tuple(ref.shape), generator=g)
        out[name] = t
    return out


def synthetic_vgg_state_dict(lpips_state_dict, seed=0):
    """He-style deterministic weights for the VGG16 trunk of an LPIPS state_dict (keys net.slice<k>.<i>.weight / .bias) — a stand-in
    for torchvision's ImageNet checkpoint, which cannot be fetched offline; every other key is passed through unchanged"""
    out = {}
    for name in sorted(lpips_state_dict.keys()):
        ref = lpips_state_dict[name]
        if not name.startswith("net."):
            out[name] = ref.clone()
            continue
        g = torch.Generator().manual_seed((zlib.crc32(("vgg." + name).encode()) + 7919 * seed) % (2 ** 31))
        if ref.dim() == 4:
            fan_in = ref.shape[1] * ref.shape[2] * ref.shape[3]
            out[name] = torch.randn(tuple(ref.shape), generator=g) * (2.0 / fan_in) ** 0.5
        else:
            out[name] = 0.05 * torch.randn(tuple(ref.shape), generator=g)
    return out
