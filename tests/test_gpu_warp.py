"""GPU: the two warps, the depth codec and the frame feedback — bit-for-bit against the reference's golden
vectors AND the C oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

from conftest import bits_equal
from oracle import warp as OW
from sgam_neurips22_amd import ops, testing
from sgam_neurips22_amd.point_rendering.warp import render_projection_from_srcs_fast

pytestmark = pytest.mark.gpu
DEV = "cuda"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731


@pytest.mark.parametrize("case", testing.SPLAT_CASES, ids=lambda c: c[0])
def test_forward_splat_bit_exact(golden, case):
    tag, seed, B, N, H, W, rs, dr, bad = case
    g = golden(f"splat_{tag}.npz")
    f, d, Ks, T = testing.synth_warp_inputs(seed, B, N, H, W, rs, bad)
    r = render_projection_from_srcs_fast(t(f), t(d), t(Ks[:, 0]), t(Ks), t(T), src_num=N, depth_range=dr, parallel=True)
    assert bits_equal(r[0].cpu().numpy(), g["merge_depths"])
    assert bits_equal(r[1].cpu().numpy(), g["merge_feats"])
    assert np.array_equal(r[2].cpu().numpy(), g["extrapolation_mask"])
    assert np.array_equal(np.packbits(r[3].cpu().numpy()), g["mask"])
    assert np.array_equal(r[5].cpu().numpy(), g["idx"].astype(np.int64))
    assert bits_equal(r[6].cpu().numpy(), g["projected_features"])
    fused = torch.from_numpy(f).view(B, N, 3, -1).permute(0, 2, 3, 1).reshape(B, 3, -1).permute(0, 2, 1)
    assert torch.equal(r[4].cpu(), fused)


@pytest.mark.parametrize("dataset", ["google_earth", "clevr-infinite"])
@pytest.mark.parametrize("channels_last", [False, True])
def test_fused_x_output_and_normalisation(dataset, channels_last):
    f, d, Ks, T = testing.synth_warp_inputs(11, 2, 3, 64, 64, 0.05, True)
    if dataset == "clevr-infinite":
        d = d * 4
    o = OW.forward_splat(f, d, Ks[:, 0], Ks, T)
    nd = OW.normalise_depth(torch.from_numpy(o["merge_depths"]), torch.from_numpy(o["extrapolation_mask"]), dataset)
    Kinv = torch.inverse(torch.from_numpy(Ks).reshape(-1, 3, 3))
    feats = t(f.transpose(0, 1, 3, 4, 2)) if channels_last else t(f)
    r = ops.forward_splat(feats, t(d), t(Ks[:, 0]), Kinv.to(DEV), t(T).reshape(-1, 4, 4), channels_last=channels_last,
                          dataset=dataset, want=("x", "extrap", "merge_depths"))
    assert bits_equal(r["x"][:, :3].cpu().numpy(), o["merge_feats"])
    assert bits_equal(r["x"][:, 3:].cpu().numpy(), nd.numpy())
    assert np.array_equal(r["extrap"].cpu().numpy().astype(bool), o["extrapolation_mask"])
    # standalone K12 agrees with the fused one
    wd, em = ops.depth_normalise(r["merge_depths"], dataset, compute_mask=True)
    assert bits_equal(wd.cpu().numpy(), nd.numpy()) and np.array_equal(em.cpu().numpy().astype(bool), o["extrapolation_mask"])


def test_ge_template_splat(golden):
    g, tr = golden("splat_ge_seed0.npz"), golden("trajectory_ge.npz")
    src = ops.rgb_lut(DEV)[t(tr["seed_rgb"]).long()][None, None]          # (1,1,H,W,3)
    Kinv = torch.inverse(torch.from_numpy(g["K"]))[None]
    r = ops.forward_splat(src, t(tr["seed_depth"])[None, None], t(g["K"])[None], Kinv.to(DEV), t(g["T"])[None],
                          channels_last=True, want=("merge_depths", "merge_feats", "extrap", "inb_mask"))
    assert int(r["inb_mask"].sum()) == 62403
    assert bits_equal(r["merge_depths"].cpu().numpy(), g["merge_depths"])
    mf = r["merge_feats"].cpu().numpy()
    assert np.array_equal(np.packbits(mf == 0), g["merge_feats_zero"])
    assert np.array_equal(np.round((mf + 1) * 127.5).astype(np.uint8), g["merge_feats_u8"])
    assert np.array_equal(np.packbits(r["extrap"].cpu().numpy().astype(bool)), g["extrapolation_mask"])


def test_splat_is_deterministic_under_collisions():
    """many-to-one collisions (all points project onto a few pixels): identical result on every run."""
    f, d, Ks, T = testing.synth_warp_inputs(21, 1, 5, 128, 128, 0.3, False)
    Kinv = torch.inverse(torch.from_numpy(Ks).reshape(-1, 3, 3)).to(DEV)
    args = (t(f), t(d), t(Ks[:, 0]), Kinv, t(T).reshape(-1, 4, 4))
    a = ops.forward_splat(*args, want=("merge_feats", "merge_depths"))
    o = OW.forward_splat(f, d, Ks[:, 0], Ks, T)
    for _ in range(5):
        b = ops.forward_splat(*args, want=("merge_feats", "merge_depths"))
        assert torch.equal(a["merge_feats"], b["merge_feats"]) and torch.equal(a["merge_depths"], b["merge_depths"])
    assert bits_equal(a["merge_feats"].cpu().numpy(), o["merge_feats"])


def test_full_size_splat_properties():
    """BASELINE size (512x512, B=4, N=3): identity warp reproduces the source; mask == (depth <= 0)."""
    B, N, H, W = 4, 3, 512, 512
    rs = np.random.RandomState(0)
    f = rs.uniform(-1, 1, (B, N, 3, H, W)).astype(np.float32)
    d = rs.uniform(1.4, 3.4, (B, N, H, W)).astype(np.float32)
    K = np.array([[497.77774, 0, 256], [0, 497.77774, 256], [0, 0, 1]], np.float32)
    Ks = np.tile(K, (B, N, 1, 1)); T = np.tile(np.eye(4, dtype=np.float32), (B, N, 1, 1))
    Kinv = torch.inverse(torch.from_numpy(Ks).reshape(-1, 3, 3)).to(DEV)
    r = ops.forward_splat(t(f), t(d), t(Ks[:, 0]), Kinv, t(T).reshape(-1, 4, 4), want=("merge_feats", "merge_depths", "extrap"))
    # identity motion: every pixel maps onto itself; the last source (largest point index) wins
    assert torch.equal(r["merge_feats"].cpu(), torch.from_numpy(f[:, N - 1]))
    assert torch.equal(r["merge_depths"].cpu()[:, 0], torch.from_numpy(d[:, N - 1]))
    assert int(r["extrap"].sum()) == 0


def _both_paths(fn):
    """run fn() with the tiled (LDS z-tile, no global atomics) and the two-pass (global atomicMax) splat; returns both results"""
    old = ops.SPLAT_TILED
    try:
        ops.SPLAT_TILED = True
        a = fn()
        ops.SPLAT_TILED = False
        b = fn()
    finally:
        ops.SPLAT_TILED = old
    return a, b


@pytest.mark.parametrize("case", testing.SPLAT_CASES, ids=lambda c: c[0])
def test_tiled_splat_matches_reference_goldens(golden, case):
    """the target-owned-tile splat (ABI v6) against the REFERENCE's outputs: merged depth / features, mask and the pre-fill
    planes bit for bit, for every golden case (1..5 sources, B = 2, non-square 48 x 80, bad depths, depth_range branch)"""
    tag, seed, B, N, H, W, rs, dr, bad = case
    g = golden(f"splat_{tag}.npz")
    f, d, Ks, T = testing.synth_warp_inputs(seed, B, N, H, W, rs, bad)
    Kinv = torch.inverse(torch.from_numpy(Ks).reshape(-1, 3, 3)).to(DEV)
    want = ("merge_depths", "merge_feats", "extrap", "proj_feats", "proj_depth")
    a, b = _both_paths(lambda: ops.forward_splat(t(f), t(d), t(Ks[:, 0]), Kinv, t(T).reshape(-1, 4, 4), depth_range=dr, want=want))
    assert bits_equal(a["merge_depths"].cpu().numpy(), g["merge_depths"]) and bits_equal(a["merge_feats"].cpu().numpy(), g["merge_feats"])
    assert np.array_equal(a["extrap"].cpu().numpy().astype(bool), g["extrapolation_mask"].astype(bool))
    for k in want:
        assert torch.equal(a[k].view(torch.uint8), b[k].view(torch.uint8)), k


@pytest.mark.parametrize("B,N,H,W,rot", [(1, 3, 256, 256, 0.02), (4, 2, 512, 512, 0.05), (2, 5, 100, 72, 0.3), (16, 3, 256, 256, 0.02),
                                         (1, 1, 40, 24, 1.0), (3, 4, 136, 264, 0.6)])
def test_tiled_splat_equals_two_pass_splat(B, N, H, W, rot):
    """any geometry — gentle scene-loop motion, wild rotations that scatter a source bin over the whole image, ragged sizes,
    16 x 16 and 32 x 32 tiles — through the pointer-table entry point the scene loop uses: bit-identical to the two-pass form
    (itself pinned to the reference), run twice (no state survives in the workspace)"""
    f, d, Ks, T = testing.synth_warp_inputs(101 + B + N, B, N, H, W, rot, True)
    feats = [t(f[b, n].transpose(1, 2, 0)) for b in range(B) for n in range(N)]
    depths = [t(d[b, n]) for b in range(B) for n in range(N)]
    Kinv = torch.inverse(torch.from_numpy(Ks).reshape(-1, 3, 3)).to(DEV)
    fn = lambda: ops.forward_splat_srcs(feats, depths, t(Ks[:, 0]), Kinv, t(T).reshape(-1, 4, 4), B=B, dataset="google_earth",  # noqa: E731
                                        want=("x", "extrap", "merge_depths", "merge_feats"))
    a, b = _both_paths(fn)
    a2, _ = _both_paths(fn)
    for k in ("x", "extrap", "merge_depths", "merge_feats"):
        assert torch.equal(a[k].view(torch.uint8), b[k].view(torch.uint8)), k
        assert torch.equal(a[k].view(torch.uint8), a2[k].view(torch.uint8)), k
    o = OW.forward_splat(f, d, Ks[:, 0], Ks, T)
    assert bits_equal(a["merge_feats"].cpu().numpy(), o["merge_feats"]) and bits_equal(a["merge_depths"].cpu().numpy(), o["merge_depths"])


def test_tiled_splat_workspace_contract():
    lib = ops._lib.load()
    # per-tile bitmaps of registered bins [tiles][N * bins / 32] (16 x 16 tiles below 256 workgroups) + cached target pixels [N][HW]
    assert lib.sgam_forward_splat_workspace_zero_bytes(1, 3, 256, 256) == 256 * (3 * 256 // 32) * 4
    assert lib.sgam_forward_splat_workspace_bytes(1, 3, 256, 256) == 256 * 24 * 4 + 3 * 65536 * 4
    assert lib.sgam_forward_splat_workspace_bytes(1, 65, 64, 64) == -1                 # more sources than the pointer table holds
    assert lib.sgam_forward_splat_workspace_bytes(1, 1, 40000, 8) == -1          # H beyond the 16-bit packed target pixel
    f, d, Ks, T = testing.synth_warp_inputs(5, 1, 2, 64, 64, 0.05, False)
    Kinv = torch.inverse(torch.from_numpy(Ks).reshape(-1, 3, 3)).to(DEV)
    ws = torch.zeros((1024,), device=DEV, dtype=torch.uint8)
    out = torch.empty((1, 1, 64, 64), device=DEV)
    rc = lib.sgam_forward_splat_tiled_f32(ops._p(t(f)), 64 * 64, 1, ops._p(t(d)), ops._p(t(Ks[:, 0])), ops._p(Kinv), ops._p(t(T)), 1, 2, 64, 64,
                                          None, 0, ops._p(ws), 1024, ops._p(out), None, None, None, None, None, None)
    assert rc == -3                                                                # SGAM_EWORKSPACE: too small


@pytest.mark.parametrize("case", testing.INVWARP_CASES, ids=lambda c: c[0])
def test_inverse_warp_bit_exact(golden, case):
    tag, seed, N, H, W, s, bad = case
    im, d, td, Ks, K, T = testing.synth_invwarp_inputs(seed, N, H, W, s, bad)
    Kinv = torch.inverse(torch.from_numpy(K))[None].to(DEV)
    out = ops.inverse_warp(t(im), t(d), t(td), t(Ks).reshape(-1, 3, 3), Kinv, t(T).reshape(-1, 4, 4))
    assert bits_equal(out[0].cpu().numpy(), golden(f"invwarp_{tag}.npz")["warped"])
    assert bits_equal(out.cpu().numpy(), OW.inverse_warp(im, d, td, Ks, K[None], T))


def test_frame_feedback_matches_host_codec():
    dec = testing.seeded_tensor("fb", (2, 4, 64, 64), 0.7)
    dec[0, 0, 0, :4] = torch.tensor([-1.0, 1.0, 0.999999, -1.5])
    for ds in ("google_earth", "clevr-infinite"):
        rgb_f, depth, u8 = ops.frame_feedback(dec.to(DEV), ds, want_u8=True)
        for b in range(2):
            want_u8 = OW.rgb_to_uint8(dec[b, :3])
            assert np.array_equal(u8[b].cpu().numpy(), want_u8)
            want_f = (want_u8 / 127.5 - 1.0).astype(np.float32)
            assert bits_equal(rgb_f[b].cpu().numpy(), want_f)
            assert bits_equal(depth[b].cpu().numpy(), OW.denormalise_depth(dec[b, 3], ds).numpy())


def test_tiled_splat_on_two_streams_keeps_private_workspaces():
    """same-shaped tiled splats in flight on two HIP streams (what `distributed.ConcurrentScenes` does with two scenes): each
    stream has its own scratch — pass 2 of one must not read the target pixels / bitmaps pass 1 of the other is writing — and
    each result equals the two-pass form's, for several interleaved rounds"""
    B, N, H, W = 1, 3, 256, 256
    jobs = []
    for s in range(2):
        f, d, Ks, T = testing.synth_warp_inputs(300 + s, B, N, H, W, 0.05 + 0.4 * s, True)
        feats = [t(f[b, n].transpose(1, 2, 0)) for b in range(B) for n in range(N)]
        depths = [t(d[b, n]) for b in range(B) for n in range(N)]
        Kinv = torch.inverse(torch.from_numpy(Ks).reshape(-1, 3, 3)).to(DEV)
        jobs.append((feats, depths, t(Ks[:, 0]), Kinv, t(T).reshape(-1, 4, 4)))
    run = lambda j: ops.forward_splat_srcs(*j, B=B, dataset="google_earth", want=("x", "extrap", "merge_depths"))  # noqa: E731
    old = ops.SPLAT_TILED
    try:
        ops.SPLAT_TILED = False
        ref = [run(j) for j in jobs]
        torch.cuda.synchronize()
        ops.SPLAT_TILED = True
        ops._SPLAT_WS.clear()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for rounds in range(6):
            got = []
            for s in (0, 1):
                with torch.cuda.stream(streams[s]):
                    got.append(run(jobs[s]))
            torch.cuda.synchronize()
            for s in (0, 1):
                for k in ("x", "extrap", "merge_depths"):
                    assert torch.equal(got[s][k].view(torch.uint8), ref[s][k].view(torch.uint8)), (rounds, s, k)
        bufs = [v.data_ptr() for v in ops._SPLAT_WS.values()]
        assert len(bufs) == 2 and bufs[0] != bufs[1]
    finally:
        ops.SPLAT_TILED = old


def test_splat_auto_switches_forms_at_the_point_threshold():
    """SGAM_SPLAT_TILED=auto: below SPLAT_TILED_MIN_POINTS source points the two-pass form runs (no workspace), from the threshold
    on the tiled form — and the frame is the same either side of it"""
    B, N, H, W = 1, 3, 64, 64
    f, d, Ks, T = testing.synth_warp_inputs(77, B, N, H, W, 0.1, True)
    feats = [t(f[b, n].transpose(1, 2, 0)) for b in range(B) for n in range(N)]
    depths = [t(d[b, n]) for b in range(B) for n in range(N)]
    Kinv = torch.inverse(torch.from_numpy(Ks).reshape(-1, 3, 3)).to(DEV)
    run = lambda: ops.forward_splat_srcs(feats, depths, t(Ks[:, 0]), Kinv, t(T).reshape(-1, 4, 4), B=B, dataset="google_earth",  # noqa: E731
                                         want=("x", "extrap", "merge_depths"))
    old = (ops.SPLAT_TILED, ops.SPLAT_TILED_MIN_POINTS)
    try:
        ops.SPLAT_TILED = None                                   # "auto"
        ops.SPLAT_TILED_MIN_POINTS = B * N * H * W + 1
        assert ops._splat_workspace(DEV, B, N, H, W)[0] is None
        below = run()
        ops.SPLAT_TILED_MIN_POINTS = B * N * H * W
        assert ops._splat_workspace(DEV, B, N, H, W)[0] is not None
        at = run()
    finally:
        ops.SPLAT_TILED, ops.SPLAT_TILED_MIN_POINTS = old
    for k in ("x", "extrap", "merge_depths"):
        assert torch.equal(below[k].view(torch.uint8), at[k].view(torch.uint8)), k
