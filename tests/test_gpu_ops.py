"""GPU: every HIP kernel family of the VQGAN against the oracle's torch-CPU fp32 op on the same seeded input
(tolerances are fp32-roundoff class: both sides accumulate in fp32, only the summation order differs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from sgam_neurips22_amd import ops, testing

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(params=["split", "mfma"])
def f32_mode(request):
    """run a test under both evaluations of fp32 products: fp16-split MFMA (default) and fp32-in MFMA"""
    old = ops.F32_MODE
    ops.set_f32_mode(request.param)
    yield request.param
    ops.set_f32_mode(old)


def _pack(w, mode):
    return ops.pack_conv_weight(w.to(DEV), dtype="f32x" if mode == "split" else torch.float32)


def _close(a, b, atol, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= atol * scale, f"{what}: max abs err {err:.3e} (scale {scale:.3g})"


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,C,HW", [(1, 4, 64 * 64), (2, 256, 16 * 16), (1, 130, 77)])
def test_layout_hops(B, C, HW):
    x = testing.seeded_tensor("layout", (B, C, HW, 1)).to(DEV)
    y = ops.nchw_to_nhwc(x)
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).cpu())
    assert torch.equal(ops.nhwc_to_nchw(y).cpu(), x.cpu())


# (tag, B, Cin, Cout, H, W, k, stride, pad(t,l,b,r), upsample)
CONV_CASES = [
    ("3x3_128_128_big", 1, 128, 128, 64, 64, 3, 1, (1, 1, 1, 1), False),     # 128x128 tile path (M=4096? -> 64 tiles)
    ("3x3_128_128_ragged", 1, 128, 128, 23, 19, 3, 1, (1, 1, 1, 1), False),  # M not a tile multiple
    ("3x3_256_256", 2, 256, 256, 16, 16, 3, 1, (1, 1, 1, 1), False),
    ("3x3_512_512_splitk", 1, 512, 512, 8, 8, 3, 1, (1, 1, 1, 1), False),    # split-K path
    ("3x3_256_512", 1, 256, 512, 16, 16, 3, 1, (1, 1, 1, 1), False),
    ("down_s2_asym", 1, 128, 128, 32, 32, 3, 2, (0, 0, 1, 1), False),
    ("down_s2_odd", 1, 128, 128, 17, 21, 3, 2, (0, 0, 1, 1), False),
    ("up2x_256", 1, 256, 256, 16, 12, 3, 1, (1, 1, 1, 1), True),
    ("1x1_256_256", 1, 256, 256, 24, 24, 1, 1, (0, 0, 0, 0), False),
    ("1x1_128_256_nin", 2, 128, 256, 16, 16, 1, 1, (0, 0, 0, 0), False),
    ("3x3_128_4_out", 1, 128, 4, 32, 32, 3, 1, (1, 1, 1, 1), False),         # Cout padded to 64, n_valid 4
    ("3x3_512_256_out", 1, 512, 256, 16, 16, 3, 1, (1, 1, 1, 1), False),
    ("3x3_128_128_256sq", 1, 128, 128, 256, 256, 3, 1, (1, 1, 1, 1), False),  # the dominant layer shape
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: c[0])
def test_conv2d_matches_oracle(case, f32_mode):
    tag, B, Cin, Cout, H, W, k, stride, pad, ups = case
    x = testing.seeded_tensor(tag + ".x", (B, Cin, H, W))
    w = testing.seeded_tensor(tag + ".w", (Cout, Cin, k, k), scale=(1.0 / (Cin * k * k)) ** 0.5)
    b = testing.seeded_tensor(tag + ".b", (Cout,), scale=0.1)
    xr = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    xr = F.pad(xr, (pad[1], pad[3], pad[0], pad[2]))
    ref = F.conv2d(xr, w, b, stride=stride)
    res_in = testing.seeded_tensor(tag + ".r", tuple(ref.shape))
    wp = _pack(w, f32_mode)
    out = ops.conv2d_nhwc(_nhwc(x).to(DEV), wp, b.to(DEV), cout=Cout, kh=k, kw=k, stride=stride, pad_t=pad[0],
                          pad_l=pad[1], pad_b=pad[2], pad_r=pad[3], upsample2x=ups)
    assert tuple(out.shape) == (B, ref.shape[2], ref.shape[3], Cout)
    _close(out.permute(0, 3, 1, 2), ref, 2e-5, tag)
    out2 = ops.conv2d_nhwc(_nhwc(x).to(DEV), wp, b.to(DEV), cout=Cout, kh=k, kw=k, stride=stride, pad_t=pad[0],
                           pad_l=pad[1], pad_b=pad[2], pad_r=pad[3], upsample2x=ups, residual=_nhwc(res_in).to(DEV))
    _close(out2.permute(0, 3, 1, 2), ref + res_in, 2e-5, tag + "+res")


@pytest.mark.parametrize("B,C,Cout,H,W,k,swish,ups", [(1, 128, 128, 40, 36, 3, True, False), (2, 256, 128, 16, 16, 3, True, False),
                                                       (1, 512, 512, 8, 8, 3, True, False), (1, 256, 256, 12, 12, 1, False, False),
                                                       (1, 256, 256, 10, 14, 3, True, True)])
def test_conv_with_fused_groupnorm_prologue(B, C, Cout, H, W, k, swish, ups):
    """GroupNorm(+swish) applied inside the conv's operand staging == normalise-then-conv of the reference, and
    bit-identical to this backend's own unfused path (same expression order)."""
    x = testing.seeded_tensor("gnconv.x", (B, C, H, W), 2.0, 0.3)
    g = 1 + 0.1 * testing.seeded_tensor("gnconv.g", (C,))
    bt = 0.1 * testing.seeded_tensor("gnconv.b", (C,))
    w = testing.seeded_tensor("gnconv.w", (Cout, C, k, k), scale=(1.0 / (C * k * k)) ** 0.5)
    b = testing.seeded_tensor("gnconv.bias", (Cout,), scale=0.1)
    h = F.group_norm(x, 32, g, bt, eps=1e-6)
    if swish:
        h = h * torch.sigmoid(h)
    if ups:
        h = F.interpolate(h, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(h, w, b, padding=k // 2)
    xd = _nhwc(x).to(DEV)
    wp = ops.pack_conv_weight(w.to(DEV))           # the fused prologue lives in the fp32-MFMA kernel
    table = ops.groupnorm_stats(xd, g.to(DEV), bt.to(DEV))
    assert table.shape == (B, C, 2)
    fused = ops.conv2d_nhwc(xd, wp, b.to(DEV), cout=Cout, kh=k, kw=k, pad_t=k // 2, pad_l=k // 2, upsample2x=ups,
                            gn=(table, swish))
    _close(fused.permute(0, 3, 1, 2), ref, 3e-5, "fused GN conv")
    unfused = ops.conv2d_nhwc(ops.groupnorm_nhwc(xd, g.to(DEV), bt.to(DEV), swish), wp, b.to(DEV), cout=Cout, kh=k, kw=k,
                              pad_t=k // 2, pad_l=k // 2, upsample2x=ups)
    # same normalise/swish expression; statistics come from differently ordered fp64 sums -> last-bit agreement
    assert (fused - unfused).abs().max().item() <= 2e-6 * max(1.0, unfused.abs().max().item())


def test_conv_is_run_to_run_deterministic(f32_mode):
    x = _nhwc(testing.seeded_tensor("det.x", (1, 512, 8, 8))).to(DEV)
    wp = _pack(testing.seeded_tensor("det.w", (512, 512, 3, 3), scale=0.02), f32_mode)
    a = ops.conv2d_nhwc(x, wp, None, cout=512, kh=3, kw=3, pad_t=1, pad_l=1)
    for _ in range(3):
        assert torch.equal(a, ops.conv2d_nhwc(x, wp, None, cout=512, kh=3, kw=3, pad_t=1, pad_l=1))


@pytest.mark.parametrize("M,N,K,strided", [(256, 256, 256, False), (4096, 4096, 256, True), (300, 68, 64, False),
                                           (256, 4096, 256, False), (64, 256, 4096, False), (16, 512, 16, False),
                                           (64, 64, 48, False)])
def test_gemm_nt_transpose_detecting(M, N, K, strided, f32_mode):
    """asymmetric operands (so a swapped C-write cannot pass), optional row-strided A/B views."""
    a = testing.seeded_tensor("gemm.a", (M, 2 * K if strided else K))
    b = testing.seeded_tensor("gemm.b", (N, 2 * K if strided else K)) * torch.linspace(0.5, 1.5, N)[:, None]
    bias = testing.seeded_tensor("gemm.bias", (N,))
    av, bv = (a[:, :K], b[:, K:]) if strided else (a, b)
    ref = av.double() @ bv.double().t() + bias.double()
    ad, bd = a.to(DEV), b.to(DEV)
    avd, bvd = (ad[:, :K], bd[:, K:]) if strided else (ad, bd)
    out = ops.gemm_nt(avd, bvd, bias=bias.to(DEV))
    _close(out, ref.float(), 3e-6 * K ** 0.5, "gemm")
    rb = testing.seeded_tensor("gemm.rb", (M,))
    out = ops.gemm_nt(avd, bvd, bias=rb.to(DEV), bias_per_row=True)
    _close(out, (av.double() @ bv.double().t() + rb.double()[:, None]).float(), 3e-6 * K ** 0.5, "gemm row-bias")


@pytest.mark.parametrize("B,C,H,W", [(1, 128, 64, 64), (2, 256, 12, 12), (1, 512, 16, 16), (1, 128, 256, 256),
                                     (1, 256, 7, 5), (2, 256, 128, 128), (1, 512, 70, 70)])
@pytest.mark.parametrize("swish", [False, True])
def test_groupnorm_swish(B, C, H, W, swish):
    x = testing.seeded_tensor("gn.x", (B, C, H, W), 3.0, 0.5)
    g = 1 + 0.1 * testing.seeded_tensor("gn.g", (C,))
    bt = 0.1 * testing.seeded_tensor("gn.b", (C,))
    ref = F.group_norm(x, 32, g, bt, eps=1e-6)
    if swish:
        ref = ref * torch.sigmoid(ref)
    out = ops.groupnorm_nhwc(_nhwc(x).to(DEV), g.to(DEV), bt.to(DEV), swish)
    _close(out.permute(0, 3, 1, 2), ref, 2e-5, "groupnorm")


@pytest.mark.parametrize("rows,cols", [(64, 256), (128, 4096), (8, 16384), (5, 1024)])
def test_softmax_rows(rows, cols):
    s = testing.seeded_tensor("sm", (rows, cols), 4.0)
    s[0, 3] = 60.0  # a spike
    ref = F.softmax(s * 0.0625, dim=1)
    out = ops.softmax_rows_(s.to(DEV).clone(), 0.0625)
    _close(out, ref, 1e-6, "softmax")
    assert torch.allclose(out.sum(1).cpu(), torch.ones(rows), atol=1e-5)


@pytest.mark.parametrize("B,n", [(3, 256), (8, 256), (2, 64)])
def test_softmax_rows_block_diagonal(B, n):
    """the block-diagonal form (B images' scores as ONE matrix): every row is the soft-max over its own diagonal block, exact
    zeros elsewhere — fp32 in place and the 16-bit output form — and an attention block at batch B equals the per-image chain."""
    s = testing.seeded_tensor("smbd", (B * n, B * n), 4.0)
    ref = torch.zeros_like(s)
    for b in range(B):
        ref[b * n:(b + 1) * n, b * n:(b + 1) * n] = F.softmax(s[b * n:(b + 1) * n, b * n:(b + 1) * n] * 0.0625, dim=1)
    out = ops.softmax_rows_(s.to(DEV).clone(), 0.0625, block=n)
    _close(out, ref, 1e-6, "block-diagonal softmax")
    assert bool((out.cpu()[ref == 0] == 0).all()), "another image's keys must get exactly zero"
    p16 = ops.softmax_rows_h16(s.to(DEV), 0.0625, torch.bfloat16, block=n)
    assert bool((p16.float().cpu()[ref == 0] == 0).all()) and (p16.float().cpu() - ref).abs().max().item() <= 2 ** -8
    with pytest.raises(ops.SgamHipError):
        ops.softmax_rows_(s.to(DEV).clone(), 0.0625, block=n + 2)


def test_encode_head():
    x, mask = testing.rect_hole_input(2, 32, 48)
    w = testing.seeded_tensor("head.w", (4, 5, 1, 1), 0.4)
    b = testing.seeded_tensor("head.b", (4,), 0.1)
    ref = F.conv2d(torch.cat([x, mask.float()], 1), w, b)
    out = ops.encode_head(x.to(DEV), mask.to(DEV), w.to(DEV), b.to(DEV), ld=32)
    assert out.shape == (2, 32, 48, 32)
    _close(out[..., :4].permute(0, 3, 1, 2), ref, 1e-6, "encode head")
    assert torch.count_nonzero(out[..., 4:]).item() == 0
    out0 = ops.encode_head(x.to(DEV), None, w.to(DEV), b.to(DEV))
    _close(out0[..., :4].permute(0, 3, 1, 2), F.conv2d(torch.cat([x, torch.zeros_like(mask).float()], 1), w, b), 1e-6, "no mask")


@pytest.mark.parametrize("T,n_e", [(256, 4096), (256, 16384), (1024, 4096), (16, 4096)])
def test_vq_nearest_bit_exact_indices(T, n_e):
    from oracle import vqgan as OV
    z = testing.seeded_tensor("vq.z", (T, 256), 0.5, 0.1)
    seed = 0
    while True:
        cb = testing.codebook_from_stats(0.1, 0.5, n_e, 256, seed)
        if float(testing.top2_relative_gap(z, cb).min()) >= 1e-4:
            break
        seed += 1
    sd = {"quantize.embedding.weight": cb}
    side = int(T ** 0.5)
    zq_ref, idx_ref, d_ref, _ = OV.quantize(sd, z.view(1, side, T // side, 256).permute(0, 3, 1, 2))
    cbd = cb.to(DEV)
    idx, zq, dist = ops.vq_nearest(z.to(DEV), cbd, ops.row_sumsq(cbd), straight_through=True, want_dist=True)
    assert torch.equal(idx.cpu(), idx_ref.reshape(-1)), "codebook indices must be bit-exact on margin-guarded inputs"
    # z + (e - z) evaluated in the reference's order: bit-exact given equal indices
    assert torch.equal(zq.cpu(), zq_ref.permute(0, 2, 3, 1).reshape(T, 256))
    _close(dist, d_ref, 2e-6, "distances")
    # pure gather is a copy
    assert torch.equal(ops.vq_gather(cbd, idx).cpu(), cb[idx_ref.reshape(-1)])


def test_vq_first_index_wins_exact_ties():
    z = torch.zeros((64, 256))
    z[:, 0] = 1.0
    cb = testing.codebook_from_stats(0.0, 1.0, 4096, 256, 0)
    cb[7] = 0.0; cb[7, 0] = 1.0      # two identical nearest rows: 7 and 3000
    cb[3000] = cb[7]
    cbd = cb.to(DEV)
    idx, _, _ = ops.vq_nearest(z.to(DEV), cbd, ops.row_sumsq(cbd))
    assert torch.all(idx.cpu() == 7)


def test_vq_topk_order():
    d = testing.seeded_tensor("topk", (32, 4096)).abs()
    d[:, 100] = d[:, 5]  # a tie
    vals, inds = ops.vq_topk(d.to(DEV), 8)
    ref = torch.topk(d, 8, dim=1, largest=False)
    assert torch.equal(vals.cpu(), ref.values)
    for r in range(32):
        got, want = inds[r].cpu().tolist(), sorted(range(4096), key=lambda j: (float(d[r, j]), j))[:8]
        assert got == want


def test_split_mode_is_fp32_class_accurate():
    """fp16-split evaluation vs an fp64 reference, next to the fp32-MFMA evaluation: same error class, including
    small-magnitude activations, a wide dynamic range and softmax-like probabilities (a_scale = 1024)."""
    K, M, N = 4096, 256, 256
    a = testing.seeded_tensor("acc.a", (M, K)) * torch.logspace(-3, 1, K)[None, :]      # 1e-3 .. 10 columns
    b = testing.seeded_tensor("acc.b", (N, K), 0.03)
    ref = a.double() @ b.double().t()
    scale = (a.double().abs() @ b.double().abs().t()).clamp_min(1e-30)                 # sum |a||b| per output
    errs = {}
    for mode in ("mfma", "split"):
        ops.set_f32_mode(mode)
        out = ops.gemm_nt(a.to(DEV), b.to(DEV)).cpu().double()
        errs[mode] = ((out - ref).abs() / scale).max().item()
    prob = torch.softmax(testing.seeded_tensor("acc.p", (M, K), 3.0), dim=1)
    v = testing.seeded_tensor("acc.v", (N, K))
    refp = prob.double() @ v.double().t()
    outp = ops.gemm_nt(prob.to(DEV), v.to(DEV), a_scale=1024.0).cpu().double()
    errs["split_prob"] = ((outp - refp).abs().max() / refp.abs().max()).item()
    print("relative-to-sum|a||b| errors:", errs)
    assert errs["mfma"] <= 2e-7 and errs["split"] <= 4e-7 and errs["split_prob"] <= 1e-6


def test_cpu_tensor_is_rejected_loudly():
    with pytest.raises(ops.SgamHipError):
        ops.groupnorm_nhwc(torch.zeros(1, 4, 4, 128), torch.ones(128), torch.zeros(128), True)


@pytest.mark.parametrize("B,C,Cout,H,W,k", [(1, 128, 128, 128, 128, 3), (1, 128, 256, 136, 120, 1), (2, 128, 256, 32, 32, 1),
                                           (1, 128, 128, 256, 256, 3)])
def test_conv_epilogue_groupnorm_statistics(B, C, Cout, H, W, k):
    """split-mode conv emits the GroupNorm statistics of its output; GroupNorm(out) from those partials equals
    GroupNorm(out) with its own statistics pass (and the torch reference)."""
    ops.set_f32_mode("split")
    x = _nhwc(testing.seeded_tensor("cst.x", (B, C, H, W), 1.5, 0.2)).to(DEV)
    w = testing.seeded_tensor("cst.w", (Cout, C, k, k), scale=(1.0 / (C * k * k)) ** 0.5)
    res = _nhwc(testing.seeded_tensor("cst.r", (B, Cout, H, W))).to(DEV)
    g = (1 + 0.1 * testing.seeded_tensor("cst.g", (Cout,))).to(DEV)
    bt = (0.1 * testing.seeded_tensor("cst.b", (Cout,))).to(DEV)
    wp = ops.pack_conv_weight(w.to(DEV), dtype="f32x")
    out = ops.conv2d_nhwc(x, wp, None, cout=Cout, kh=k, kw=k, pad_t=k // 2, pad_l=k // 2, residual=res)
    if not hasattr(out, "_gn_partials"):
        pytest.skip("this shape runs split-K on this plan table: statistics come from the GroupNorm pass")
    y_fused = ops.groupnorm_nhwc(out, g, bt, True)
    y_plain = ops.groupnorm_nhwc(out.clone(), g, bt, True)        # clone drops the attached statistics
    ref = F.group_norm(out.permute(0, 3, 1, 2).cpu(), 32, g.cpu(), bt.cpu(), eps=1e-6)
    ref = ref * torch.sigmoid(ref)
    _close(y_plain.permute(0, 3, 1, 2), ref, 2e-5, "plain")
    _close(y_fused.permute(0, 3, 1, 2), ref, 2e-5, "fused statistics")


@pytest.mark.parametrize("B,C,Cout,H,W,swish,with_res", [(1, 128, 128, 64, 64, True, True), (2, 128, 256, 32, 48, False, False),
                                                        (1, 256, 128, 16, 16, True, False), (1, 128, 128, 256, 256, True, True),
                                                        (1, 128, 4, 256, 256, True, False), (2, 128, 8, 128, 128, True, True)])
def test_conv_groupnorm_fused_into_halo_staging(B, C, Cout, H, W, swish, with_res):
    """Conv3x3(GroupNorm(+swish)(x)) with the normalisation applied while the halo-staged kernel stages its input
    equals the two-pass form (GroupNorm kernel, then conv) and the torch reference; zero padding is post-norm."""
    ops.set_f32_mode("split")
    x = _nhwc(testing.seeded_tensor("cgf.x", (B, C, H, W), 1.3, 0.4)).to(DEV)
    w = testing.seeded_tensor("cgf.w", (Cout, C, 3, 3), scale=(1.0 / (C * 9)) ** 0.5)
    bias = testing.seeded_tensor("cgf.bias", (Cout,), 0.1).to(DEV)
    res = _nhwc(testing.seeded_tensor("cgf.r", (B, Cout, H, W))).to(DEV) if with_res else None
    g = (1 + 0.1 * testing.seeded_tensor("cgf.g", (C,))).to(DEV)
    bt = (0.1 * testing.seeded_tensor("cgf.b", (C,))).to(DEV)
    wp = ops.pack_conv_weight(w.to(DEV), dtype="f32x")
    kw = dict(cout=Cout, kh=3, kw=3, pad_t=1, pad_l=1, residual=res)
    fused = ops.conv2d_nhwc(x, wp, bias, norm=(g, bt, swish, 32, 1e-6), **kw)
    two_pass = ops.conv2d_nhwc(ops.groupnorm_nhwc(x, g, bt, swish), wp, bias, **kw)
    ref = F.group_norm(x.permute(0, 3, 1, 2).cpu(), 32, g.cpu(), bt.cpu(), eps=1e-6)
    if swish:
        ref = ref * torch.sigmoid(ref)
    ref = F.conv2d(ref.double(), w.double(), bias.cpu().double(), padding=1).float()
    if with_res:
        ref = ref + res.permute(0, 3, 1, 2).cpu()
    _close(two_pass.permute(0, 3, 1, 2), ref, 2e-5, "two-pass")
    _close(fused.permute(0, 3, 1, 2), ref, 2e-5, "fused")
    assert (fused - two_pass).abs().max().item() <= 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("plan", [(64, 128, 1), (64, 128, 3), (128, 128, 2), (64, 64, 1), (64, 64, 2)])
def test_halo_kernel_tile_plans(plan):
    """the halo-staged 3x3 kernel under each of its tile plans (8x8 and 8x16 patches, split-K) against fp64"""
    ops.set_f32_mode("split")
    B, C, Cout, H, W = 2, 64, 128, 24, 32
    x = _nhwc(testing.seeded_tensor("hk.x", (B, C, H, W))).to(DEV)
    w = testing.seeded_tensor("hk.w", (Cout, C, 3, 3), scale=(1.0 / (C * 9)) ** 0.5)
    wp = ops.pack_conv_weight(w.to(DEV), dtype="f32x")
    key = f"f32x|B{B}|{H}x{W}x{C}|{H}x{W}|N{Cout}|k3x3s1u0"
    old = ops.PLAN_CACHE.get(key)
    ops.PLAN_CACHE[key] = plan
    try:
        out = ops.conv2d_nhwc(x, wp, None, cout=Cout, kh=3, kw=3, pad_t=1, pad_l=1)
    finally:
        if old is None:
            ops.PLAN_CACHE.pop(key, None)
        else:
            ops.PLAN_CACHE[key] = old
    ref = F.conv2d(x.permute(0, 3, 1, 2).cpu().double(), w.double(), padding=1).float()
    _close(out.permute(0, 3, 1, 2), ref, 2e-5, f"halo plan {plan}")


def test_halo_kernel_64_channel_tile_with_fused_groupnorm():
    """plan tile (64, 64) on a 3x3 / s1 / p1 shape = the halo kernel with a 2 x 2 wavefront grid of 32 rows x 32 channels (whole-K
    workgroups for the 64 x 64 maps: no split-K plan, no combine launch): GroupNorm + swish fused into its staging and the output
    statistics from its epilogue, against the (64, 128) tile of the same kernel."""
    ops.set_f32_mode("split")
    B, C, H, W = 1, 256, 64, 64
    x = _nhwc(testing.seeded_tensor("h64.x", (B, C, H, W), 1.5, 0.3)).to(DEV)
    w = testing.seeded_tensor("h64.w", (C, C, 3, 3), scale=(1.0 / (C * 9)) ** 0.5)
    b = testing.seeded_tensor("h64.b", (C,), scale=0.1).to(DEV)
    g, bt = (1 + 0.1 * testing.seeded_tensor("h64.g", (C,))).to(DEV), (0.1 * testing.seeded_tensor("h64.bt", (C,))).to(DEV)
    res = _nhwc(testing.seeded_tensor("h64.r", (B, C, H, W))).to(DEV)
    wp = ops.pack_conv_weight(w.to(DEV), dtype="f32x")
    key = f"f32x|B{B}|{H}x{W}x{C}|{H}x{W}|N{C}|k3x3s1u0"
    old = ops.PLAN_CACHE.get(key)
    outs = {}
    try:
        for plan in ((64, 128, 1), (64, 64, 1)):
            ops.PLAN_CACHE[key] = plan
            outs[plan] = ops.conv2d_nhwc(x, wp, b, cout=C, kh=3, kw=3, pad_t=1, pad_l=1, residual=res, norm=(g, bt, True, 32, 1e-6))
    finally:
        if old is None:
            ops.PLAN_CACHE.pop(key, None)
        else:
            ops.PLAN_CACHE[key] = old
    a, c = outs[(64, 128, 1)], outs[(64, 64, 1)]
    assert (a - c).abs().max().item() <= 1e-6      # same products, same K order per output
    assert hasattr(c, "_gn_partials")
    st = ops.groupnorm_meanrstd(c).cpu()
    og = c.permute(0, 3, 1, 2).cpu().double().reshape(B, 32, -1)
    assert torch.allclose(st[:, :, 0].double(), og.mean(-1), rtol=0, atol=1e-6)
    assert torch.allclose(st[:, :, 1].double(), (og.var(-1, unbiased=False) + 1e-6).rsqrt(), rtol=1e-5, atol=0)


@pytest.mark.parametrize("C,H,W,ks,folds", [(512, 16, 16, 4, False), (256, 32, 32, 4, True), (256, 64, 64, 4, False),
                                            (512, 16, 16, 16, True), (512, 16, 16, 8, True), (256, 32, 32, 8, True), (256, 32, 32, 2, False),
                                            (128, 16, 32, 4, True)])
def test_splitk_combine_delivers_groupnorm_statistics(C, H, W, ks, folds):
    """The split-K combine of a conv also leaves the partial GroupNorm sums of its output; the next 3x3 conv normalises
    with them while staging (one 32-workgroup fold in between, no pass over the tensor).  Equals the explicit form.
    `folds`: the 16^2 / 32^2 maps — the combine runs group-major (<= 16 chunk partials per image) and a consumer whose
    workgroups walk at most two channel slabs folds them in its own prologue: no fold launch at all."""
    ops.set_f32_mode("split")
    x = _nhwc(testing.seeded_tensor("sms.x", (1, C, H, W), 1.1, 0.3)).to(DEV)
    w1 = testing.seeded_tensor("sms.w1", (C, C, 3, 3), scale=(1.0 / (C * 9)) ** 0.5)
    w2 = testing.seeded_tensor("sms.w2", (C, C, 3, 3), scale=(1.0 / (C * 9)) ** 0.5)
    res = _nhwc(testing.seeded_tensor("sms.r", (1, C, H, W))).to(DEV)
    g = (1 + 0.1 * testing.seeded_tensor("sms.g", (C,))).to(DEV)
    bt = (0.1 * testing.seeded_tensor("sms.b", (C,))).to(DEV)
    p1, p2 = ops.pack_conv_weight(w1.to(DEV), dtype="f32x"), ops.pack_conv_weight(w2.to(DEV), dtype="f32x")
    kw = dict(cout=C, kh=3, kw=3, pad_t=1, pad_l=1)
    key = f"f32x|B1|{H}x{W}x{C}|{H}x{W}|N{C}|k3x3s1u0"
    old = ops.PLAN_CACHE.get(key)
    ops.PLAN_CACHE[key] = (64, 128, ks)                # force a split-K plan
    try:
        h = ops.conv2d_nhwc(x, p1, None, residual=res, **kw)
        assert hasattr(h, "_gn_partials")
        assert (h._gn_partials[1] <= 16) == (H * W <= 1024)
        ref_h = F.conv2d(x.permute(0, 3, 1, 2).cpu().double(), w1.double(), padding=1).float() + res.permute(0, 3, 1, 2).cpu()
        _close(h.permute(0, 3, 1, 2), ref_h, 2e-5, "conv + residual through the combine")
        st = ops.groupnorm_meanrstd(h).cpu()
        hg = h.permute(0, 3, 1, 2).cpu().double().reshape(1, 32, -1)
        assert torch.allclose(st[0, :, 0].double(), hg.mean(-1)[0], rtol=0, atol=2e-6)
        assert torch.allclose(st[0, :, 1].double(), (hg.var(-1, unbiased=False) + 1e-6).rsqrt()[0], rtol=2e-6, atol=0)
        fused = ops.conv2d_nhwc(h, p2, None, norm=(g, bt, True, 32, 1e-6), **kw)
        plain = ops.conv2d_nhwc(ops.groupnorm_nhwc(h.clone(), g, bt, True), p2, None, **kw)
        recs, _ = ops.kernel_timeline(lambda: ops.conv2d_nhwc(h, p2, None, norm=(g, bt, True, 32, 1e-6), **kw))
        names = [r[0] for r in recs]
        assert any("true,false,true" in n for n in names) == folds, names          # the folding form of the halo kernel
        assert any("gn_finalize" in n for n in names) != folds, names              # ... replaces the fold launch
        for _ in range(3):
            assert torch.equal(fused, ops.conv2d_nhwc(h, p2, None, norm=(g, bt, True, 32, 1e-6), **kw))
    finally:
        if old is None:
            ops.PLAN_CACHE.pop(key, None)
        else:
            ops.PLAN_CACHE[key] = old
    scale = plain.abs().max().item()
    assert (fused - plain).abs().max().item() <= 3e-6 * scale


@pytest.mark.parametrize("n,spread", [(1024, 1.0), (4096, 1.0), (4096, 6.0), (256, 1.0)])
def test_fused_attention_matches_fp64_and_the_gemm_chain(n, spread):
    """sgam_attention_f32x (one pass over the keys, online soft-max) against softmax(q k^T / 16) v in fp64 and against
    the GEMM -> softmax -> GEMM chain it replaces; `spread` widens the logits so that the running maximum moves and a few
    keys dominate (the regime where the rescaling and the fp16 split of tiny probabilities matter)."""
    C = 256
    qkv = testing.seeded_tensor(f"attn.{n}", (n, 3 * C)).to(DEV)
    qkv[:, :2 * C] *= spread
    scale = C ** -0.5
    o = ops.attention(qkv, C, scale)
    q, k, v = (qkv[:, i * C:(i + 1) * C].double() for i in range(3))
    ref = torch.softmax(q @ k.t() * scale, dim=1) @ v
    _close(o, ref, 2e-5, "fused attention vs fp64")
    s = ops.gemm_nt(qkv[:, :C], qkv[:, C:2 * C])
    ops.softmax_rows_(s, scale)
    vt = qkv[:, 2 * C:].t().contiguous()
    chain = ops.gemm_nt(s, vt, a_scale=1024.0)
    _close(o, chain, 2e-5, "fused attention vs GEMM chain")
    for _ in range(10):                                          # run-to-run deterministic (LDS-DMA staging, counted waits)
        assert torch.equal(o, ops.attention(qkv, C, scale))
    assert not ops.attention_fusable(n, 512) and not ops.attention_fusable(n + 32, C)


def test_fused_attention_at_the_512sq_size():
    """n = 16384 tokens (the 512x512 configuration: 1024 workgroups, 64 key blocks per range): a sample of query rows
    against fp64."""
    n, C = 16384, 256
    qkv = testing.seeded_tensor("attn.16k", (n, 3 * C)).to(DEV)
    scale = C ** -0.5
    o = ops.attention(qkv, C, scale)
    rows = torch.arange(0, n, 67, device=DEV)
    q, k, v = qkv[rows, :C].double(), qkv[:, C:2 * C].double(), qkv[:, 2 * C:].double()
    ref = torch.softmax(q @ k.t() * scale, dim=1) @ v
    _close(o[rows], ref, 2e-5, "fused attention, n = 16384")


@pytest.mark.parametrize("B,n", [(1, 4096), (1, 1024), (2, 4096), (8, 256)])
def test_attention_with_fused_projection_matches_the_separate_launches(B, n):
    """sgam_attention_proj_f32x_batched (ABI v8): the merge of the key ranges as the operand staging of proj_out + residual, against
    `attention` followed by the 1x1 convolution it replaces (same merge arithmetic, a different order of the K = 256 MFMA chain:
    fp32 round-off), against fp64, run-to-run identical, and the chunk statistics it leaves against the sums of what it wrote."""
    C = 256
    scale = C ** -0.5
    qkv = testing.seeded_tensor(f"attnP.{B}.{n}", (B * n, 3 * C)).to(DEV)
    x = testing.seeded_tensor(f"attnP.x.{B}.{n}", (B * n, C)).to(DEV)
    w = (testing.seeded_tensor("attnP.w", (C, C)) * 0.06).to(DEV)
    bias = testing.seeded_tensor("attnP.b", (C,)).to(DEV)
    wp = ops.split_rows(w, ops._pow2_scale(float(w.abs().max())))
    got = ops.attention_proj(qkv, C, scale, wp, bias, x, B=B)
    o = ops.attention(qkv, C, scale, B=B)
    sep = ops.gemm_nt(o, wp, bias=bias, residual=x)
    _close(got, sep, 2e-6, "fused projection vs separate launches")
    ref = o.double() @ w.double().t() + bias.double() + x.double()
    _close(got, ref, 2e-5, "fused projection vs fp64 product of the attention output")
    for _ in range(5):
        assert torch.equal(got, ops.attention_proj(qkv, C, scale, wp, bias, x, B=B))
    part, chunks = got._gn_partials
    assert chunks == n // 32
    st = part.view(B * chunks, 32, 2)
    blk = got.double().view(B * chunks, 32, 32, C // 32)                     # (tile, row, group, channel in group)
    assert torch.allclose(st[..., 0], blk.sum(dim=(1, 3)), rtol=1e-6, atol=1e-4)
    assert torch.allclose(st[..., 1], (blk * blk).sum(dim=(1, 3)), rtol=1e-6, atol=1e-4)
    # without a residual / bias
    got0 = ops.attention_proj(qkv, C, scale, wp, None, None, B=B)
    _close(got0, o.double() @ w.double().t(), 2e-5, "fused projection, no bias / residual")


@pytest.mark.parametrize("B,n,dt", [(2, 4096, "f32"), (4, 4096, "f32"), (8, 4096, "f32"), (3, 1024, "f32"), (8, 256, "f32"),
                                    (4, 4096, "fp16"), (8, 4096, "bf16")])
def test_batched_attention_keeps_every_query_inside_its_image(B, n, dt):
    """sgam_attention_{f32x,h16}_batched: B images stacked along the rows in ONE launch sequence (fewer key ranges per image
    as the batch fills the chip: 8, 4, 2; never more than 2048 keys per range).  Every image's rows against its own unbatched run (same arithmetic per key
    block, a different number of ranges merged at the end: fp32 round-off) and image 0 against fp64 — a query that saw another
    image's keys would be off by O(1)."""
    C = 256
    scale = C ** -0.5
    qkv = testing.seeded_tensor(f"attnB.{B}.{n}", (B * n, 3 * C)).to(DEV)
    if dt == "f32":
        o = ops.attention(qkv, C, scale, B=B)
        solo = torch.cat([ops.attention(qkv[b * n:(b + 1) * n], C, scale) for b in range(B)])
        tol_solo, tol64 = 2e-6, 2e-5
    else:
        qkv = qkv.to(ops.DTYPES[dt])
        o = ops.attention_h16(qkv, C, scale, B=B)
        solo = torch.cat([ops.attention_h16(qkv[b * n:(b + 1) * n], C, scale) for b in range(B)])
        tol_solo, tol64 = (2e-3, 4e-3) if dt == "fp16" else (1.6e-2, 3e-2)
    assert o.shape == (B * n, C) and torch.isfinite(o.float()).all()
    _close(o.float(), solo.float(), tol_solo, "batched vs per-image launches")
    assert torch.equal(o, ops.attention(qkv, C, scale, B=B) if dt == "f32" else ops.attention_h16(qkv, C, scale, B=B))
    q, k, v = (qkv[:n, i * C:(i + 1) * C].double() for i in range(3))
    _close(o[:n].float(), torch.softmax(q @ k.t() * scale, dim=1) @ v, tol64, "image 0 vs fp64")
    q, k, v = (qkv[(B - 1) * n:, i * C:(i + 1) * C].double() for i in range(3))
    _close(o[(B - 1) * n:].float(), torch.softmax(q @ k.t() * scale, dim=1) @ v, tol64, "last image vs fp64")


@pytest.mark.parametrize("case", [(1, 4096, 256), (2, 256, 512), (3, 1024, 256)], ids=lambda c: f"B{c[0]}n{c[1]}C{c[2]}")
def test_fused_groupnorm_qkv_gemm(case):
    """csrc/gemm_gn_f32x.hip: GroupNorm(x) @ [Wq; Wk; Wv]^T + bias with the normalisation applied while the operand panel is
    staged, against torch fp64 (AttnBlock, model.py:168-175) and against the two-launch path it replaces"""
    B, n, C = case
    x = testing.seeded_tensor(f"gnqkv.x{n}", (B * n, C), 1.4, 0.3).to(DEV)
    g = (1 + 0.2 * testing.seeded_tensor("gnqkv.g", (C,))).to(DEV)
    bt = (0.2 * testing.seeded_tensor("gnqkv.b", (C,))).to(DEV)
    w = (testing.seeded_tensor("gnqkv.w", (3 * C, C)) * C ** -0.5).to(DEV)
    bias = (0.1 * testing.seeded_tensor("gnqkv.bias", (3 * C,))).to(DEV)
    old = ops.F32_MODE
    ops.set_f32_mode("split")
    try:
        assert ops.gemm_gn_fits(B * n, 3 * C, C, n)
        x4 = x.view(B, n, 1, C)
        mr = ops.groupnorm_meanrstd(x4)
        ws = ops.split_rows(w, ops._pow2_scale(float(w.abs().max())))
        got = ops.gemm_gn_f32x(x, mr, g, bt, ws, bias, n)
        assert torch.equal(got, ops.gemm_gn_f32x(x, mr, g, bt, ws, bias, n))
        two = ops.gemm_nt(ops.groupnorm_nhwc(x4, g, bt, False).view(B * n, C), ws, bias=bias)
    finally:
        ops.set_f32_mode(old)
    xn = F.group_norm(x.double().cpu().view(B, n, C).permute(0, 2, 1), 32, g.double().cpu(), bt.double().cpu(), eps=1e-6)
    ref = xn.permute(0, 2, 1).reshape(B * n, C) @ w.double().cpu().t() + bias.double().cpu()
    scale = ref.abs().max().item()
    assert (got.double().cpu() - ref).abs().max().item() <= 2e-6 * scale
    assert (got - two).abs().max().item() <= 4e-6 * scale


@pytest.mark.experimental
@pytest.mark.parametrize("case", [(1, 64, 64, 256, 256, True), (1, 128, 128, 256, 128, False), (2, 64, 64, 128, 256, False)],
                         ids=lambda c: f"B{c[0]}_{c[1]}x{c[2]}_{c[3]}to{c[4]}")
def test_panel_gemm_1x1_conv_with_residual_and_statistics(case):
    """the whole-K-panel kernel behind the 1x1 convolutions of the split-fp32 path (proj_out with its residual, nin_shortcut):
    result against torch fp64, bit-identical repeats, and the statistics it leaves for the next GroupNorm"""
    B, H, W, cin, cout, with_res = case
    x = testing.seeded_tensor("panel.x", (B, cin, H, W), 1.1, 0.2)
    w = testing.seeded_tensor("panel.w", (cout, cin, 1, 1), scale=cin ** -0.5)
    b = testing.seeded_tensor("panel.b", (cout,), scale=0.1)
    res = testing.seeded_tensor("panel.r", (B, cout, H, W)) if with_res else None
    old, old_panel = ops.F32_MODE, ops.PANEL_GEMM
    ops.set_f32_mode("split")
    ops.PANEL_GEMM = True                         # opt-in path (measured slightly slower than the tuned generic kernel in the frame)
    try:
        wp = ops.pack_conv_weight(w.to(DEV), dtype="f32x")
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        rd = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(DEV)
        out = ops.conv2d_nhwc(xd, wp, b.to(DEV), cout=cout, kh=1, kw=1, residual=rd)
        assert ops.PANEL_GEMM and hasattr(out, "_gn_partials") and out._gn_partials[1] == H * W // 64
        assert torch.equal(out, ops.conv2d_nhwc(xd, wp, b.to(DEV), cout=cout, kh=1, kw=1, residual=rd))
        st = ops.groupnorm_meanrstd(out).cpu()
    finally:
        ops.set_f32_mode(old)
        ops.PANEL_GEMM = old_panel
    ref = F.conv2d(x.double(), w.double(), b.double()) + (0 if res is None else res.double())
    assert (out.permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    og = ref.reshape(B, 32, -1)
    assert torch.allclose(st[:, :, 0].double(), og.mean(-1), rtol=0, atol=1e-5)
    assert torch.allclose(st[:, :, 1].double(), (og.var(-1, unbiased=False) + 1e-6).rsqrt(), rtol=1e-5, atol=0)


@pytest.mark.parametrize("B,produced", [(1, False), (2, False), (1, True)])
def test_attn_block_in_three_launches_matches_the_gemm_and_split_sequence(B, produced, monkeypatch):
    """ABI v9: the fused front end of the split-fp32 AttnBlock computes the q | k | v projection transposed and writes K / V^T
    straight in the attention's fragment order — the same normalisation expression, the same three MFMAs per product in the same
    order; against the gemm_gn_f32x + attn_split_kv sequence (fp32 round-off of a transposed MFMA chain: not bit-identical, measured),
    against the fp32 oracle of the block, run-to-run identical, one launch fewer"""
    from oracle import vqgan as OV
    from sgam_neurips22_amd.generative_sensing_module.modules.diffusionmodules import model as dm
    ops.set_f32_mode("split")
    mod = dm.AttnBlock(256)
    sd = testing.synthetic_state_dict(mod.state_dict(), seed=11)
    mod.load_state_dict(sd)
    mod = mod.to(DEV).eval()
    xc = testing.seeded_tensor("ab3.x", (B, 256, 64, 64), 1.0, 0.3)
    x = ops.nchw_to_nhwc(xc.to(DEV))
    if produced:
        # the block input as a convolution leaves it, with its chunk statistics (folded by one small launch in front of either form)
        w = testing.seeded_tensor("ab3.w", (256, 256, 3, 3), scale=(1.0 / (256 * 9)) ** 0.5).to(DEV)
        x = ops.conv2d_nhwc(x, ops.pack_conv_weight(w, dtype="f32x"), None, cout=256, kh=3, kw=3, pad_t=1, pad_l=1)
        assert hasattr(x, "_gn_partials") and 1 <= x._gn_partials[1] <= 128
        xc = x.permute(0, 3, 1, 2).cpu()
    with torch.no_grad():
        monkeypatch.setattr(ops, "ATTN_BLOCK_F32X", False)
        sep = mod.forward_nhwc(x)                                    # (the first call of either form packs its weights)
        recs0, _ = ops.kernel_timeline(lambda: mod.forward_nhwc(x))
        monkeypatch.setattr(ops, "ATTN_BLOCK_F32X", True)
        fused = mod.forward_nhwc(x)
        recs1, _ = ops.kernel_timeline(lambda: mod.forward_nhwc(x))
        again = mod.forward_nhwc(x)
    names0, names1 = [r[0] for r in recs0], [r[0] for r in recs1]
    assert any("attn_qkv_gn_f32x" in k for k in names1) and not any("split_kv" in k or "gemm_gn_f32x" in k for k in names1), names1
    assert len(names1) == len(names0) - 1, (names0, names1)
    assert torch.equal(fused, again) and torch.equal(fused._gn_partials[0], again._gn_partials[0])
    d = (fused - sep).abs().max().item()
    print(f"[attn_block_f32x B={B}] max |fused - separate| = {d:.3e} (max |out| {sep.abs().max().item():.3f})")
    _close(fused, sep, 2e-6, "three-launch block vs q|k|v GEMM + split launch")
    ref = OV.attn_block({"a." + k: v for k, v in sd.items()}, "a", xc).permute(0, 2, 3, 1)
    _close(fused, ref.to(DEV), 2e-5, "three-launch block vs the fp32 oracle")
    assert fused._gn_partials[1] == sep._gn_partials[1]
    assert torch.allclose(fused._gn_partials[0], sep._gn_partials[0], rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("B,n", [(1, 256), (3, 256), (2, 128)])
def test_small_attention_in_one_launch_matches_the_gemm_chain_and_fp64(B, n):
    """sgam_attention_small_f32x (ABI v9): scores, soft-max and P v of the 16 x 16 blocks (n tokens per image, C = 512) in one launch,
    against fp64 per image, against the split-fp32 GEMM chain it replaces (fp32 round-off), run-to-run identical, every query inside
    its image"""
    C = 512
    scale = C ** -0.5
    ops.set_f32_mode("split")
    qkv = (testing.seeded_tensor(f"attnS.{B}.{n}", (B * n, 3 * C)) * 1.3).to(DEV)
    assert ops.attention_small_fits(n, C, B)
    o = ops.attention_small(qkv, C, scale, B=B)
    recs, _ = ops.kernel_timeline(lambda: ops.attention_small(qkv, C, scale, B=B))
    assert len(recs) == 1 and "attn_small_f32x" in recs[0][0], [r[0] for r in recs]
    for _ in range(4):
        assert torch.equal(o, ops.attention_small(qkv, C, scale, B=B))
    for b in range(B):
        blk = qkv[b * n:(b + 1) * n]
        q, k, v = blk[:, :C].double(), blk[:, C:2 * C].double(), blk[:, 2 * C:].double()
        ref = torch.softmax(q @ k.t() * scale, dim=1) @ v
        _close(o[b * n:(b + 1) * n], ref, 4e-6, f"small attention vs fp64, image {b}")
        # the chain: v^T, q k^T, row soft-max, P v on the split-fp32 GEMM
        vt = ops.nhwc_to_nchw(blk[:, 2 * C:].unsqueeze(0).unsqueeze(0), c=C).view(C, n)
        s = ops.gemm_nt(blk[:, :C], blk[:, C:2 * C])
        ops.softmax_rows_(s, scale)
        chain = ops.gemm_nt(s, vt, a_scale=1024.0)
        _close(o[b * n:(b + 1) * n], chain, 4e-6, f"small attention vs the GEMM chain, image {b}")
    # a large-magnitude row (soft-max nearly one-hot) and a constant row (uniform weights)
    qkv2 = qkv.clone()
    qkv2[5, :C] *= 40.0
    qkv2[7, :C] = 0.0
    o2 = ops.attention_small(qkv2, C, scale, B=B)
    blk = qkv2[:n]
    ref = torch.softmax(blk[:, :C].double() @ blk[:, C:2 * C].double().t() * scale, dim=1) @ blk[:, 2 * C:].double()
    _close(o2[:n], ref, 4e-6, "small attention, peaked and uniform rows")


def test_abi_v9_entries_refuse_what_they_do_not_support():
    """error behaviour of the round-5 entry points through the raw C ABI: unsupported shapes / a scale that is not a power of two ->
    SGAM_EINVAL (-1), a short workspace -> SGAM_EWORKSPACE (-3), a misaligned operand -> SGAM_EALIGN (-2); none of them launches"""
    from sgam_neurips22_amd import _lib
    lib = _lib.load()
    C, n = 256, 4096
    assert lib.sgam_attn_block_f32x_workspace_bytes(n, C, 1) == lib.sgam_attention_f32x_batched_workspace_bytes(n, C, 1) + n * C * 4
    assert lib.sgam_attn_block_f32x_workspace_bytes(n, 512, 1) == -1 and lib.sgam_attn_block_h16_workspace_bytes(1000, C, 1) == -1
    assert lib.sgam_attention_small_f32x_fits(256, 512, 3) == 1 and lib.sgam_attention_small_f32x_fits(256, 256, 1) == 0
    assert lib.sgam_attention_small_f32x_fits(192, 512, 1) == 0
    x = torch.zeros((n, C), device=DEV)
    mr = torch.zeros((1, 32, 2), device=DEV)
    g = torch.ones((C,), device=DEV)
    w = ops.split_rows(torch.zeros((3 * C, C), device=DEV), 1.0)
    wp = ops.split_rows(torch.zeros((C, C), device=DEV), 1.0)
    b3 = torch.zeros((3 * C,), device=DEV)
    out = torch.empty((n, C), device=DEV)
    need = lib.sgam_attn_block_f32x_workspace_bytes(n, C, 1)
    ws = torch.empty((need,), device=DEV, dtype=torch.uint8)

    def call(scale=1 / 16.0, ws_bytes=need, xp=None, nn=n):
        return lib.sgam_attn_block_f32x(xp if xp is not None else ops._p(x), C, ops._p(mr), ops._p(g), ops._p(g), ops._p(w.planes), 1.0, ops._p(b3), nn, C, 1,
                                        scale, ops._p(wp.planes), 1.0, None, ops._p(out), C, None, ops._p(ws), ws_bytes, None)
    assert call() == 0
    assert call(scale=0.07) == -1                      # folded into q: must be an exact power of two
    assert call(nn=1000) == -1
    assert call(ws_bytes=need - 1) == -3
    assert call(xp=x.data_ptr() + 4) == -2
    # the small attention: wrong channel count, a row stride that is not a multiple of four
    q = torch.zeros((256, 3 * 512), device=DEV)
    o = torch.empty((256, 512), device=DEV)
    assert lib.sgam_attention_small_f32x(ops._p(q), ops._p(q[:, 512:]), ops._p(q[:, 1024:]), 1536, 256, 512, 1, 0.05, ops._p(o), 512, None) == 0
    assert lib.sgam_attention_small_f32x(ops._p(q), ops._p(q[:, 512:]), ops._p(q[:, 1024:]), 1536, 256, 256, 1, 0.05, ops._p(o), 512, None) == -1
    assert lib.sgam_attention_small_f32x(ops._p(q), ops._p(q[:, 512:]), ops._p(q[:, 1024:]), 1534, 256, 512, 1, 0.05, ops._p(o), 512, None) == -1
    # the 16-bit block: no statistics, a bad type code
    xh = torch.zeros((n, C), device=DEV, dtype=torch.bfloat16)
    wf = ops.pack_qkv_weight_h16(torch.zeros((3 * C, C), device=DEV), torch.bfloat16)
    oh = torch.empty((n, C), device=DEV, dtype=torch.bfloat16)
    needh = lib.sgam_attn_block_h16_workspace_bytes(n, C, 1)
    wsh = torch.empty((needh,), device=DEV, dtype=torch.uint8)
    part = torch.zeros((64 * 32 * 2,), device=DEV, dtype=torch.float64)

    def callh(ht=0, partial=None, ws_bytes=needh):
        return lib.sgam_attn_block_h16(ops._p(xh), C, partial if partial is not None else ops._p(part), 64, ops._p(g), ops._p(g), 1e-6, ops._p(wf), ops._p(b3),
                                       ht, n, C, 1, 1 / 16.0, ops._p(oh), C, ops._p(wsh), ws_bytes, None)
    assert callh() == 0
    assert callh(ht=2) == -1
    assert callh(partial=0) == -1
    assert callh(ws_bytes=needh - 1) == -3
    torch.cuda.synchronize()
