"""GPU: split-K without a combine launch (sgam_conv_desc.arrive, csrc/conv_f32x.hip xfixup / csrc/h16*.hip): the LAST split of an
output tile to arrive sums the partial tiles inside the convolution kernel.  Same slab order as the combine kernels, so the
output must be BIT-identical to the two-launch form whichever split arrives last; the statistics it leaves for the next
GroupNorm are chunked per tile instead of per combine workgroup (same sums up to fp32 / fp64 rounding of a different
partition).  Every case is repeated: a race between arrival and partial stores would show as a sporadic mismatch."""
import pytest
import torch

from sgam_neurips22_amd import ops, testing

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _fold(partials, chunks, B, n):
    """(B, 32, 2) {mean, E[x^2]} from chunk partials, in fp64 on the host"""
    p = partials.view(B, chunks, 32, 2).double().sum(1).cpu()
    return p / n


CASES = [  # B, C, N, H, W, k, (bm, bn, ksplit), bias, residual
    (1, 512, 512, 16, 16, 3, (64, 128, 8), True, True),
    (1, 512, 512, 16, 16, 3, (64, 128, 16), False, True),
    (1, 256, 256, 32, 32, 3, (64, 128, 4), True, False),
    (2, 256, 256, 32, 32, 3, (64, 128, 3), True, True),
    (1, 256, 256, 64, 64, 3, (64, 128, 2), False, False),
    (1, 512, 512, 16, 16, 1, (64, 64, 4), True, True),
    (1, 512, 256, 32, 32, 1, (64, 64, 2), True, False),
    (1, 256, 128, 32, 32, 3, (128, 128, 4), True, True),
    (3, 128, 128, 16, 24, 3, (64, 128, 2), True, True),
    (1, 512, 384, 16, 16, 1, (64, 64, 4), True, False),       # cpg = 12: no statistics on either path
]


@pytest.mark.parametrize("B,C,N,H,W,k,plan,use_bias,use_res", CASES)
def test_f32x_fixup_equals_the_combine_launch(B, C, N, H, W, k, plan, use_bias, use_res):
    ops.set_f32_mode("split")
    x = _nhwc(testing.seeded_tensor("fx.x", (B, C, H, W), 1.0, 0.2)).to(DEV)
    w = testing.seeded_tensor("fx.w", (N, C, k, k), scale=(1.0 / (C * k * k)) ** 0.5)
    bias = testing.seeded_tensor("fx.b", (N,)).to(DEV) if use_bias else None
    res = _nhwc(testing.seeded_tensor("fx.r", (B, N, H, W))).to(DEV) if use_res else None
    wp = ops.pack_conv_weight(w.to(DEV), dtype="f32x")
    kw = dict(cout=N, kh=k, kw=k, pad_t=k // 2, pad_l=k // 2)
    key = f"f32x|B{B}|{H}x{W}x{C}|{H}x{W}|N{wp.shape[0]}|k{k}x{k}s1u0"
    old, old_fix, old_panel = ops.PLAN_CACHE.get(key), ops.XFIXUP, ops.PANEL_GEMM
    ops.PLAN_CACHE[key] = plan
    try:
        ops.XFIXUP = False
        ops.PANEL_GEMM = False            # (the 1 x 1 cases are about the generic kernel's split-K plans, not the whole-K-panel GEMM)
        ref = ops.conv2d_nhwc(x, wp, bias, residual=res, **kw)
        recs, _ = ops.kernel_timeline(lambda: ops.conv2d_nhwc(x, wp, bias, residual=res, **kw))
        assert any("splitk_reduce" in r[0] for r in recs), [r[0] for r in recs]
        ops.XFIXUP = True
        got = ops.conv2d_nhwc(x, wp, bias, residual=res, **kw)
        recs, _ = ops.kernel_timeline(lambda: ops.conv2d_nhwc(x, wp, bias, residual=res, **kw))
        assert len(recs) == 1 and "splitk" not in recs[0][0], [r[0] for r in recs]     # ONE launch
        assert torch.equal(got, ref)
        assert int(wp.arrive.abs().max()) == 0                                        # the counters are left zero
        n = H * W * (N // 32)
        if hasattr(ref, "_gn_partials"):
            assert hasattr(got, "_gn_partials") and got._gn_partials[1] == H * W // plan[0]
            a, b = _fold(*ref._gn_partials, B, n), _fold(*got._gn_partials, B, n)
            assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (a - b).abs().max()
            og = got.permute(0, 3, 1, 2).cpu().double().reshape(B, 32, -1)
            assert torch.allclose(b[:, :, 0], og.mean(-1), rtol=0, atol=2e-6)
            assert torch.allclose(b[:, :, 1], (og * og).mean(-1), rtol=2e-6, atol=0)
        else:
            assert not hasattr(got, "_gn_partials")
        for _ in range(40):                                                            # arrival order varies from launch to launch
            again = ops.conv2d_nhwc(x, wp, bias, residual=res, **kw)
            assert torch.equal(again, ref)
            if hasattr(got, "_gn_partials"):
                assert torch.equal(again._gn_partials[0], got._gn_partials[0])
    finally:
        ops.XFIXUP, ops.PANEL_GEMM = old_fix, old_panel
        if old is None:
            ops.PLAN_CACHE.pop(key, None)
        else:
            ops.PLAN_CACHE[key] = old


def test_fixup_inside_a_captured_graph_replays():
    """the counters return to zero at the end of every launch, so a captured launch can be replayed"""
    ops.set_f32_mode("split")
    B, C, H, W = 1, 512, 16, 16
    x = _nhwc(testing.seeded_tensor("fxg.x", (B, C, H, W))).to(DEV)
    w = testing.seeded_tensor("fxg.w", (C, C, 3, 3), scale=(1.0 / (C * 9)) ** 0.5)
    wp = ops.pack_conv_weight(w.to(DEV), dtype="f32x")
    kw = dict(cout=C, kh=3, kw=3, pad_t=1, pad_l=1)
    key = f"f32x|B{B}|{H}x{W}x{C}|{H}x{W}|N{C}|k3x3s1u0"
    old = ops.PLAN_CACHE.get(key)
    ops.PLAN_CACHE[key] = (64, 128, 8)
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ref = ops.conv2d_nhwc(x, wp, None, **kw).clone()        # eager warm-up: creates the layer's counters
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = ops.conv2d_nhwc(x, wp, None, **kw)
        for _ in range(10):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, ref)
    finally:
        if old is None:
            ops.PLAN_CACHE.pop(key, None)
        else:
            ops.PLAN_CACHE[key] = old


# ---- GroupNorm statistics as order-independent accumulators (sgam_conv_desc.stats_acc, ops.StatsArena) --------------------------
def _acc_sums(acc, B):
    a = acc.view(B, ops.STATS_R, 32, 4).sum(1).cpu().double()      # the replicas add as integers
    return torch.stack([a[..., 0] * 2.0 ** -8 + a[..., 1] * 2.0 ** -40, a[..., 2] * 2.0 ** -8 + a[..., 3] * 2.0 ** -40], -1)


@pytest.mark.parametrize("dt", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("B,C,H,W,plan", [(1, 128, 64, 64, None), (2, 256, 32, 32, None), (1, 512, 16, 16, (64, 128, 8)),
                                          (1, 256, 32, 32, (64, 128, 4)), (3, 128, 32, 48, None)])
def test_statistics_accumulators_equal_the_chunk_records(dt, B, C, H, W, plan, monkeypatch):
    """producer: the accumulator record holds the same sums as the fold of the chunk records (to 2^-40 per partial); consumer:
    a GroupNorm-fusing convolution fed the record equals the one fed the folded {mean, rstd} table; both are run-to-run
    identical although the atomics arrive in any order"""
    ops.set_f32_mode("split")
    monkeypatch.setattr(ops, "STATS_ACC", True)          # (opt-in in the binding: SGAM_STATS_ACC=1)
    tdt = ops.DTYPES[dt]
    cast = (lambda t: t) if dt == "f32" else (lambda t: ops.cast(t, tdt))
    x = cast(_nhwc(testing.seeded_tensor("acc.x", (B, C, H, W), 1.0, 0.3)).to(DEV))
    w1 = testing.seeded_tensor("acc.w1", (C, C, 3, 3), scale=(1.0 / (C * 9)) ** 0.5).to(DEV)
    w2 = testing.seeded_tensor("acc.w2", (C, C, 3, 3), scale=(1.0 / (C * 9)) ** 0.5).to(DEV)
    res = cast(_nhwc(testing.seeded_tensor("acc.r", (B, C, H, W))).to(DEV))
    g = (1 + 0.1 * testing.seeded_tensor("acc.g", (C,))).to(DEV)
    bt = (0.1 * testing.seeded_tensor("acc.b", (C,))).to(DEV)
    wdt = "f32x" if dt == "f32" else tdt
    p1, p2 = ops.pack_conv_weight(w1, dtype=wdt), ops.pack_conv_weight(w2, dtype=wdt)
    if dt != "f32":
        p1._sgam_frag_src, p2._sgam_frag_src = w1, w2
    kw = dict(cout=C, kh=3, kw=3, pad_t=1, pad_l=1)
    key = f"{'f32x' if dt == 'f32' else str(tdt).replace('torch.', '')}|B{B}|{H}x{W}x{C}|{H}x{W}|N{C}|k3x3s1u0"
    old = ops.PLAN_CACHE.get(key)
    if plan is not None:
        ops.PLAN_CACHE[key] = plan
    try:
        ref = ops.conv2d_nhwc(x, p1, None, residual=res, **kw)
        assert hasattr(ref, "_gn_partials") and ref._gn_partials[1] > 0
        sums_ref = ref._gn_partials[0].view(B, ref._gn_partials[1], 32, 2).double().sum(1).cpu()
        ref2 = ops.conv2d_nhwc(ref, p2, None, norm=(g, bt, True, 32, 1e-6), **kw)
        arena = ops.StatsArena(x.device, B)
        with ops.stats_arena(arena):
            got = ops.conv2d_nhwc(x, p1, None, residual=res, **kw)
            assert got._gn_partials[1] == 0 and got._gn_partials[0].dtype == torch.int64
            recs, _ = ops.kernel_timeline(lambda: ops.conv2d_nhwc(got, p2, None, norm=(g, bt, True, 32, 1e-6), **kw))
            got2 = ops.conv2d_nhwc(got, p2, None, norm=(g, bt, True, 32, 1e-6), **kw)
        assert torch.equal(got, ref)
        assert not any("gn_finalize" in r[0] for r in recs), [r[0] for r in recs]       # no fold launch before the consumer
        sums = _acc_sums(got._gn_partials[0], B)
        assert torch.allclose(sums, sums_ref, rtol=1e-11, atol=1e-9), (sums - sums_ref).abs().max()
        scale = ref2.float().abs().max().item()
        assert (got2.float() - ref2.float()).abs().max().item() <= (3e-6 if dt == "f32" else 1e-2) * scale
        first = None
        for _ in range(20):
            with ops.stats_arena(arena):
                a = ops.conv2d_nhwc(x, p1, None, residual=res, **kw)
                rec = a._gn_partials[0].clone()
                b2 = ops.conv2d_nhwc(a, p2, None, norm=(g, bt, True, 32, 1e-6), **kw)
            if first is None:
                first = (rec, b2.clone())
            assert torch.equal(rec, first[0]) and torch.equal(b2, first[1])
        # a consumer that wants the finished table takes the record through the fold entry point (nchunk = 0)
        st = ops.groupnorm_meanrstd(got).cpu().double()
        n = H * W * (C // 32)
        mean = sums[..., 0] / n
        var = (sums[..., 1] / n - mean * mean).clamp_min(0)
        assert torch.allclose(st[..., 0], mean, rtol=0, atol=1e-6) and torch.allclose(st[..., 1], (var + 1e-6).rsqrt(), rtol=2e-6, atol=0)
    finally:
        if old is None:
            ops.PLAN_CACHE.pop(key, None)
        else:
            ops.PLAN_CACHE[key] = old
