"""Launch-plan autotuner for the implicit-GEMM convolutions / GEMMs of the VQGAN.

For every distinct (dtype, shape) the model launches, time each legal (tile, split-K) candidate of
sgam_conv2d_nhwc_{f32,h16} on the GPU and keep the fastest; the table is stored in
``tuned_plans_gfx950.json`` next to this file and applied transparently by ``ops`` (descriptor fields
plan_bm / plan_bn / plan_ksplit).  Results are unaffected up to fp32 summation order (split-K changes the
order of the K reduction, never the set of products).

    python -m sgam_neurips22_amd.tune [--dtypes f32,fp16] [--out path]
"""
import argparse
import ctypes
import json
import os
import sys

import torch

from . import _lib, ops, testing
from ._lib import ConvDesc

TILES = [(128, 128), (64, 128), (64, 64), (32, 32)]


def _parse(key):
    dt, b, ishape, oshape, n, rest = key.split("|")
    Hi, Wi, Cin = map(int, ishape.split("x"))
    Ho, Wo = map(int, oshape.split("x"))
    k, su = rest[1:].split("s")
    KH, KW = map(int, k.split("x"))
    stride, ups = map(int, su.split("u"))
    return dt, int(b[1:]), Hi, Wi, Cin, Ho, Wo, int(n[1:]), KH, KW, stride, ups


def _time(desc, x, w, out, reps=20):
    lib = _lib.load()
    h16 = x.dtype in ops.H16
    split = isinstance(w, ops.SplitWeight)
    nb0 = (lib.sgam_conv2d_f32x_workspace_bytes if split else
           (lib.sgam_conv2d_h16_workspace_bytes if h16 else lib.sgam_conv2d_workspace_bytes))(ctypes.byref(desc))
    ws0 = torch.empty((max(nb0, 16),), device=x.device, dtype=torch.uint8) if nb0 >= 0 else None
    # 16-bit 3x3 convs run on the halo-staged kernel when the plan allows it (no split-K): time what would actually run
    wfrag = hws = None
    hnb = 0
    if h16 and lib.sgam_conv2d_h16_uses_halo(ctypes.byref(desc)) == 1:
        wfrag = (torch.randn((desc.N // 32, desc.ldb // 32, 128, 8), device=x.device) * 0.03).to(x.dtype)
        hnb = lib.sgam_conv2d_halo_h16_workspace_bytes(ctypes.byref(desc))
        hws = torch.empty((max(hnb, 16),), device=x.device, dtype=torch.uint8)

    def run():
        nb, ws = nb0, ws0
        if nb < 0:
            return False
        if split:
            rc = lib.sgam_conv2d_nhwc_f32x(ctypes.byref(desc), ops._p(x), 1.0, ops._p(w.planes), float(w.scale), None, None,
                                           ops._p(out), ops._p(ws), nb, ops._stream())
        elif h16 and wfrag is not None:
            rc = lib.sgam_conv2d_halo_nhwc_h16(ctypes.byref(desc), ops.H16[x.dtype], ops._p(x), None, None, None, 0, ops._p(wfrag),
                                               None, None, ops._p(out), 0, None, ops._p(hws), hnb, ops._stream())
        elif h16:
            rc = lib.sgam_conv2d_nhwc_h16(ctypes.byref(desc), ops.H16[x.dtype], ops._p(x), ops._p(w), None, None, ops._p(out),
                                          0, ops._p(ws), nb, ops._stream())
        else:
            rc = lib.sgam_conv2d_nhwc_f32(ctypes.byref(desc), ops._p(x), ops._p(w), None, None, ops._p(out), ops._p(ws), nb,
                                          ops._stream())
        return rc == 0

    if not run():
        return None
    run()
    torch.cuda.synchronize()
    # time GPU execution, not host launch rate: replay the reps from a captured HIP graph
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        for _ in range(reps):
            run()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def tune_shape(key, verbose=True):
    dt, B, Hi, Wi, Cin, Ho, Wo, N, KH, KW, stride, ups = _parse(key)
    split = dt == "f32x"
    dtype = ops.DTYPES["f32" if dt in ("float32", "f32x") else ("bf16" if dt == "bfloat16" else "fp16")]
    dev = "cuda"
    x = testing.seeded_tensor("tune.x", (B * Hi * Wi, Cin)).to(dev).to(dtype)
    K = KH * KW * Cin
    w = (testing.seeded_tensor("tune.w", (N, K)) * 0.03).to(dev).to(dtype)
    if split:
        w = ops.split_rows(w, 1024.0)
    out = torch.empty((B * Ho * Wo, N), device=dev, dtype=dtype)
    pad = (KH // 2) if stride == 1 else 0
    base = dict(B=B, Hi=Hi, Wi=Wi, Cin=Cin, Ho=Ho, Wo=Wo, N=N, KH=KH, KW=KW, stride=stride, pad_t=pad, pad_l=pad,
                upsample2x=ups, lda=Cin, ldb=K, ldc=N, ldr=0, n_valid=N, bias_per_row=0)
    slab = 64 if dtype in ops.H16 else 32
    iters = KH * KW * ((Cin + slab - 1) // slab)
    M = B * Ho * Wo
    results = []
    # ResnetBlock / conv_out: GroupNorm(+swish) first (maps of <= 1024 pixels normalise in one launch and are never fused)
    norm_input = KH == 3 and stride == 1 and not ups and Cin >= 64 and Hi * Wi > 1024
    auto = ConvDesc(**base, plan_bm=0, plan_bn=0, plan_ksplit=0)
    t_auto = _time(auto, x, w, out)
    if t_auto is not None and split and norm_input and _lib.load().sgam_conv2d_f32x_gn_fusable(ctypes.byref(auto)) != 1:
        t_auto += max(4e-3, 2.0 * B * Hi * Wi * Cin * 4 / 4e12 * 1e3)
    # the 16-bit halo kernel also has a 256-row tile (16 x 16 patch, one workgroup per CU): a candidate where many tile waves run
    tiles = list(TILES)
    if dtype in ops.H16 and KH == 3 and stride == 1 and not ups:
        tiles.append((256, 128))
    if split and KH == 3 and stride == 1 and not ups and Ho % 16 == 0 and Wo % 16 == 0 and M <= 4096 and Cin >= 64:
        tiles.append((256, 32))          # weight-stationary kernel of the small maps (always split-K)
    for bm, bn in tiles:
        if bn == 128 and N % 128:
            continue
        blocks = -(-M // bm) * -(-N // bn)
        cands = {1}
        for target in (256, 512, 768, 1024):
            ks = max(1, round(target / blocks))
            # (back-to-back timing keeps the split-K workspace L2-hot; in a frame it is cold: deep splits that win here
            # lose there, measured — cap them)
            if ks <= iters // 2 and ks <= 16:
                cands.add(ks)
        if (bm, bn) == (256, 32):
            slabs = Cin // 32
            cands = {k for k in (slabs, slabs // 2, slabs // 4) if k >= 2}
        for ks in sorted(cands):
            if blocks * ks > 8192:
                continue
            cand = ConvDesc(**base, plan_bm=bm, plan_bn=bn, plan_ksplit=ks)
            t = _time(cand, x, w, out)
            if t is not None:
                # a 3x3 conv behind a GroupNorm: plans that run the halo-staged kernel also absorb the normalise pass
                # (a read + a write of the activation at ~4 TB/s effective, >= 4 us for a launch of its own)
                if split and norm_input and _lib.load().sgam_conv2d_f32x_gn_fusable(ctypes.byref(cand)) != 1:
                    t += max(4e-3, 2.0 * B * Hi * Wi * Cin * 4 / 4e12 * 1e3)
                results.append((t, bm, bn, ks))
    results.sort()
    if not results:                      # every candidate grid past the launch limit (the largest batches): keep the heuristic
        if verbose:
            print(f"{key:70s} no candidate plan fits: heuristic kept", flush=True)
        return None, t_auto, (t_auto if t_auto is not None else 0.0, 0, 0, 0)
    best = results[0]
    if verbose:
        gf = 2.0 * M * N * K / 1e9
        print(f"{key:70s} auto {t_auto * 1e3:7.1f} us -> best {best[0] * 1e3:7.1f} us {best[1:]}  "
              f"({gf / best[0] / 1e3:6.1f} TFLOP/s)", flush=True)
    CANDIDATES[key] = results[:4]
    # keep the heuristic unless the tuned plan wins by >3% (run-to-run noise)
    if t_auto is not None and best[0] > 0.97 * t_auto:
        return None, t_auto, best
    return best[1:], t_auto, best


CANDIDATES = {}      # key -> the four fastest (time, bm, bn, ks) of the isolated timing, for refine_in_frame


def refine_in_frame(plans, dataset="google_earth", res=256, frames=12, verbose=True):
    """Second pass: the isolated timing of tune_shape replays one layer back to back (its split-K workspace and
    weights stay cache-hot, the clock settles for that kernel alone).  Here every shape's runner-up plans are tried
    INSIDE the real frame — the whole scene loop is re-captured and timed with the candidate in place — and a
    candidate is kept only if the frame gets faster.  Greedy, one shape at a time, most expensive shapes first."""
    import time

    from .config import default_params
    from .generative_sensing_module.model import VQModel
    from .inference_pipeline import InfiniteSceneGeneration, synthetic_seed_frame
    m = VQModel(**default_params(dataset))
    m.load_state_dict(testing.synthetic_state_dict(m.state_dict(), seed=0))
    m = m.to("cuda").eval()
    seed = synthetic_seed_frame(dataset, 0, res)

    def frame_ms():
        m.enable_hip_graph(False)
        m.enable_hip_graph(True)                     # drop the captured graphs: plans are baked into them
        sc = InfiniteSceneGeneration(m, dataset, output_dim=(frames + 6, 1), seed_frame=seed)
        best = 1e9
        for rep in range(2):
            for _ in range(3 if rep == 0 else 0):
                sc.one_step_prediction(sc.next_pose(sc.curr)); sc.curr += 1
            torch.cuda.synchronize()
            t = time.perf_counter()
            n = frames // 2
            for _ in range(n):
                sc.one_step_prediction(sc.next_pose(sc.curr)); sc.curr += 1
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) / n * 1e3)
        return best

    def install(p):
        ops.PLAN_CACHE.clear()
        ops.PLAN_CACHE.update({k: tuple(v) for k, v in p.items()})

    plans = dict(plans)
    install(plans)
    cur = frame_ms()
    if verbose:
        print(f"in-frame refinement: start {cur:.3f} ms/frame", flush=True)
    order = sorted(CANDIDATES, key=lambda k: -CANDIDATES[k][0][0])
    for key in order:
        for t, bm, bn, ks in CANDIDATES[key][:3]:
            if plans.get(key) == [bm, bn, ks] or plans.get(key) == (bm, bn, ks):
                continue
            trial = dict(plans)
            trial[key] = [bm, bn, ks]
            install(trial)
            ms = frame_ms()
            if ms < cur * 0.995:
                if verbose:
                    print(f"  {key}: {plans.get(key)} -> {[bm, bn, ks]}  {cur:.3f} -> {ms:.3f} ms/frame", flush=True)
                plans, cur = trial, ms
    install(plans)
    if verbose:
        print(f"in-frame refinement: end {cur:.3f} ms/frame", flush=True)
    return plans


def collect_shapes(dtypes, dataset="google_earth", res=256, batch=1):
    from .config import default_params
    from .generative_sensing_module.model import VQModel
    keys = {}
    m = VQModel(**default_params(dataset))
    m.load_state_dict(testing.synthetic_state_dict(m.state_dict(), seed=0))
    m = m.to("cuda").eval()
    x, mask = testing.rect_hole_input(batch, res, res)
    for dt in dtypes:
        m.set_compute_dtype(dt)
        ops.PLAN_RECORD = {}
        with torch.no_grad():
            m(x.cuda(), topk=1 if batch == 1 else None, extrapolation_mask=mask.cuda())
        keys.update(ops.PLAN_RECORD)
        ops.PLAN_RECORD = None
    return sorted(keys)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtypes", default="f32,fp16,bf16", help="f32 tunes the current ops.F32_MODE (split by default)")
    ap.add_argument("--merge", action="store_true", help="keep the plans already in the output file for other dtypes")
    ap.add_argument("--out", default=ops._PLAN_FILE)
    ap.add_argument("--configs", default="256x1", help="comma list of <resolution>x<batch> model inputs whose layer shapes are "
                                                       "tuned: 256x1 = BASELINE configs 1-4, 512x4 = config 5")
    ap.add_argument("--refine", action="store_true", help="after the isolated timing, re-judge the runner-up plans inside the "
                                                          "real frame (f32 only)")
    a = ap.parse_args()
    os.environ["SGAM_NO_TUNED_PLANS"] = "1"
    ops.load_plans()
    keys = sorted({k for cfg in a.configs.split(",")
                   for k in collect_shapes(a.dtypes.split(","), res=int(cfg.split("x")[0]), batch=int(cfg.split("x")[1]))})
    print(f"{len(keys)} distinct conv/GEMM shapes", flush=True)
    plans, saved = {}, 0.0
    for k in keys:
        pl, t_auto, best = tune_shape(k)
        if pl:
            plans[k] = list(pl)
            saved += (t_auto - best[0])
    print(f"tuned {len(plans)} of {len(keys)} shapes; sum of per-shape savings {saved * 1e3:.0f} us (one launch each)")
    if a.refine:
        os.environ.pop("SGAM_NO_TUNED_PLANS", None)
        plans = refine_in_frame(plans)
    if a.merge and os.path.exists(ops._PLAN_FILE):
        old = json.load(open(ops._PLAN_FILE)).get("plans", {})
        for k in keys:                 # every shape just examined: the new verdict (plan or heuristic) replaces the old one
            old.pop(k, None)
        old.update(plans)
        plans = old
    with open(a.out, "w") as f:
        json.dump({"device": torch.cuda.get_device_name(0), "plans": plans}, f, indent=0, sort_keys=True)
    print("wrote", a.out)


if __name__ == "__main__":
    sys.exit(main())
