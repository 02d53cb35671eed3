#!/usr/bin/env python
"""Greedy IN-FRAME search over the launch plans of one arithmetic mode: every conv / GEMM shape of the B = 1 frame, most expensive
first, is tried under alternative (tile rows, tile columns, split-K) plans with the whole scene loop re-captured and timed; a plan
is kept only if the frame gets faster by more than the noise and a second measurement confirms it.  (Back-to-back tuner timings
keep weights and split-K workspaces cache-hot and lost to in-frame judged plans three times — DESIGN.md 5.5 / 5.5b; this is the
in-frame judge alone, for all three modes.)
    python scripts/plan_search.py [f32|bf16|fp16] [frames=16] [min_gain=0.004]        -> gpurun_out/plan_search_<mode>.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402

from sgam_neurips22_amd import ops, testing  # noqa: E402
from sgam_neurips22_amd.config import default_params  # noqa: E402
from sgam_neurips22_amd.generative_sensing_module.model import VQModel  # noqa: E402
from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration, synthetic_seed_frame  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16
min_gain = float(sys.argv[3]) if len(sys.argv) > 3 else 0.004
p = default_params("google_earth")
m = VQModel(**p)
sd = testing.synthetic_state_dict(m.state_dict(), seed=0)
sd["quantize.embedding.weight"] = testing.codebook_from_stats(0.0, 0.5, p["n_embed"], 256, 1)
m.load_state_dict(sd)
m = m.to("cuda").eval()
m.set_compute_dtype(mode)
seed = synthetic_seed_frame("google_earth", 0, 256)


def frame_ms(reps=2):
    m.enable_hip_graph(False)
    m.enable_hip_graph(True)                     # drop the captured graphs: plans are baked into them
    sc = InfiniteSceneGeneration(m, "google_earth", output_dim=(reps * frames + 10, 1), seed_frame=seed)
    for _ in range(4):
        sc.one_step_prediction(sc.next_pose(sc.curr)); sc.curr += 1
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(frames):
            sc.one_step_prediction(sc.next_pose(sc.curr)); sc.curr += 1
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / frames * 1e3)
    return best


# the shapes of this mode's frame
ops.PLAN_RECORD = {}
x, mask = testing.rect_hole_input(1, 256, 256)
with torch.no_grad(), m.eager():
    m(x.cuda(), topk=1, extrapolation_mask=mask.cuda())
keys = sorted(ops.PLAN_RECORD)
ops.PLAN_RECORD = None
plans = dict(ops.PLAN_CACHE)
base = frame_ms(3)
print(f"{mode}: {len(keys)} shapes, start {base:.4f} ms/frame = {1e3 / base:.1f} frames/s", flush=True)


def candidates(key):
    dt, b, ishape, oshape, n, rest = key.split("|")
    Hi, Wi, Cin = map(int, ishape.split("x"))
    Ho, Wo = map(int, oshape.split("x"))
    N = int(n[1:])
    k3 = rest.startswith("k3x3")
    M = Ho * Wo
    slabs = max(1, (9 if k3 else 1) * Cin // 32)
    out = []
    for bm, bn in ((128, 128), (64, 128), (64, 64)):
        if bn == 128 and N % 128:
            continue
        blocks = -(-M // bm) * -(-N // bn)
        if blocks > 4096:
            continue
        for ks in (1, 2, 4, 8, 16):
            if ks > 1 and (blocks * ks > 1024 or ks > max(1, (Cin // 32))):
                continue
            out.append((bm, bn, ks))
    return out


changed = {}
cur = base
for key in keys:
    # the 3x3 / stride 1 convolutions carry the frame; SGAM_PLAN_SEARCH=<substring> searches the shapes whose key contains it instead
    # (e.g. "k3x3s2": the four Downsample convolutions on the generic kernel)
    if os.environ.get("SGAM_PLAN_SEARCH", "|k3x3s1") not in key:
        continue
    best_pl, best_ms = None, cur
    for pl in candidates(key):
        if tuple(plans.get(key, ())) == pl:
            continue
        ops.PLAN_CACHE.clear(); ops.PLAN_CACHE.update(plans); ops.PLAN_CACHE[key] = pl
        try:
            ms = frame_ms(1)
        except Exception as e:                   # a plan the kernels refuse
            print(f"   {key} {pl}: {type(e).__name__}", flush=True)
            continue
        if ms < best_ms * (1 - min_gain):
            ms2 = frame_ms(2)                    # confirm
            if ms2 < best_ms * (1 - min_gain):
                best_pl, best_ms = pl, ms2
    if best_pl is not None:
        print(f"  {key}: {plans.get(key)} -> {best_pl}   {cur:.4f} -> {best_ms:.4f} ms/frame", flush=True)
        plans[key] = best_pl
        changed[key] = list(best_pl)
        cur = best_ms
ops.PLAN_CACHE.clear(); ops.PLAN_CACHE.update(plans)
end = frame_ms(3)
print(f"{mode}: end {end:.4f} ms/frame = {1e3 / end:.1f} frames/s ({len(changed)} plans changed)", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
tag = os.environ.get("SGAM_PLAN_SEARCH", "").replace("|", "")
json.dump({"mode": mode, "start_ms": base, "end_ms": end, "changed": changed}, open("gpurun_out/plan_search_" + mode + ("_" + tag if tag else "") + ".json", "w"), indent=1)
