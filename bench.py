#!/usr/bin/env python
"""bench.py — generated RGB-D frames/s of the SGAM per-step generative-sensing hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): GoogleEarth-Infinite, 256x256,
frame-autoregressive loop with the forward-splat conditioning warp (N <= 3 nearest visited frames) ->
conditional VQGAN encode / quantise (4096 codes) / decode -> frame feedback (uint8 RGB truncation + depth
de-normalisation) into the in-HBM frame store.  One "step" = one generated frame; weights are seeded synthetic
(checkpoints are not fetchable), the seed frame is synthetic.  N GPUs run N independent scenes (weak scaling,
no data-path collective; one RCCL all-gather of the per-rank record at the end).

    python bench.py --gpus 1 --steps 31 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no torchrun environment launches the N ranks itself (same command line the
driver uses), one process per GPU, each pinned to its GPU's NUMA node.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` — the kernel with the most GPU time in
a frame, found at run time from the library's own per-kernel timeline (HIP events recorded on the launch stream around
EVERY kernel of one eager frame, include/sgam_hip.h sgam_prof_*; same names and durations as `rocprofv3 --kernel-trace
--stats`), its top-5 table and the frame-level fraction — and `cpu_baseline` (the oracle = CPU port of the same step,
timed on the host cores, rank 0 / N=1 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # one hardware queue per concurrent scene (see distributed.py)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from sgam_neurips22_amd import distributed as sdist  # noqa: E402
from sgam_neurips22_amd import ops, testing  # noqa: E402
from sgam_neurips22_amd.config import default_params  # noqa: E402
from sgam_neurips22_amd.generative_sensing_module.model import VQModel  # noqa: E402
from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration, synthetic_seed_frame  # noqa: E402

DATASET = "google_earth"
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 256 CUs x 2.4 GHz
H16_MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/fp16 MFMA peak (AMD's 5 PF figure includes 2:1 sparsity)
GFLOP_PER_FRAME = 486.4         # VQGAN at 256x256, B=1


def build_model(device):
    import contextlib
    p = default_params(DATASET)
    with contextlib.redirect_stdout(sys.stderr):      # the constructor prints "Working with z of shape ..." like the reference's
        m = VQModel(**p)
    sd = testing.synthetic_state_dict(m.state_dict(), seed=0)
    sd["quantize.embedding.weight"] = testing.codebook_from_stats(0.0, 0.5, p["n_embed"], 256, 1)
    m.load_state_dict(sd)
    return m.to(device).eval(), sd, p


def kernel_peak(name):
    """(matrix-pipe roof in TFLOP/s of ALGORITHMIC flops, how) for the kernel families that run on MFMA; None otherwise"""
    if "f32x" in name:     # 3 fp16 MFMAs per fp32 product
        return round(H16_MFMA_PEAK_TFLOPS / 3.0, 1), "fp32 via exact hi/lo fp16 split: 3x v_mfma_f32_32x32x16_f16, fp32 accumulate"
    if "h16" in name and ("conv" in name or "flash" in name):
        return H16_MFMA_PEAK_TFLOPS, "16-bit operands, v_mfma_f32_32x32x16, fp32 accumulate"
    if "conv_gemm_f32" in name:
        return FP32_MFMA_PEAK_TFLOPS, "fp32-in v_mfma_f32_32x32x2_f32"
    return None, None


def peak_basis(name):
    if "f32x" in name:
        return "fp16 dense MFMA 2500 TFLOP/s / 3 MFMA products per fp32 product = 833.3 (builder-defined roof of the exact hi/lo split)"
    if "h16" in name:
        return "bf16/fp16 dense MFMA 2500 TFLOP/s"
    return "fp32-in MFMA 157.3 TFLOP/s"


def frame_timeline(scene):
    """One extra (untimed) EAGER frame under the library's kernel timeline.  Returns {kernel: {calls, ms, gflop, gbyte}}
    (durations net of the bracket's own cost) and that cost in ms."""
    model = scene.dynamic_model

    def one():
        with model.eager():
            scene.one_step_prediction(scene.next_pose(scene.curr))
        scene.curr += 1

    one()                                        # eager warm-up (allocator pools, first-use paths)
    recs, bracket = ops.kernel_timeline(one)
    if os.environ.get("SGAM_DUMP_TIMELINE"):      # every launch of the frame, in order: kernel, us (net), GEMM view
        with open(os.environ["SGAM_DUMP_TIMELINE"], "w") as f:
            for name, ms, flops, nbytes, shp in recs:
                f.write(f"{name}\t{1e3 * max(ms - bracket, 0):.2f}\t{flops / 1e9:.3f}\t{shp[0]}x{shp[1]}x{shp[2]}/{shp[3]}\n")
    agg = {}
    for name, ms, flops, nbytes, _shp in recs:
        a = agg.setdefault(name, {"calls": 0, "ms": 0.0, "gflop": 0.0, "gbyte": 0.0})
        a["calls"] += 1
        a["ms"] += max(ms - bracket, 0.0)
        a["gflop"] += flops / 1e9
        a["gbyte"] += nbytes / 1e9
    return agg, bracket


def roofline_from_timeline(agg, bracket_ms, ms_per_step):
    rows = []
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        peak, how = kernel_peak(name)
        tf = a["gflop"] / a["ms"] if a["ms"] > 0 and a["gflop"] > 0 else None      # GFLOP / ms = TFLOP/s
        rows.append({"kernel": name, "calls": a["calls"], "ms": round(a["ms"], 4), "avg_us": round(1e3 * a["ms"] / a["calls"], 2),
                     "gflop": round(a["gflop"], 2), "tflops": None if tf is None else round(tf, 1), "peak": peak,
                     "frac": None if (tf is None or peak is None) else round(tf / peak, 4), "how": how})
    total_ms = sum(r["ms"] for r in rows)
    dom = rows[0]
    # the roofline entry is about an MFMA kernel; if (some day) the top row is not one, the first MFMA row is named too
    top_mfma = next((r for r in rows if r["peak"] is not None), dom)
    peak = top_mfma["peak"]
    out = {"bound": "mfma", "achieved": top_mfma["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": top_mfma["frac"],
           "traffic": None, "kernel": top_mfma["kernel"], "how": top_mfma["how"],
           "peak_basis": peak_basis(top_mfma["kernel"]),
           "frac_vs_fp32_mfma_peak": None if top_mfma["tflops"] is None else round(top_mfma["tflops"] / FP32_MFMA_PEAK_TFLOPS, 4),
           "frac_vs_h16_dense_peak": None if top_mfma["tflops"] is None else round(
               top_mfma["tflops"] * (3.0 if "f32x" in top_mfma["kernel"] else 1.0) / H16_MFMA_PEAK_TFLOPS, 4),
           "gflop_per_launch": round(top_mfma["gflop"] / max(top_mfma["calls"], 1), 3),
           "calls_per_frame": top_mfma["calls"], "ms_per_frame": top_mfma["ms"], "avg_launch_us": top_mfma["avg_us"],
           "gflop_per_frame_in_kernel": top_mfma["gflop"],
           "share_of_kernel_time": round(top_mfma["ms"] / total_ms, 4),
           "is_top_kernel_by_time": top_mfma is dom,
           "top5": [{k: r[k] for k in ("kernel", "calls", "ms", "avg_us", "gflop", "tflops", "peak", "frac")} for r in rows[:5]],
           "kernel_time_ms_per_frame": round(total_ms, 4), "kernels_per_frame": sum(r["calls"] for r in rows),
           "bracket_overhead_us": round(bracket_ms * 1e3, 2),
           "frame": {"gflop": GFLOP_PER_FRAME, "ms": ms_per_step, "tflops": round(GFLOP_PER_FRAME / ms_per_step, 1),
                     "frac": round(GFLOP_PER_FRAME / ms_per_step / peak, 4) if peak else None},
           "method": "HIP events on the launch stream around every kernel of one eager frame (sgam_prof_*), GPU parked "
                     "behind a spin while the host enqueues, bracket cost (median of 32 empty brackets) subtracted"}
    return out, rows


def timed_loop(step_fn, warmup, steps):
    """`warmup` untimed then `steps` timed calls of step_fn(), device-synchronised on both sides; seconds of the timed part"""
    for _ in range(warmup):
        step_fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t


_PMC_FRAME = {}
_PMC_COMMIT = [None]    # "library @ <commit>" the committed counter files were collected on (profiles/pmc_index.json)
_PMC_DIGEST = [None]    # the library digest (sgam_build_digest) of that build, when the file carries one (round 6 on)


def pmc_frame_entry(mode, timeline_name):
    """the in-frame counter record of a kernel (profiles/<pmc_index['in_frame'][mode]>, written by scripts/pmc_frame.sh +
    pmc_frame.py from rocprofv3 --pmc passes over this same eager frame): the timeline spells a kernel as its launch site
    does (`k<128,128,true>`), rocprofv3 with the defaulted template arguments (`k<128,128,true,false>`)"""
    idx_path = os.path.join(ROOT, "profiles", "pmc_index.json")
    if mode not in _PMC_FRAME:
        _PMC_FRAME[mode] = (None, {})
        if os.path.exists(idx_path):
            idx = json.load(open(idx_path))
            fn = idx.get("in_frame", {}).get(mode)
            _PMC_COMMIT[0] = idx.get("collected_at")
            if fn and os.path.exists(os.path.join(ROOT, "profiles", fn)):
                doc = json.load(open(os.path.join(ROOT, "profiles", fn)))
                _PMC_FRAME[mode] = (fn, doc["kernels"])
                _PMC_COMMIT[0] = doc.get("collected_at") or _PMC_COMMIT[0]      # the file's own stamp wins over the index's
                _PMC_DIGEST[0] = doc.get("lib_digest")
    fn, kernels = _PMC_FRAME[mode]
    if timeline_name in kernels:
        return fn, kernels[timeline_name]
    if timeline_name.endswith(">"):
        base, args = timeline_name[:-1].split("<", 1)
        have = args.split(",")
        cands = [k for k in kernels if k.startswith(timeline_name[:-1] + ",")]
        # the omitted trailing template arguments are the defaulted ones: `false` everywhere in this library except the 16-bit halo
        # kernel's tail <..., SW = true, GNF = false, NB = 3>
        tail = _TEMPLATE_DEFAULT_TAIL.get(base)
        if tail is not None:
            full = f"{base}<{','.join(have + tail[len(have) - (_TEMPLATE_ARITY[base] - len(tail)):])}>"
            if full in kernels:
                return fn, kernels[full]
        dflt = [k for k in cands if set(k[len(timeline_name):-1].split(",")) <= {"false"}]
        if len(dflt) == 1 or len(cands) == 1:
            return fn, kernels[(dflt or cands)[0]]
    return fn, None


_TEMPLATE_ARITY = {"conv3x3_h16_halo_kernel": 8}
_TEMPLATE_DEFAULT_TAIL = {"conv3x3_h16_halo_kernel": ["true", "false", "3"]}


def attach_counters(roofline, mode):
    """per top-5 row: in-kernel MFMA-pipe busy fraction and HBM GB/s (counter bytes per launch / this run's average launch
    time) from the committed in-frame PMC file; `roofline.traffic` = the named kernel's counter bytes per launch"""
    if roofline is None:
        return
    for row in [roofline] + roofline.get("top5", []):
        name = row.get("kernel")
        fn, ent = pmc_frame_entry(mode, name) if name else (None, None)
        if ent is None:
            continue
        us = row.get("avg_us", row.get("avg_launch_us"))
        row["mfma_busy_frac"] = ent["mfma_busy_frac"]
        row["hbm_bytes_per_launch"] = ent["hbm_traffic_bytes_per_launch"]
        row["hbm_gb_per_s"] = round(ent["hbm_traffic_bytes_per_launch"] / (us * 1e3), 1) if (us and ent["hbm_traffic_bytes_per_launch"]) else None
    fn, ent = pmc_frame_entry(mode, roofline["kernel"])
    if ent is not None:
        roofline["traffic"] = ent["hbm_traffic_bytes_per_launch"]
        commit, digest = lib_stamp()
        cc = (_PMC_COMMIT[0] or "").replace("library @ ", "").strip()
        roofline["counters_commit"] = cc or None       # the build the committed counters were collected on; equals `head` when fresh
        # stale = the counters were collected on another build of the library than the one benched (digest when the file has
        # one, else the commit stamp)
        roofline["counters_stale"] = (_PMC_DIGEST[0] != digest) if _PMC_DIGEST[0] else (cc != commit)
        roofline["counters_source"] = (f"traffic + mfma_busy_frac: NOT measured in this run — committed rocprofv3 --pmc passes over the same "
                                       f"eager frame, profiles/{fn}" + (f" ({_PMC_COMMIT[0]})" if _PMC_COMMIT[0] else ""))
        roofline["traffic_note"] = (f"HBM bytes per launch of {roofline['kernel']}, averaged over its {ent['launches_per_frame']} in-frame "
                                    f"launches: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate rocprofv3 --pmc passes over "
                                    f"the eager bench frame (profiles/{fn})")


def pin_cpu_leg(n):
    """Bind EVERY thread of this process to `n` distinct physical cores (one hardware thread each) of one package, chosen from the
    cores the process may use: the CPU leg drifted 2.5 -> 1.4 frames/s across rounds on the same host model because the 32 OpenMP
    threads were free to share SMT siblings or straddle sockets.  Threads that exist already (torch's intra-op pool may) are bound
    one by one through /proc/self/task; threads created afterwards inherit the mask.  Returns the cpu list (None: not pinned)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        by_pkg = {}
        for c in allowed:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            with open(base + "physical_package_id") as f:
                pkg = int(f.read())
            with open(base + "core_id") as f:
                core = int(f.read())
            by_pkg.setdefault(pkg, {}).setdefault(core, c)          # first hardware thread of every core
        pkg = max(by_pkg, key=lambda k: (len(by_pkg[k]), -k))       # the package with most usable cores (lowest id on a tie)
        cpus = sorted(by_pkg[pkg].values())[:n]
        if not cpus:
            return None
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), cpus)
            except OSError:
                pass
        return cpus
    except (OSError, AttributeError, ValueError):
        return None


def cpu_baseline(sd, p, seed_frame, n_frames):
    """The oracle (CPU port of the same step: C forward splat + torch-CPU VQGAN + host feedback codec) on the
    host cores.  Reported baseline only."""
    from oracle import vqgan as OV
    from oracle import warp as OW
    from sgam_neurips22_amd.inference_pipeline import intrinsics
    # all 256 hardware threads of the GPU box oversubscribe torch-CPU badly (measured 92 s/frame); 32 is the
    # sweet spot on that host.  `cores` reports what was actually used, `pinned_cpus` where they ran.
    nthr = min(32, os.cpu_count() or 1)
    pinned = pin_cpu_leg(nthr)
    if pinned:
        nthr = len(pinned)
    torch.set_num_threads(nthr)
    K = intrinsics(DATASET).astype(np.float32)
    lut = (np.arange(256, dtype=np.float64) / 127.5 - 1.0).astype(np.float32)
    rgb = lut[seed_frame[0]].transpose(2, 0, 1)
    depth = seed_frame[1]
    T = np.eye(4, dtype=np.float32)
    T[1, 3] = -0.0594   # one grid step, like the pipeline's relative pose
    t0 = time.perf_counter()
    done = 0
    for _ in range(n_frames):
        if done and time.perf_counter() - t0 > 20.0:   # bounded sample: ~10-30 s of CPU work
            break
        done += 1
        w = OW.forward_splat(rgb[None, None], depth[None, None], K[None], K[None, None], T[None, None])
        nd = OW.normalise_depth(torch.from_numpy(w["merge_depths"]), torch.from_numpy(w["extrapolation_mask"]), DATASET)
        x = torch.cat([torch.from_numpy(w["merge_feats"]), nd], 1)
        o = OV.forward(sd, p["ddconfig"], x, torch.from_numpy(w["extrapolation_mask"]), topk=1)
        dec = o["dec"][0][0]
        rgb = lut[OW.rgb_to_uint8(dec[0, :3])].transpose(2, 0, 1)
        depth = OW.denormalise_depth(dec[0, 3], DATASET).numpy()
    dt = time.perf_counter() - t0
    n_frames = done
    cpu_name = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_name = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": n_frames / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu": f"{cpu_name} ({os.cpu_count()} hardware threads visible)",
            "pinned_cpus": (f"{pinned[0]}-{pinned[-1]} ({len(pinned)} physical cores of one package, one thread each)" if pinned else None),
            "sample": f"{n_frames} frames of the same 256x256 GoogleEarth step (oracle: C splat + torch-CPU fp32 VQGAN)"}


HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E, 8 stacks
HBM_ACHIEVABLE_GBS = 6300.0    # same guide: best sustained streaming rate measured on the part


def warp_roofline(dev, reps=8):
    """HBM roofline of the two conditioning warps (forward splat + median + mask = H*W*(16N+17) bytes per target
    frame, inverse warp H*W*(16N+16)), measured with the library's per-kernel HIP-event brackets over `reps` launches on
    synthetic sources read in place through the pointer table, exactly as the scene loop calls them.  Cases: config 3
    (256x256, N = 3), config 5 (512x512, B = 4 candidates, N = 2), lock step S = 16 (48 sources in one launch) and one
    deliberately large launch (512x512, B = 16, N = 3) where launch latency no longer hides the kernel."""
    cases = [("config3_256_N3", 1, 3, 256), ("config5_512_B4_N2", 4, 2, 512), ("lockstep_256_S16_N3", 16, 3, 256),
             ("large_512_B16_N3", 16, 3, 512)]
    out = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achievable": HBM_ACHIEVABLE_GBS,
           "bytes_model": "forward splat H*W*(16N+17)*B, inverse warp H*W*(16N+16)*B (SURVEY 8d)", "cases": {}, "inverse_warp": {}}
    from sgam_neurips22_amd.inference_pipeline import intrinsics
    for tag, B, N, res in cases:
        # the scene loop's own geometry at this size: smooth seeded RGB-D frames (synthetic_seed_frame: depth in the GoogleEarth
        # template range), sources one / two / three grid steps (0.0594) away from the target along the trajectory, GoogleEarth
        # intrinsics.  (A white-noise depth map would scatter every source bin over +- 20 pixels of parallax: a stress case the
        # parity tests cover, not the workload.)
        Kn = intrinsics(DATASET, (res, res)).astype(np.float32)
        feats, depths, Tn = [], [], np.tile(np.eye(4, dtype=np.float32), (B * N, 1, 1))
        lut = (np.arange(256, dtype=np.float64) / 127.5 - 1.0).astype(np.float32)
        for b in range(B):
            for n in range(N):
                rgb8, dep = synthetic_seed_frame(DATASET, seed_index=3 * b + n, res=res)
                feats.append(torch.from_numpy(np.ascontiguousarray(lut[rgb8])).to(dev))
                depths.append(torch.from_numpy(dep).to(dev))
                Tn[b * N + n, 1, 3] = -0.0594 * (n + 1)
        Ks = np.tile(Kn, (B, N, 1, 1))
        K = torch.from_numpy(Ks[:, 0].copy()).to(dev)
        Kinv = torch.inverse(torch.from_numpy(Ks).reshape(-1, 3, 3)).to(dev)
        Td = torch.from_numpy(Tn).to(dev)
        bufs = {"x": torch.empty((B, 4, res, res), device=dev), "extrap": torch.empty((B, 1, res, res), device=dev, dtype=torch.bool),
                "winner": torch.empty((B, res * res), device=dev, dtype=torch.int32)}

        def splat():
            ops.forward_splat_srcs(feats, depths, K, Kinv, Td, B=B, dataset=DATASET, want=("x", "extrap"), extrap_bool=True, out=bufs)

        def timed(tiled):
            old_mode, ops.SPLAT_TILED = ops.SPLAT_TILED, tiled
            try:
                splat()
                recs, br = ops.kernel_timeline(lambda: [splat() for _ in range(reps)])
            finally:
                ops.SPLAT_TILED = old_mode
            per = {}
            for name, ms, *_ in recs:
                per[name.split("<")[0]] = per.get(name.split("<")[0], 0.0) + max(ms - br, 0.0)
            return {k: round(1e3 * v / reps, 2) for k, v in per.items()}
        # both forms: target-owned LDS z-tiles (no global atomics) and the two-pass device-scope atomicMax form (whose winner-buffer
        # memset the timeline does not see); `us` is the one ops.forward_splat_srcs picks for this size
        us_tiled, us_two = timed(True), timed(False)
        picked_tiled = ops.SPLAT_TILED if ops.SPLAT_TILED is not None else (B * N * res * res >= ops.SPLAT_TILED_MIN_POINTS)
        us = us_tiled if picked_tiled else us_two
        tot_us = sum(us.values())
        nbytes = res * res * (16 * N + 17) * B
        out["cases"][tag] = {"B": B, "N": N, "H": res, "W": res, "algorithmic_bytes": nbytes, "us": round(tot_us, 2), "kernels_us": us,
                             "form": "tiled" if picked_tiled else "two_pass",
                             "tiled_us": round(sum(us_tiled.values()), 2), "two_pass_global_atomics_us": round(sum(us_two.values()), 2),
                             "achieved": round(nbytes / tot_us / 1e3, 1), "frac": round(nbytes / tot_us / 1e3 / HBM_PEAK_GBS, 4),
                             "frac_of_achievable": round(nbytes / tot_us / 1e3 / HBM_ACHIEVABLE_GBS, 4)}
        # inverse warp at the same geometry (target depth = the first source's depth: any finite depth does for timing)
        tgt_depth = torch.stack([depths[b * N] for b in range(B)])
        outw = torch.empty((B, 3, res, res), device=dev)
        Ksd, Kinv_t = torch.from_numpy(Ks).reshape(-1, 3, 3).to(dev), torch.inverse(K)

        def inv():
            ops.inverse_warp_srcs(feats, depths, tgt_depth, Ksd, Kinv_t, Td, B=B, out=outw)
        inv()
        recs, br = ops.kernel_timeline(lambda: [inv() for _ in range(reps)])
        t_us = 1e3 * sum(max(ms - br, 0.0) for name, ms, *_ in recs if name.startswith("inverse_warp")) / reps
        nb2 = res * res * (16 * N + 16) * B
        out["inverse_warp"][tag] = {"algorithmic_bytes": nb2, "us": round(t_us, 2), "achieved": round(nb2 / t_us / 1e3, 1),
                                    "frac": round(nb2 / t_us / 1e3 / HBM_PEAK_GBS, 4)}
        del feats, depths, bufs
    return out


class PlaneDepthScene(InfiniteSceneGeneration):
    """the rgbd_integration branch on CONSISTENT geometry: every frame's depth map (the seed's included) is the view-space depth
    of the world plane z = 0 — GoogleEarth's ground under the tilted camera, 1.9 .. 3.1 m away like the templates — at that
    frame's pose, in place of the noise a randomly initialised decoder generates; colours stay the model's.  What the TSDF
    kernels cost on the kind of surface a trained model's frames describe (a thin band of bricks)."""

    def plane_depth(self, coord):
        node = self.transform_grid[coord[0]][coord[1]]
        H, W = self.image_resolution
        vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
        rays = np.stack([(uu - self.K[0, 2]) / self.K[0, 0], (vv - self.K[1, 2]) / self.K[1, 1], np.ones_like(uu)], -1)
        dz = rays @ node["T_inv"][2, :3]                   # world z of the ray direction (camera -> world rotation, row z)
        oz = node["T_inv"][2, 3]                            # camera height
        with np.errstate(divide="ignore", invalid="ignore"):
            s_hit = np.where(dz < 0, -oz / dz, 0.0)        # view-space z of the hit (ray direction has z_cam = 1)
        return torch.from_numpy(np.clip(s_hit, 0.0, 4.7).astype(np.float32)).to(self.device)

    def prepare_planes(self):
        self._planes = {c: self.plane_depth(c) for c in self._ordered_grid_coords}
        self.frames[(0, 0)]["depth"] = self._planes[(0, 0)]

    def save_to_store(self, coord, rgb_u8, rgb_f, depth):
        super().save_to_store(coord, rgb_u8, rgb_f, self._planes[coord])


def rgbd_branch_legs(model, scene_id, seed_frame, args):
    """BASELINE config 3, branch (B): the rgbd_integration conditioning path — TSDF fusion of the source frames, depth ray cast
    at the target pose, target-depth-driven inverse warp — in front of the same VQGAN + feedback.  Two scenes: the frames the
    randomly initialised model generates (noise depths: every unit of the frustum opens, the worst case) and consistent
    geometry (PlaneDepthScene).  Each with the loop's rate, the per-kernel durations of the conditioning launches of one eager
    step (library timeline) and their HBM roofline: algorithmic bytes = bricks touched x 32 KB + H W (16 N + 16)."""
    out = {"note": "conditioning = TSDF integrate (<= 3 source frames, one pass) + depth ray cast + inverse warp (csrc/tsdf.hip, warp.hip)"}
    H = W = 256
    for tag, cls in (("noise", InfiniteSceneGeneration), ("plane", PlaneDepthScene)):
        sc3 = cls(model, DATASET, seed_index=scene_id, output_dim=(args.warmup + args.steps + 6, 1), seed_frame=seed_frame,
                  use_rgbd_integration=True)
        if tag == "plane":
            sc3.prepare_planes()

        def one3():
            sc3.one_step_prediction(sc3.next_pose(sc3.curr)); sc3.curr += 1
        dt3 = timed_loop(one3, args.warmup, args.steps)
        st = sc3.volume.stats()

        def eager3():
            with model.eager():
                one3()
        eager3()
        recs, br = ops.kernel_timeline(eager3)
        n_src = len(sc3.get_src_grid_coords(sc3.next_pose(sc3.curr - 1))[0])
        cond = {}
        for name, ms, *_ in recs:
            base = name.split("<")[0]
            if base.startswith("tsdf_") or base.startswith("inverse_warp") or base.startswith("depth_normalise"):
                cond[base] = cond.get(base, 0.0) + max(ms - br, 0.0)
        us = 1e3 * sum(cond.values())
        bricks = sc3.volume.stats()[1]
        nbytes = bricks * 32768 + H * W * (16 * n_src + 16)
        out[tag] = {"value": round(args.steps / dt3, 3), "unit": "frames/s", "ms_per_step": round(1e3 * dt3 / args.steps, 3),
                    "tsdf_bricks_allocated": st[0], "bricks_touched_per_step": bricks, "sources": n_src,
                    "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes": nbytes,
                                 "bytes_model": "bricks touched x 32 KB (tsdf + weight, fp32) + H W (16 N + 16)", "us": round(us, 1),
                                 "achieved": round(nbytes / us / 1e3, 1) if us else None,
                                 "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4) if us else None,
                                 "kernels_us": {k: round(1e3 * v, 1) for k, v in cond.items()}}}
        del sc3
    out["noise"]["scene"] = ("frames generated by the seeded random weights: noise depths (0 .. 18 m, uncorrelated between neighbouring "
                             "pixels) open every unit of the view frustum — the worst case for the fusion, and what round 5 timed")
    out["plane"]["scene"] = ("every frame's depth = view-space depth of the world plane z = 0 at its pose (consistent geometry: a thin band of bricks)")
    out["value"], out["unit"], out["ms_per_step"] = out["noise"]["value"], "frames/s", out["noise"]["ms_per_step"]
    return out



LINE_BUDGET = 4096      # bytes of the ONE JSON line (the driver's parser choked on the 20 KB line of round 3)


def lib_stamp():
    """(commit, digest) compiled into libsgam_hip.so at build time (csrc/build_info.hip, sgam_neurips22_amd/build.py): the last
    commit that touched the library's sources and a sha256 over sources + flags.  The driver's box has no .git — the stamp
    travels inside the library."""
    from sgam_neurips22_amd import _lib
    lib = _lib.load()
    return lib.sgam_build_commit().decode(), lib.sgam_build_digest().decode()


def compact_line(full):
    """The ONE line of the contract from the full record: headline fields, `roofline` (dominant kernel, both roofs, where the
    counter figures come from), `cpu_baseline`, `roofline_warp` (one number per case) and one number per secondary leg.
    Everything else (top-5 tables, every lock-step leg, throughput-mode tables, config 5, training) lives in the side file."""
    def pick(d, keys):
        return None if d is None else {k: d.get(k) for k in keys if k in d}
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    line = {k: full.get(k) for k in keep}
    cfg = full.get("config") or {}
    line["config"] = {k: cfg.get(k) for k in ("workload", "frames_per_gpu", "scenes", "parallelism", "launch", "f32_products") if k in cfg}
    r = full.get("roofline")
    if r is not None:
        line["roofline"] = pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "calls_per_frame",
                                    "gflop_per_launch", "peak_basis", "frac_vs_fp32_mfma_peak", "frac_vs_h16_dense_peak",
                                    "mfma_busy_frac", "counters_source", "counters_commit", "counters_stale", "share_of_kernel_time", "kernels_per_frame",
                                    "kernel_time_ms_per_frame", "frame"))
    else:
        line["roofline"] = None
    rg = full.get("rgbd_integration_branch")
    if rg is not None:
        # co-headline: the reference's default CLI branch (--use_rgbd_integration, BASELINE configs[2]) beside `value`
        line["value_rgbd_branch"] = rg.get("value")
        line["roofline_rgbd"] = {k: (rg.get(k) or {}).get("roofline", {}).get("frac") for k in ("noise", "plane")}
        line["roofline_rgbd"].update({"bound": "hbm", "unit": "frac of 8 TB/s", "us_noise": (rg.get("noise") or {}).get("roofline", {}).get("us"),
                                      "us_plane": (rg.get("plane") or {}).get("roofline", {}).get("us"),
                                      "bytes": "bricks touched x 32 KB + H W (16 N + 16)"})
    line["cpu_baseline"] = pick(full.get("cpu_baseline"), ("value", "unit", "cores", "kind", "sample", "cpu", "pinned_cpus"))
    w = full.get("roofline_warp")
    if w is not None:
        line["roofline_warp"] = {"bound": "hbm", "unit": "GB/s", "peak": w.get("peak"),
                                 "cases": {k: pick(v, ("achieved", "frac", "us", "algorithmic_bytes")) for k, v in (w.get("cases") or {}).items()}}
    summ = {}
    def put(name, leg, key="value"):
        if leg is not None and leg.get(key) is not None:
            summ[name] = leg[key]
    put("f32_mfma_mode_fps", full.get("f32_mfma_mode"))
    put("rgbd_branch_fps", full.get("rgbd_integration_branch"))
    put("rgbd_branch_plane_fps", (full.get("rgbd_integration_branch") or {}).get("plane"))
    put("concurrent_scenes_fps", full.get("concurrent_scenes"))
    tm = full.get("throughput_mode") or {}
    for dtn in ("fp16", "bf16"):
        if dtn in tm:
            summ[f"{dtn}_fps"] = tm[dtn]["value"]
            rr = tm[dtn].get("roofline") or {}
            summ[f"{dtn}_halo128_frac"] = tm[dtn].get("halo128_frac")
            summ[f"{dtn}_kernels_per_frame"] = rr.get("kernels_per_frame")
    for k, v in (full.get("lockstep_scenes") or {}).items():
        if isinstance(v, dict) and "value" in v:
            summ[f"lockstep_{k}_fps"] = v["value"]
    c5 = full.get("config5_512sq_batch4") or {}
    for dtn in ("f32", "fp16"):
        if dtn in c5:
            summ[f"config5_{dtn}_ms_per_batch"] = c5[dtn]["ms_per_batch"]
    put("training_ms_per_update", full.get("training_step"), "ms_per_update")
    line["secondary"] = {k: v for k, v in summ.items() if v is not None}
    for k in ("f32x_range_flag", "numa_node", "frame_checksums", "head", "lib_digest", "extra"):
        if k in full:
            line[k] = full[k]
    text = json.dumps(line, separators=(",", ":"))
    # belt and braces: never emit a line the driver cannot take — drop the optional objects, largest first
    for drop in ("secondary", "roofline_warp", "frame_checksums"):
        if len(text) <= LINE_BUDGET:
            break
        line.pop(drop, None)
        text = json.dumps(line, separators=(",", ":"))
    return text


def write_extra(full):
    """the full record (every leg, every table) beside the line: ./bench_extra.json and, when the scratch directory of a
    gpurun call exists, gpurun_out/bench_extra.json (copied into profiles/ for the judge)"""
    paths = [os.path.join(ROOT, "bench_extra.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_extra.json"))
    written = []
    for pth in paths:
        try:
            with open(pth, "w") as f:
                json.dump(full, f, indent=1)
            written.append(os.path.relpath(pth, ROOT))
        except OSError:
            pass
    return written


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=31)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-frames", type=int, default=6, help="frames of the CPU oracle baseline (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the fp16 throughput-mode leg")
    ap.add_argument("--f32-mode", default=None, choices=["split", "mfma"],
                    help="fp32 products: fp16-split MFMA (default) or fp32-in MFMA")
    ap.add_argument("--concurrent-scenes", type=int, default=4, help="secondary leg: this many independent trajectories "
                                                                      "on one GPU, one stream each (0/1 = skip)")
    ap.add_argument("--lockstep-scenes", type=lambda v: [int(t) for t in v.split(",") if t], default=[4, 8, 16],
                    help="secondary leg: scenes per GPU advanced in lock step at batch S (comma list; empty = skip)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying HIP graphs")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "fp16"],
                    help="arithmetic of the VQGAN body: f32 = parity path (fp32-in MFMA), bf16/fp16 = 16-bit MFMA path")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: launch the ranks (gloo), shard the scenes, gather "
                                                           "a synthetic record — checks the N > 1 command line end to end")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver calls it for N = 1: launch the N ranks ourselves
        sys.exit(sdist.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if args.f32_mode:
        ops.set_f32_mode(args.f32_mode)
    rank, local_rank, world = sdist.init(backend="gloo" if args.dry_run else None)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    my_scenes = sdist.shard_scenes(world, rank, world)      # N scenes over N ranks: rank r owns scene r (weak scaling)
    if args.dry_run:
        g = sdist.gather_metrics(args.steps * len(my_scenes), 1.0 + 0.01 * rank, float(sum(my_scenes)), "cpu")
        if rank == 0:
            print(json.dumps({"metric": "generated RGB-D frames/sec (256x256, GoogleEarth)", "dry_run": True, "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "value": g["frames_per_s"],
                              "scenes_per_rank": [r[2] for r in g["per_rank"]]}), flush=True)
        sdist.barrier()
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    numa = sdist.pin_to_gpu_numa_node(local_rank)
    dev = torch.device("cuda", local_rank)
    model, sd, p = build_model(dev)
    model.set_compute_dtype(args.dtype)
    model.enable_hip_graph(not args.no_graph)
    ops.range_flag(dev)        # the split path's range guard: verified after the timed loop (`f32x_range_flag` below)
    scene_id = my_scenes[0]
    seed_frame = synthetic_seed_frame(DATASET, seed_index=scene_id)
    n_frames = args.warmup + args.steps + 4
    scene = InfiniteSceneGeneration(model, DATASET, seed_index=scene_id, output_dim=(n_frames + 1, 1), seed_frame=seed_frame)

    for _ in range(args.warmup):
        scene.one_step_prediction(scene.next_pose(scene.curr))
        scene.curr += 1
    torch.cuda.synchronize()
    sdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scene.one_step_prediction(scene.next_pose(scene.curr))
        scene.curr += 1
    torch.cuda.synchronize()
    sdist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    range_tripped = ops.f32x_range_tripped() if args.dtype == "f32" and ops.F32_MODE == "split" else False
    checksum = float(sum(int(f["rgb_u8"].sum()) for f in scene.frames.values()) % (1 << 31))
    g = sdist.gather_metrics(args.steps, dt, checksum, dev, numa_node=numa)
    t_max = g["max_seconds"]

    roofline = None
    if rank == 0 and not args.no_roofline:
        agg, bracket = frame_timeline(scene)
        roofline, _rows = roofline_from_timeline(agg, bracket, 1e3 * t_max / args.steps)
        # HBM traffic / MFMA-pipe occupancy per kernel: in-frame PMC counters (scripts/pmc_frame.sh), committed under profiles/
        attach_counters(roofline, "f32" if args.dtype == "f32" else args.dtype)

    f32_mfma_leg = None
    if rank == 0 and world == 1 and args.dtype == "f32" and ops.F32_MODE == "split" and not args.no_secondary:
        # the same loop with every fp32 product on the fp32-in MFMA (bit-for-bit an fp32 fmaf chain; no range precondition)
        ops.set_f32_mode("mfma")
        model.enable_hip_graph(False)
        model.enable_hip_graph(not args.no_graph)
        scm = InfiniteSceneGeneration(model, DATASET, seed_index=scene_id, output_dim=(args.warmup + args.steps + 2, 1),
                                      seed_frame=seed_frame)
        for _ in range(args.warmup):
            scm.one_step_prediction(scm.next_pose(scm.curr)); scm.curr += 1
        torch.cuda.synchronize()
        tm = time.perf_counter()
        for _ in range(args.steps):
            scm.one_step_prediction(scm.next_pose(scm.curr)); scm.curr += 1
        torch.cuda.synchronize()
        dtm = time.perf_counter() - tm
        f32_mfma_leg = {"value": round(args.steps / dtm, 3), "unit": "frames/s", "ms_per_step": round(1e3 * dtm / args.steps, 3),
                        "note": "--f32-mode mfma: v_mfma_f32_32x32x2_f32 (157.3 TFLOP/s roof), same parity tests"}
        del scm
        ops.set_f32_mode("split")
        model.enable_hip_graph(False)
        model.enable_hip_graph(not args.no_graph)

    rgbd_leg = None
    if rank == 0 and world == 1 and args.dtype == "f32" and not args.no_secondary:
        rgbd_leg = rgbd_branch_legs(model, scene_id, seed_frame, args)

    conc_leg = None
    if rank == 0 and world == 1 and args.dtype == "f32" and not args.no_secondary and args.concurrent_scenes > 1:
        # several independent trajectories on this ONE GPU, one HIP stream + one model instance each: kernels of other
        # scenes fill the launch-latency gaps of a single scene.  Aggregate rate; NOT `value` (BASELINE's N=1 workload is
        # one trajectory per GPU).
        def make_scene(i):
            mi = model if i == 0 else build_model(dev)[0]
            mi.set_compute_dtype("f32")
            mi.enable_hip_graph(not args.no_graph)
            return InfiniteSceneGeneration(mi, DATASET, seed_index=i, output_dim=(args.warmup + args.steps + 2, 1),
                                           seed_frame=seed_frame)
        cs = sdist.ConcurrentScenes(make_scene, args.concurrent_scenes)
        for _ in range(args.warmup):
            cs.step()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        for _ in range(args.steps):
            cs.step()
        torch.cuda.synchronize()
        dt4 = time.perf_counter() - t4
        conc_leg = {"scenes_on_this_gpu": args.concurrent_scenes, "value": round(args.concurrent_scenes * args.steps / dt4, 3),
                    "unit": "frames/s (aggregate)", "ms_per_round": round(1e3 * dt4 / args.steps, 3),
                    "note": "independent trajectories on separate HIP streams of one GPU (sgam_neurips22_amd.distributed."
                            "ConcurrentScenes); each scene's frames are identical to running it alone"}
        del cs

    secondary = None
    if rank == 0 and world == 1 and args.dtype == "f32" and not args.no_secondary:
        # the 16-bit throughput mode on the same workload (16-bit activations / weights, fp32 accumulate): NOT the parity
        # path — reported beside the headline, never as `value`; fp16 and bf16 separately
        secondary = {"note": "16-bit MFMA path (halo-staged 3x3 kernel with fused GroupNorm, csrc/h16_halo.hip; fp32 accumulate); "
                             "agreement-rate mode, never `value`.  `index_agreement_vs_f32_path` is MEASURED in this run: one forward of "
                             "each mode on the same seeded 256x256 input against this backend's fp32 parity path (agreement with "
                             "the REFERENCE's indices on its fixtures: tests/test_gpu_configs.py, gpurun_out/report_*.json)"}
        xa, ma = testing.rect_hole_input(1, 256, 256, seed=3)
        xa, ma = xa.to(dev), ma.to(dev)
        with torch.no_grad(), model.eager():
            idx32 = model(xa, extrapolation_mask=ma, get_codebook_count=True)[2].clone()
        for dtn in ("fp16", "bf16"):
            model.set_compute_dtype(dtn)
            with torch.no_grad(), model.eager():
                agree = float((model(xa, extrapolation_mask=ma, get_codebook_count=True)[2] == idx32).float().mean())
            sc2 = InfiniteSceneGeneration(model, DATASET, seed_index=scene_id, output_dim=(args.warmup + args.steps + 4, 1),
                                          seed_frame=seed_frame)

            def one2():
                sc2.one_step_prediction(sc2.next_pose(sc2.curr)); sc2.curr += 1
            dt2 = timed_loop(one2, args.warmup, args.steps)
            agg2, br2 = frame_timeline(sc2)
            r2, _rows2 = roofline_from_timeline(agg2, br2, 1e3 * dt2 / args.steps)
            attach_counters(r2, dtn)
            h128 = next((r for r in _rows2 if r["kernel"].startswith("conv3x3_h16_halo") and "<128,128" in r["kernel"]), None)
            secondary[dtn] = {"value": round(args.steps / dt2, 3), "unit": "frames/s", "ms_per_step": round(1e3 * dt2 / args.steps, 3),
                              "index_agreement_vs_f32_path": round(agree, 5),
                              "halo128_frac": None if h128 is None else h128["frac"], "halo128_avg_us": None if h128 is None else h128["avg_us"],
                              "roofline": {k: r2.get(k) for k in ("kernel", "achieved", "peak", "frac", "calls_per_frame", "ms_per_frame",
                                                                  "mfma_busy_frac", "traffic", "top5", "frame", "kernel_time_ms_per_frame",
                                                                  "kernels_per_frame")}}
            del sc2
        secondary["dtype"], secondary["value"] = "fp16", secondary["fp16"]["value"]
        model.set_compute_dtype("f32")

    lock_leg = None
    if rank == 0 and world == 1 and args.dtype == "f32" and not args.no_secondary and args.lockstep_scenes:
        # S independent trajectories of this GPU advanced in LOCK STEP: one launch sequence at B = S (one forward splat over
        # S x N source frames, one VQGAN forward, one feedback launch) instead of S streams — what the kernels deliver when
        # they are fed (sgam_neurips22_amd.distributed.LockstepScenes; every scene's frames match its solo run,
        # tests/test_gpu_lockstep.py).  Aggregate frames/s; NOT `value` (BASELINE's N = 1 workload is one trajectory per GPU).
        lock_leg = {"note": "S scenes per GPU through ONE launch sequence at batch S (LockstepScenes); aggregate frames/s = S x steps / "
                            "seconds; `roofline` = the kernel with the most GPU time in one eager lock-stepped step (in-run timeline)"}
        steps_l = max(4, min(args.steps, 12))
        for dtn in ("f32", "fp16", "bf16"):
            model.set_compute_dtype(dtn)
            for S in args.lockstep_scenes:
                seeds = [synthetic_seed_frame(DATASET, seed_index=i) for i in range(S)]
                ls = sdist.LockstepScenes(model, DATASET, seeds, output_dim=(args.warmup + steps_l + 5, 1))
                dtl = timed_loop(ls.step, args.warmup, steps_l)

                def one_l():
                    with model.eager():
                        ls.step()
                one_l()
                recs, br = ops.kernel_timeline(one_l)
                aggl = {}
                for name, ms, flops, nbytes, _shp in recs:
                    a = aggl.setdefault(name, {"calls": 0, "ms": 0.0, "gflop": 0.0, "gbyte": 0.0})
                    a["calls"] += 1; a["ms"] += max(ms - br, 0.0); a["gflop"] += flops / 1e9; a["gbyte"] += nbytes / 1e9
                rl, _ = roofline_from_timeline(aggl, br, 1e3 * dtl / steps_l)
                rl["frame"] = {"gflop": GFLOP_PER_FRAME * S, "ms": round(1e3 * dtl / steps_l, 3),
                               "tflops": round(GFLOP_PER_FRAME * S / (1e3 * dtl / steps_l), 1),
                               "frac": round(GFLOP_PER_FRAME * S / (1e3 * dtl / steps_l) / rl["peak"], 4) if rl["peak"] else None}
                lock_leg[f"{dtn}_S{S}"] = {
                    "scenes": S, "dtype": dtn, "value": round(S * steps_l / dtl, 3), "unit": "frames/s (aggregate)",
                    "ms_per_round": round(1e3 * dtl / steps_l, 3),
                    "roofline": {k: rl.get(k) for k in ("kernel", "achieved", "peak", "frac", "calls_per_frame", "avg_launch_us",
                                                        "share_of_kernel_time", "top5", "frame", "kernel_time_ms_per_frame",
                                                        "kernels_per_frame")}}
                del ls
        model.set_compute_dtype("f32")

    stress = None
    if rank == 0 and world == 1 and args.dtype == "f32" and not args.no_secondary:
        # BASELINE config 5 (a parity-test case, timed here as a side note): 512x512 input, batch of 4 warp candidates
        # through the VQGAN forward (32x32 latent, attention over 16 384 tokens) in both arithmetic modes
        xs, ms = zip(*[testing.rect_hole_input(1, 512, 512, seed=40 + i) for i in range(4)])
        x5, m5 = torch.cat(xs).to(dev), torch.cat(ms).to(dev)
        stress = {"workload": "VQGAN forward, 512x512, batch of 4 candidates (attention over 16384 tokens)",
                  "note": "both modes run the fused single-pass attention (no 16384 x 16384 score matrix)"}
        for dtn in ("f32", "fp16"):
            model.set_compute_dtype(dtn)
            with torch.no_grad():
                for _ in range(2):
                    model(x5, extrapolation_mask=m5)
                torch.cuda.synchronize()
                t5 = time.perf_counter()
                for _ in range(4):
                    model(x5, extrapolation_mask=m5)
                torch.cuda.synchronize()
                d5 = (time.perf_counter() - t5) / 4
            stress[dtn] = {"ms_per_batch": round(1e3 * d5, 2), "candidates_per_s": round(4 / d5, 1)}
        model.set_compute_dtype("f32")
        del x5, m5

    train_leg = None
    if rank == 0 and world == 1 and args.dtype == "f32" and not args.no_secondary:
        # SURVEY §8 f4 (a side note like the other legs): the reference's training step on the same 256x256 model, past
        # disc_start — forward with tape, L1 + codebook + adaptive-weighted generator loss, backward, Adam on the encoder (the
        # phase that trains the hot-path model) with the LPIPS term on (synthetic VGG16 trunk), then the PatchGAN's hinge loss,
        # backward and Adam; a second instance of the model, so the inference legs are not disturbed
        from sgam_neurips22_amd import training
        mt = build_model(dev)[0]
        xt, mk = testing.rect_hole_input(1, 256, 256, seed=9)
        xd = testing.seeded_tensor("bench.train.dst", (1, 4, 256, 256), scale=0.5).clamp(-1, 1).to(dev)
        xt, mk = xt.to(dev), mk.to(dev)
        from sgam_neurips22_amd.generative_sensing_module.modules.losses.vqperceptual import VQLPIPSWithDiscriminator
        lcfg = VQLPIPSWithDiscriminator(disc_start=0, perceptual_weight=1.0, disc_in_channels=4, disc_weight=0.8,
                                        use_discriminative_loss=True).to(dev).train()      # trained_models/*/config.yaml lossconfig
        lcfg.perceptual_loss.load_state_dict({k: v.to(dev) for k, v in testing.synthetic_vgg_state_dict(
            lcfg.perceptual_loss.state_dict(), seed=4).items()})        # (the ImageNet trunk cannot be fetched: synthetic)
        tr = training.VQGANTrainer(mt, lcfg, phase="conditional_generation", lr=4.5e-6)
        l0 = tr.step(xt, xd, mk)[1]["train/rec_loss"]
        torch.cuda.synchronize()
        tt = time.perf_counter()
        for _ in range(3):
            l1 = tr.step(xt, xd, mk)[1]["train/rec_loss"]
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - tt) / 3
        train_leg = {"ms_per_update": round(1e3 * dt_, 1), "updates_per_s": round(1 / dt_, 2), "batch": 1,
                     "rec_loss_first": round(float(l0), 6), "rec_loss_after_4": round(float(l1), 6),
                     "note": "f4: autoencoder (encoder parameter set) + PatchGAN discriminator updates of training_step past disc_start, "
                             "LPIPS on (synthetic VGG16 trunk); fp32-in MFMA GEMMs + csrc/train.hip; untuned"}
        del mt, tr

    warp_leg = None
    if rank == 0 and world == 1 and not args.no_roofline:
        warp_leg = warp_roofline(dev)

    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        cpu = cpu_baseline({k: v.cpu() for k, v in sd.items()}, p, seed_frame, args.cpu_frames)

    if rank == 0:
        full = {
            "metric": "generated RGB-D frames/sec (256x256, GoogleEarth)", "value": round(g["total_frames"] / t_max, 3),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * t_max / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "GoogleEarth-Infinite 256x256 inference loop (BASELINE configs[2]): forward-splat warp (N<=3) + VQGAN "
                                   "encode/quantise(4096)/decode + frame feedback, in-HBM frame store",
                       "frames_per_gpu": args.steps, "scenes": world, "parallelism": f"scene-parallel x{world}",
                       "weights": "seeded synthetic (68 990 620 params)", "topk": 1,
                       "launch": "eager" if args.no_graph else "hip-graph replay of the VQGAN forward",
                       "f32_products": ("exact hi/lo fp16 split on the fp16 matrix cores, fp32 accumulate"
                                        if ops.F32_MODE == "split" else "fp32-in MFMA") if args.dtype == "f32" else None},
            "vqgan_tflops_wallclock": round(GFLOP_PER_FRAME * g["total_frames"] / t_max / 1e3 / world, 2),
            "roofline": roofline, "cpu_baseline": cpu, "roofline_warp": warp_leg, "f32_mfma_mode": f32_mfma_leg, "numa_node": numa,
            "f32x_range_flag": int(range_tripped), "rgbd_integration_branch": rgbd_leg, "concurrent_scenes": conc_leg,
            "lockstep_scenes": lock_leg, "throughput_mode": secondary, "config5_512sq_batch4": stress, "training_step": train_leg,
            "frame_checksums": [r[2] for r in g["per_rank"]], "head": lib_stamp()[0], "lib_digest": lib_stamp()[1],
            "per_rank": g.get("per_rank_records"), "rccl": g.get("rccl"),
        }
        written = write_extra(full)
        full["extra"] = written[0] if written else None
        sys.stdout.flush()
        print(compact_line(full), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()          # ranks leave together (rank 0 was still profiling)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
