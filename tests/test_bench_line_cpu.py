"""The ONE JSON line bench.py prints must stay small enough for the driver to parse (round 3's 20.5 KB line left
`BENCH_r03.json.parsed` null): serialise a full synthetic record — every secondary leg present, top-5 tables with the
longest kernel names of the library — through bench.compact_line and check size, round trip and the contract's keys."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _row(name):
    return {"kernel": name, "calls": 31, "ms": 1.6453, "avg_us": 53.07, "gflop": 1171.72, "tflops": 712.2, "peak": 2500.0,
            "frac": 0.2849, "mfma_busy_frac": 0.2411, "hbm_bytes_per_launch": 43787686, "hbm_gb_per_s": 1651.1}


def _roofline(name):
    return {"bound": "mfma", "achieved": 337.2, "peak": 833.3, "unit": "TFLOP/s", "frac": 0.4046, "traffic": 91300000,
            "kernel": name, "how": "fp32 via exact hi/lo fp16 split: 3x v_mfma_f32_32x32x16_f16, fp32 accumulate",
            "peak_basis": "fp16 dense MFMA 2500 TFLOP/s / 3 MFMA products per fp32 product = 833.3 (builder-defined roof of the exact hi/lo split)",
            "frac_vs_fp32_mfma_peak": 2.1437, "frac_vs_h16_dense_peak": 0.4046, "gflop_per_launch": 19.327,
            "calls_per_frame": 10, "ms_per_frame": 0.5733, "avg_launch_us": 57.33, "gflop_per_frame_in_kernel": 193.27,
            "share_of_kernel_time": 0.2391, "is_top_kernel_by_time": True, "mfma_busy_frac": 0.4451,
            "counters_source": "traffic + mfma_busy_frac: NOT measured in this run — committed rocprofv3 --pmc passes over the same eager "
                               "frame, profiles/r04_pmc_frame_f32.json (libsgam_hip @ 0123456789ab)",
            "counters_commit": "0123456789ab", "counters_stale": False,
            "top5": [_row("conv3x3_f32x_halo2_kernel<64,128,true,false,true>") for _ in range(5)],
            "kernel_time_ms_per_frame": 2.3978, "kernels_per_frame": 212, "bracket_overhead_us": 4.41,
            "frame": {"gflop": 486.4, "ms": 2.923, "tflops": 166.4, "frac": 0.1997}, "method": "x" * 300}


def full_record():
    lock = {"note": "y" * 300}
    for dtn in ("f32", "fp16", "bf16"):
        for S in (4, 8, 16):
            lock[f"{dtn}_S{S}"] = {"scenes": S, "dtype": dtn, "value": 1341.812, "unit": "frames/s (aggregate)", "ms_per_round": 11.924,
                                   "roofline": _roofline("conv3x3_h16_halo_kernel<128,128,0,true,false>")}
    tm = {"note": "z" * 400, "dtype": "fp16", "value": 492.9}
    for dtn in ("fp16", "bf16"):
        tm[dtn] = {"value": 505.644, "unit": "frames/s", "ms_per_step": 1.978, "index_agreement_vs_f32_path": 0.98828,
                   "halo128_frac": 0.2659, "halo128_avg_us": 26.52, "roofline": _roofline("conv3x3_h16_halo_kernel<64,128,0,true,false>")}
    warp_cases = {k: {"B": 16, "N": 3, "H": 512, "W": 512, "algorithmic_bytes": 272629760, "us": 123.45, "kernels_us": {"a": 1.0, "b": 2.0},
                      "achieved": 2208.4, "frac": 0.276, "frac_of_achievable": 0.3505}
                  for k in ("config3_256_N3", "config5_512_B4_N2", "lockstep_256_S16_N3", "large_512_B16_N3")}
    return {
        "metric": "generated RGB-D frames/sec (256x256, GoogleEarth)", "value": 342.123, "unit": "frames/s", "n_gpus": 1, "steps": 20,
        "warmup": 5, "ms_per_step": 2.923, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "GoogleEarth-Infinite 256x256 inference loop (BASELINE configs[2]): forward-splat warp (N<=3) + VQGAN "
                               "encode/quantise(4096)/decode + frame feedback, in-HBM frame store", "frames_per_gpu": 20, "scenes": 1,
                   "parallelism": "scene-parallel x1", "weights": "seeded synthetic (68 990 620 params)", "topk": 1,
                   "launch": "hip-graph replay of the VQGAN forward",
                   "f32_products": "exact hi/lo fp16 split on the fp16 matrix cores, fp32 accumulate"},
        "vqgan_tflops_wallclock": 166.4, "roofline": _roofline("conv3x3_f32x_halo2_kernel<128,128,true>"),
        "cpu_baseline": {"value": 2.5612345, "unit": "frames/s", "cores": 32, "kind": "port",
                         "cpu": "AMD EPYC 9575F 64-Core Processor (256 hardware threads visible)",
                         "pinned_cpus": "0-31 (32 physical cores of one package, one thread each)",
                         "sample": "6 frames of the same 256x256 GoogleEarth step (oracle: C splat + torch-CPU fp32 VQGAN)"},
        "roofline_warp": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "achievable": 6300.0, "bytes_model": "w" * 100, "cases": warp_cases,
                          "inverse_warp": warp_cases},
        "f32_mfma_mode": {"value": 150.012, "unit": "frames/s", "ms_per_step": 6.67, "note": "n" * 120}, "numa_node": 0,
        "f32x_range_flag": 0, "rgbd_integration_branch": {"value": 307.1, "unit": "frames/s", "ms_per_step": 3.26, "note": "n" * 300, **{
            tag: {"value": 307.1, "unit": "frames/s", "ms_per_step": 3.256, "tsdf_bricks_allocated": 27474, "bricks_touched_per_step": 11586,
                  "sources": 3, "scene": "s" * 200,
                  "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "algorithmic_bytes": 383844352, "bytes_model": "b" * 80,
                               "us": 474.6, "achieved": 808.8, "frac": 0.1011,
                               "kernels_us": {"tsdf_touch_kernel": 24.1, "tsdf_integrate_kernel": 254.1, "tsdf_raycast_kernel": 190.9,
                                              "inverse_warp_kernel": 5.4, "depth_normalise_kernel": 2.1}}} for tag in ("noise", "plane")}},
        "concurrent_scenes": {"scenes_on_this_gpu": 4, "value": 447.0, "unit": "frames/s (aggregate)", "ms_per_round": 8.9, "note": "n" * 200},
        "lockstep_scenes": lock, "throughput_mode": tm,
        "config5_512sq_batch4": {"workload": "w" * 80, "note": "n" * 100, "f32": {"ms_per_batch": 34.2, "candidates_per_s": 117.0},
                                 "fp16": {"ms_per_batch": 17.4, "candidates_per_s": 229.9}},
        "training_step": {"ms_per_update": 37.1, "updates_per_s": 26.9, "batch": 1, "rec_loss_first": 0.5, "rec_loss_after_4": 0.49, "note": "n" * 300},
        "frame_checksums": [123456789.0], "head": "0123456789ab", "lib_digest": "ba6838f22271", "extra": "bench_extra.json",
    }


def test_bench_line_is_small_and_round_trips():
    import bench
    full = full_record()
    assert len(json.dumps(full)) > 15000          # the record the line is cut from really is the big one
    text = bench.compact_line(full)
    assert "\n" not in text
    assert len(text) < 4096, len(text)
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["workload"] and "model" not in line["config"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "calls_per_frame", "peak_basis",
              "frac_vs_fp32_mfma_peak", "counters_source", "frame"):
        assert k in r, k
    assert "top5" not in r and "method" not in r
    c = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c)
    assert line["roofline_warp"]["cases"]["large_512_B16_N3"]["achieved"] == 2208.4
    assert line["secondary"]["lockstep_bf16_S8_fps"] == 1341.812 and line["secondary"]["bf16_halo128_frac"] == 0.2659
    # the reference's default CLI branch beside `value`, with its HBM roofline (both scenes)
    assert line["value_rgbd_branch"] == 307.1 and line["secondary"]["rgbd_branch_plane_fps"] == 307.1
    assert line["roofline_rgbd"]["noise"] == 0.1011 and line["roofline_rgbd"]["bound"] == "hbm"
    # the build stamp travels in the library (no git on the driver's box); stale counters are flagged
    assert line["head"] == "0123456789ab" and line["lib_digest"] == "ba6838f22271" and r["counters_stale"] is False


def test_bench_line_without_secondary_legs():
    import bench
    full = full_record()
    for k in ("f32_mfma_mode", "rgbd_integration_branch", "concurrent_scenes", "lockstep_scenes", "throughput_mode",
              "config5_512sq_batch4", "training_step", "roofline_warp"):
        full[k] = None
    full["roofline"] = None
    full["cpu_baseline"] = None
    line = json.loads(bench.compact_line(full))
    assert line["roofline"] is None and line["cpu_baseline"] is None and line["secondary"] == {}


def test_bench_line_degrades_instead_of_overflowing(monkeypatch):
    """a record that would overflow the budget loses its optional objects, never the contract's keys"""
    import bench
    full = full_record()
    full["lockstep_scenes"] = {f"f32_S{i}": {"value": float(i)} for i in range(400)}
    text = bench.compact_line(full)
    assert len(text) <= bench.LINE_BUDGET
    line = json.loads(text)
    assert "secondary" not in line and "roofline" in line and "cpu_baseline" in line


def test_committed_in_frame_counters_are_one_collection():
    """bench.py quotes `roofline.traffic` / `mfma_busy_frac` from committed rocprofv3 --pmc files (profiles/pmc_index.json ->
    "in_frame"): every mode's file must exist, carry its own `collected_at` stamp and the SAME one — counters of one mode from an
    older build than the others (round 4 served round-3 fp16 counters) are stale evidence — and the index must name that stamp."""
    idx = json.load(open(os.path.join(ROOT, "profiles", "pmc_index.json")))
    stamps = {}
    for mode, fn in idx["in_frame"].items():
        path = os.path.join(ROOT, "profiles", fn)
        assert os.path.exists(path), fn
        doc = json.load(open(path))
        assert doc.get("collected_at", "").startswith("library @ "), (fn, doc.get("collected_at"))
        assert doc["kernels"], fn
        stamps[mode] = doc["collected_at"]
    assert set(stamps) == {"f32", "bf16", "fp16"} and len(set(stamps.values())) == 1, stamps
    assert idx["collected_at"] == next(iter(stamps.values()))


def test_cpu_leg_pins_distinct_physical_cores():
    """`cpu_baseline` binds its threads to distinct physical cores of one package (the leg drifted 2.5 -> 1.4 frames/s between rounds
    unpinned); the helper must pick from the cores this process may use and restore nothing it did not set"""
    import bench
    before = os.sched_getaffinity(0)
    try:
        cpus = bench.pin_cpu_leg(2)
        assert cpus is not None and 1 <= len(cpus) <= 2 and set(cpus) <= before and os.sched_getaffinity(0) == set(cpus)
        cores = set()
        for c in cpus:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/core_id") as f:
                cores.add(int(f.read()))
        assert len(cores) == len(cpus)
    finally:
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), before)
            except OSError:
                pass
