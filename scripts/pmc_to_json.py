#!/usr/bin/env python
"""Fold the rocprofv3 --pmc CSVs of scripts/pmc_conv.sh into one JSON (per-launch averages for one kernel) and
copy the CSVs next to it.   python scripts/pmc_to_json.py gpurun_out/pmc conv_gemm_f32x_kernel profiles/r01c_pmc_conv128_f32x <us_per_launch>"""
import collections
import csv
import glob
import json
import shutil
import sys

src, kname, dst, us = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
# optional: B,Cin,Cout,H,W,k of the layer (default: the 128 -> 128 3x3 layer on the 256 x 256 map), element bytes of the
# activations (4 = fp32 split path, 2 = 16-bit), and the timeline name under which bench.py looks the entry up
shape = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "1,128,128,256,256,3").split(",")]
ebytes = int(sys.argv[6]) if len(sys.argv) > 6 else 4
tl_name = sys.argv[7] if len(sys.argv) > 7 else None
counters, n = {}, 0
for f in sorted(glob.glob(src + "/*/**/*counter_collection.csv", recursive=True)):
    group = f.split(src.rstrip("/") + "/")[1].split("/")[0]
    agg = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(f)) if kname in r["Kernel_Name"]]
    for r in rows:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        counters[c] = sum(v) / len(v)
        n = len(v)
    keep = [r for r in rows]
    if keep:
        with open(f"{dst}_{group}.csv", "w", newline="") as o:
            w = csv.DictWriter(o, fieldnames=list(keep[0].keys()))
            w.writeheader()
            w.writerows(keep)
Bs, Cin_, Cout_, H_, W_, k_ = shape
M, N, K = Bs * H_ * W_, Cout_, k_ * k_ * Cin_
fetch = counters.get("FETCH_SIZE", 0) * 1024      # FETCH_SIZE / WRITE_SIZE are reported in KiB
write = counters.get("WRITE_SIZE", 0) * 1024
alg = M * Cin_ * ebytes + M * N * ebytes + N * K * (4 if ebytes == 4 else 2)     # input + output + weights (hi / lo fp16 halves = 4 B)
xcd_cycles = counters.get("GRBM_GUI_ACTIVE", 0) / 8.0
d = {"kernel": f"{kname} on the layer B={Bs}, {Cin_}->{Cout_}, {k_}x{k_}, {H_}x{W_}: M={M} N={N} K={K}",
     "timeline_name": tl_name,
     "command": "scripts/pmc_conv.sh (rocprofv3 --kernel-trace --pmc <group> -- python scripts/conv_micro.py "
                f"--shape {','.join(map(str, shape))} --reps 10 [--norm]; one pass per counter group)",
     "counters": counters, "launches_averaged": n,
     "derived": {"fetch_bytes_raw": fetch, "fetch_bytes_gfx950_corrected_x2": 2 * fetch, "write_bytes": write,
                 "hbm_traffic_bytes_per_launch": 2 * fetch + write, "algorithmic_bytes_per_launch": alg,
                 "us_per_launch": us,
                 "mfma_busy_frac": (counters.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 4 / 256 / xcd_cycles) if xcd_cycles else None,
                 "effective_clock_ghz": (xcd_cycles / (us * 1e3)) if xcd_cycles else None,
                 "lds_bank_conflict_frac": (counters.get("SQ_LDS_BANK_CONFLICT", 0) / counters["SQ_LDS_IDX_ACTIVE"])
                 if counters.get("SQ_LDS_IDX_ACTIVE") else None,
                 "l2_hit_rate": (counters["TCC_HIT_sum"] / (counters["TCC_HIT_sum"] + counters["TCC_MISS_sum"]))
                 if counters.get("TCC_HIT_sum") else None,
                 "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for "
                         "16-B/lane loads)."}}
json.dump(d, open(dst + ".json", "w"), indent=1)
if tl_name:     # bench.py's roofline.traffic comes from this index, keyed by the kernel name of its timeline
    import os
    idx_path = os.path.join(os.path.dirname(dst), "pmc_index.json")
    idx = json.load(open(idx_path)) if os.path.exists(idx_path) else {}
    idx[tl_name] = {"hbm_traffic_bytes_per_launch": round(2 * fetch + write), "algorithmic_bytes_per_launch": alg,
                    "mfma_busy_frac": d["derived"]["mfma_busy_frac"], "us_per_launch": us, "file": os.path.basename(dst) + ".json",
                    "note": f"bytes/launch on {d['kernel']}: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE from separate rocprofv3 "
                            f"--pmc passes; algorithmic {alg} B; in-kernel MFMA pipe busy "
                            f"{d['derived']['mfma_busy_frac']:.3f}"}
    json.dump(idx, open(idx_path, "w"), indent=1, sort_keys=True)
print(json.dumps(d["derived"], indent=1))
