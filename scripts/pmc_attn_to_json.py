#!/usr/bin/env python
"""Fold the rocprofv3 --pmc CSVs of `CMD="python scripts/attn_time.py 4096 fused" scripts/pmc_conv.sh` into one JSON for
attn_flash_f32x_kernel (per-launch averages).   python scripts/pmc_attn_to_json.py gpurun_out/pmc profiles/r01h_pmc_attn_flash <us>"""
import collections
import csv
import glob
import json
import sys

src, dst, us = sys.argv[1], sys.argv[2], float(sys.argv[3])
kname = "attn_flash_f32x_kernel"
counters, n = {}, 0
for f in sorted(glob.glob(src + "/*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        counters[c] = sum(v) / len(v)
        n = len(v)
n_tok, C, nsplit = 4096, 256, 8
fetch, write = counters.get("FETCH_SIZE", 0) * 1024, counters.get("WRITE_SIZE", 0) * 1024
flop = 4.0 * n_tok * n_tok * C
alg = n_tok * C * 4 * nsplit + 2 * n_tok * C * 4 + nsplit * n_tok * C * 4     # q per key range, K / V fragments, partial O
xcd_cycles = counters.get("GRBM_GUI_ACTIVE", 0) / 8.0
d = {"kernel": f"{kname}: n = 4096 tokens, C = 256, 8 key ranges (256 workgroups of 4 wavefronts, one per SIMD)",
     "command": 'CMD="python scripts/attn_time.py 4096 fused" scripts/pmc_conv.sh (rocprofv3 --kernel-trace --pmc <group>; one '
                "pass per counter group)",
     "counters": counters, "launches_averaged": n,
     "derived": {"us_per_launch": us, "fp32_flop_per_launch": flop, "tflops_fp32_equiv": flop / us / 1e6,
                 "frac_of_833": flop / us / 1e6 / 833.3,
                 "fetch_bytes_gfx950_corrected_x2": 2 * fetch, "write_bytes": write,
                 "hbm_traffic_bytes_per_launch": 2 * fetch + write, "algorithmic_bytes_per_launch": alg,
                 "mfma_busy_frac": (counters.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 4 / 256 / xcd_cycles) if xcd_cycles else None,
                 "effective_clock_ghz": (xcd_cycles / (us * 1e3)) if xcd_cycles else None,
                 "lds_bank_conflict_frac": (counters.get("SQ_LDS_BANK_CONFLICT", 0) / counters["SQ_LDS_IDX_ACTIVE"])
                 if counters.get("SQ_LDS_IDX_ACTIVE") else None}}
json.dump(d, open(dst + ".json", "w"), indent=1)
print(json.dumps(d["derived"], indent=1))
