#!/usr/bin/env python
"""Per-launch averages of every counter rocprofv3 --pmc collected for the kernels whose name contains <substr>.
   python scripts/pmc_kernel_counters.py gpurun_out/pmc attn_flash [out.json]"""
import collections
import csv
import glob
import json
import sys

src, sub = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for f in sorted(glob.glob(src + "/*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            res[k][c] = sum(v) / len(v)
        res[k]["launches_averaged"] = len(next(iter(cs.values())))
for k, c in res.items():
    w = c.get("SQ_WAVES", 0)
    if w and "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        c["derived"] = {
            "cycles_per_xcd": cyc,
            "wave_lifetime_cycles": 4 * c.get("SQ_WAVE_CYCLES", 0) / w,
            "active_inst_any_cycles_per_wave": 4 * c.get("SQ_ACTIVE_INST_ANY", 0) / w,
            "wait_inst_any_cycles_per_wave": 4 * c.get("SQ_WAIT_INST_ANY", 0) / w,
            "wait_any_cycles_per_wave": 4 * c.get("SQ_WAIT_ANY", 0) / w,
            "valu_insts_per_wave": c.get("SQ_INSTS_VALU", 0) / w, "mfma_insts_per_wave": c.get("SQ_INSTS_MFMA", 0) / w,
            "lds_insts_per_wave": c.get("SQ_INSTS_LDS", 0) / w, "salu_insts_per_wave": c.get("SQ_INSTS_SALU", 0) / w,
            "mfma_busy_frac": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (w / 1.0) / cyc if cyc else None,
        }
print(json.dumps(res, indent=1))
if len(sys.argv) > 3:
    json.dump(res, open(sys.argv[3], "w"), indent=1)
