#!/usr/bin/env python
"""Aggregate the rocprofv3 --pmc passes of scripts/pmc_frame.sh per KERNEL NAME over its in-frame launches.

    python scripts/pmc_frame.py <dir with sq/ fetch/ write/ pass outputs> <out.json> <frames in the run> [collected_at]

`collected_at` = "library @ <commit>" of the build the passes ran on (the GPU box has no .git: scripts/pmc_frame.sh takes it
from $SGAM_COMMIT, which the caller bakes into the gpurun command line); bench.py prints it as roofline.counters_commit.

Per kernel: launches per frame, and per-launch averages of HBM traffic = FETCH_SIZE x 2 (gfx950 correction,
MI355X_MICROARCH.md: 128-byte requests tallied at 64 B) + WRITE_SIZE (both reported in KiB), the in-kernel matrix-pipe busy
fraction SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x per-XCD GRBM_GUI_ACTIVE), the issue-stall and parked fractions of the
wave cycles, and the effective shader clock.  Launches outside the frame loop (weight packing, seed synthesis) carry other
kernel names or are a few among thousands; every kernel of the frame is listed, so a reader can recompute any line of
bench.py's roofline block from this file."""
import collections
import csv
import glob
import json
import re
import sys


def canon(name):
    """'void (anonymous namespace)::k<128, 128, true, false>(XParams)' / 'k<...>(...) [clone .kd]' -> 'k<128,128,true,false>'"""
    n = name.strip()
    n = re.sub(r"^void\s+", "", n)
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\.kd$", "", n)
    depth, out = 0, []
    for ch in n:                       # cut the parameter list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).replace(" ", "")


def load(src, group):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f"{src}/{group}/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            agg[canon(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main(src, out, frames, collected_at=None, lib_digest=None):
    sq, fe, wr = load(src, "sq"), load(src, "fetch"), load(src, "write")
    rows = {}
    for k, c in sq.items():
        n = len(c.get("GRBM_GUI_ACTIVE", []))
        if n == 0 or "spin_kernel" in k:
            continue
        mean = {cn: sum(v) / len(v) for cn, v in c.items()}
        xcd = mean["GRBM_GUI_ACTIVE"] / 8.0
        wc = mean.get("SQ_WAVE_CYCLES", 0.0)
        fetch = 1024.0 * sum(fe[k]["FETCH_SIZE"]) / len(fe[k]["FETCH_SIZE"]) if fe.get(k, {}).get("FETCH_SIZE") else None
        write = 1024.0 * sum(wr[k]["WRITE_SIZE"]) / len(wr[k]["WRITE_SIZE"]) if wr.get(k, {}).get("WRITE_SIZE") else None
        rows[k] = {"launches": n, "launches_per_frame": round(n / frames, 2),
                   "gui_cycles_per_xcd": round(xcd, 1),
                   "mfma_busy_frac": round(mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 4 / 256 / xcd, 4) if xcd else None,
                   "wait_inst_any_frac": round(mean.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4) if wc else None,
                   "wait_any_frac": round(mean.get("SQ_WAIT_ANY", 0.0) / wc, 4) if wc else None,
                   "wait_inst_lds_frac": round(mean.get("SQ_WAIT_INST_LDS", 0.0) / wc, 4) if wc else None,
                   "waves": round(mean.get("SQ_WAVES", 0.0), 1),
                   "fetch_bytes_raw": None if fetch is None else round(fetch),
                   "write_bytes": None if write is None else round(write),
                   "hbm_traffic_bytes_per_launch": None if (fetch is None or write is None) else round(2 * fetch + write)}
    order = sorted(rows, key=lambda k: -rows[k]["launches"] * rows[k]["gui_cycles_per_xcd"])
    doc = {"command": "scripts/pmc_frame.sh: rocprofv3 --kernel-trace --pmc <group> -- python bench.py --steps N --warmup 2 "
                      "--no-graph --no-secondary --no-roofline --dtype <mode>; passes: sq (SQ_* + GRBM_GUI_ACTIVE), fetch (FETCH_SIZE), "
                      "write (WRITE_SIZE)",
           "frames_in_run": frames,
           "collected_at": collected_at,
           "lib_digest": lib_digest,          # sgam_build_digest() of that build: bench.py flags the counters stale when it differs
           "notes": "per-launch averages over ALL launches of the kernel in the run (in-frame shapes mixed as the frame mixes them); "
                    "hbm_traffic = 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / "
                    "(1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); profiled passes clock ~3-5 % lower than un-profiled runs",
           "kernels": {k: rows[k] for k in order}}
    json.dump(doc, open(out, "w"), indent=1)
    for k in order[:14]:
        r = rows[k]
        print(f"{k[:64]:64s} n/frame={r['launches_per_frame']:6.2f} mfma_busy={r['mfma_busy_frac']} hbm_B={r['hbm_traffic_bytes_per_launch']} "
              f"wait_inst={r['wait_inst_any_frac']} wait_any={r['wait_any_frac']}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None, sys.argv[5] if len(sys.argv) > 5 else None)
