#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the per-kernel table that
`rocprofv3 --kernel-trace --stats` prints: calls, total / average / min / max duration, share of GPU time.

    python scripts/rocprof_summary.py gpurun_out/prof/bench_results.db profiles/r01_x.csv [frames]
"""
import csv
import sqlite3
import sys


def main(db, out_csv, frames=None):
    c = sqlite3.connect(db)
    q = """select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start),
                  max(d.end - d.start)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    # (the 40 ms spin_kernel is bench.py parking the queue for its in-frame event brackets: not part of a frame)
    rows = [r for r in c.execute(q) if "spin_kernel" not in r[0]]
    total = sum(r[2] for r in rows)
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"] +
                   (["CallsPerFrame", "MsPerFrame"] if frames else []))
        for r in rows:
            extra = [round(r[1] / frames, 2), round(r[2] / frames / 1e6, 4)] if frames else []
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / total, 2)] + extra)
    print(f"{len(rows)} kernels, total GPU kernel time {total / 1e6:.2f} ms" +
          (f" = {total / frames / 1e6:.3f} ms/frame over {frames} frames" if frames else ""))
    for r in rows[:12]:
        print(f"  {r[0][:70]:70s} n={r[1]:6d} avg={r[3] / 1e3:8.1f}us {100.0 * r[2] / total:5.1f}%")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None)
