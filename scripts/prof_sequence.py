#!/usr/bin/env python
"""In-frame launch sequence from a rocprofv3 rocpd database: dispatches between two encode_head_kernel launches are one
VQGAN forward; positions are averaged over the frames that have the modal launch count.
    python scripts/prof_sequence.py bench_results.db"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute(f"""select s.kernel_name, d.grid_size_x / d.workgroup_size_x, d.start, d.end
                          from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"""))
frames, cur = [], None
for r in rows:
    if "encode_head_kernel" in r[0]:
        if cur:
            frames.append(cur)
        cur = []
    if cur is not None:
        cur.append(r)
mode = collections.Counter(len(f) for f in frames).most_common(1)[0][0]
frames = [f for f in frames if len(f) == mode]
print(f"{len(frames)} frames of {mode} launches")
print(f"{'#':>3s} {'kernel':52s} {'wgs':>6s} {'avg_us':>7s} {'gap_us':>7s}")
tot = gaps = 0.0
for i in range(mode):
    name = frames[0][i][0].replace("_ZN12_GLOBAL__N_1", "").replace("NS_7XParamsE", "")[:52]
    d = sum(f[i][3] - f[i][2] for f in frames) / len(frames) / 1e3
    g = sum((f[i][2] - f[i - 1][3]) for f in frames) / len(frames) / 1e3 if i else 0.0
    tot += d
    gaps += max(g, 0.0)
    print(f"{i:3d} {name:52s} {frames[0][i][1]:6d} {d:7.1f} {g:7.1f}")
print(f"kernel time {tot / 1e3:.3f} ms, positive gaps {gaps / 1e3:.3f} ms")
