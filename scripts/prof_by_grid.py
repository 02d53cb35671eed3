#!/usr/bin/env python
"""Per-(kernel, grid) durations from a rocprofv3 rocpd database: tells the layers of one kernel instantiation apart.
    python scripts/prof_by_grid.py gpurun_out/prof_f32/.../bench_results.db [frames]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute(f"pragma table_info({disp})")]
gx = "grid_size_x" if "grid_size_x" in cols else [x for x in cols if "grid" in x][0]
wx = "workgroup_size_x" if "workgroup_size_x" in cols else [x for x in cols if "workgroup" in x][0]
q = f"""select s.kernel_name, d.{gx}, d.{wx}, count(*), avg(d.end - d.start), min(d.end - d.start), sum(d.end - d.start)
        from {disp} d join {sym} s on d.kernel_id = s.id group by 1, 2, 3 order by 7 desc"""
print(f"{'kernel':60s} {'wgs':>7s} {'n/frame':>8s} {'avg_us':>8s} {'min_us':>8s} {'ms/frame':>9s}")
for name, g, w, n, avg, mn, tot in c.execute(q):
    if "spin_kernel" in name:
        continue
    short = name.replace("_ZN12_GLOBAL__N_1", "").replace("NS_7XParamsE", "")[:60]
    print(f"{short:60s} {g // max(w, 1):7d} {n / frames:8.2f} {avg / 1e3:8.1f} {mn / 1e3:8.1f} {tot / frames / 1e6:9.4f}")
