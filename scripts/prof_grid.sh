#!/bin/bash
# per-(kernel, grid) durations of the conv kernels over a bench run (eager launches), via rocprofv3 kernel trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_grid
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_grid -o bench -- python $R/bench.py --steps 31 --warmup 3 --cpu-frames 0 --no-secondary --no-graph --no-roofline $BENCH_ARGS > $R/gpurun_out/prof_grid.log 2>&1)
python - <<'PY'
import sqlite3,glob,os
db=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_grid/*.db")[0]
c=sqlite3.connect(db)
q="""select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, count(*), avg(d.end-d.start), min(d.end-d.start), sum(d.end-d.start)
from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by 1,2,3,4 order by 8 desc"""
tot=0
rows=list(c.execute(q))
for r in rows: tot+=r[7]
print(f"total kernel ms/frame {tot/34/1e6:.3f}")
for r in rows[:int(os.environ.get("TOPN","40"))]:
    n=r[0]
    for a,b in (("_ZN12_GLOBAL__N_1",""),("conv_gemm_",""),("EEvNS_7XParamsE.kd",""),("ELb",",")): n=n.replace(a,b)
    print(f"{n[:40]:40s} grid=({r[1]//256},{r[2]},{r[3]}) n/frame={r[4]/34:5.1f} avg={r[5]/1e3:7.1f}us min={r[6]/1e3:6.1f} ms/frame={r[7]/34/1e6:.3f}")
PY
rm -rf $R/gpurun_out/prof_grid
