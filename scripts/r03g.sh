#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for f in 0 1; do SGAM_GN_FOLD=$f SGAM_DUMP_TIMELINE=gpurun_out/tl_fold$f.tsv python bench.py --steps 20 --warmup 3 --no-secondary --cpu-frames 0 2>/dev/null | cut -c1-100; done
python - <<'PY'
import collections
for f in (0,1):
    agg=collections.defaultdict(lambda:[0,0.0])
    for l in open(f"gpurun_out/tl_fold{f}.tsv"):
        n,us,gf,shp=l.rstrip("\n").split("\t"); k=(n,shp) if ("halo2_kernel<64" in n or "reduce" in n or "finalize" in n) else (n,"")
        agg[k][0]+=1; agg[k][1]+=float(us)
    print("== fold",f)
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:24]: print(f"  {k[0][-52:]:52s} {k[1]:20s} n={v[0]:3d} avg={v[1]/v[0]:6.2f} tot={v[1]:7.1f}")
PY
