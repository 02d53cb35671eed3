#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -k "splitk or full_model or lockstep or hip_graph" > gpurun_out/pytest_r03h.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_r03h.log
for rep in 1 2; do for f in 0 1; do
  echo -n "GN_FOLD=$f rep $rep: "; SGAM_GN_FOLD=$f SGAM_DUMP_TIMELINE=gpurun_out/tl_fold$f.tsv python bench.py --steps 60 --warmup 5 --no-secondary --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['roofline']['kernels_per_frame'], d['roofline']['kernel_time_ms_per_frame'])"
done; done
python - <<'PY'
import collections
for f in (0,1):
    agg=collections.defaultdict(lambda:[0,0.0])
    for l in open(f"gpurun_out/tl_fold{f}.tsv"):
        n,us,gf,shp=l.rstrip("\n").split("\t")
        if not ("halo2_kernel<64" in n or "reduce" in n or "finalize" in n): continue
        agg[(n,shp)][0]+=1; agg[(n,shp)][1]+=float(us)
    print("== fold",f)
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:12]: print(f"  {k[0][-52:]:52s} {k[1]:20s} n={v[0]:3d} avg={v[1]/v[0]:6.2f} tot={v[1]:7.1f}")
PY
