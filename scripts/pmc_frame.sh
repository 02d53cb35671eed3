#!/bin/bash
# SQ counters of every kernel of the eager frame (one --pmc pass per group), averaged per kernel name.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcf
rm -rf $OUT; mkdir -p $OUT
cd /tmp
run() { name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-secondary --no-roofline --no-graph > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv,glob,os,collections
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmcf"
for f in sorted(glob.glob(out+"/*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows=[]
    for k,v in agg.items():
        m={c:sum(x)/len(x) for c,x in v.items()}
        n=len(next(iter(v.values())))
        rows.append((n*m.get("GRBM_GUI_ACTIVE",0),k,n,m))
    for tot,k,n,m in sorted(rows,reverse=True)[:12]:
        wc=m.get("SQ_WAVE_CYCLES",1); g=m.get("GRBM_GUI_ACTIVE",1)
        print(f"{k[:58]:58s} n={n:4d} gui={g:9.0f} mfma_busy={m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/4/256/(g/8):5.2f} wait_any/wave={m.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst/wave={m.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} wait_lds/wave={m.get('SQ_WAIT_INST_LDS',0)/wc:5.2f} waves={m.get('SQ_WAVES',0):6.0f}")
PY
