#!/bin/bash
# PMC passes for the dominant conv kernel (separate passes: SQ counters; FETCH_SIZE; WRITE_SIZE), kernel-trace only.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${PMC_DIR:-pmc}
rm -rf $OUT; mkdir -p $OUT
cd /tmp
SHAPE=${SHAPE:-1,128,128,256,256,3}
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- ${CMD:-python $GRAFT_REPO_ROOT/scripts/conv_micro.py --shape $SHAPE --reps 10 $MICRO_ARGS} > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
run sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_ANY
run sq4 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_LDS
find $OUT -name "*.csv" | head -20
python3 - <<'PY'
import csv,glob,os,collections
out=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/"+os.environ.get("PMC_DIR","pmc")
for f in sorted(glob.glob(out+"/*/**/*counter_collection.csv", recursive=True)):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==",f.split("/pmc/")[1])
    for k,v in agg.items():
        if "conv_gemm" in k or "conv3x3" in k:
            print(" ",k, {c:(sum(x)/len(x)) for c,x in v.items()}, "n=",len(next(iter(v.values()))))
PY
