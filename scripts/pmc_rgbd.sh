#!/bin/bash
# rocprofv3 --pmc passes over the rgbd_integration branch of the loop (scripts/rgbd_loop.py), kernel trace only; per-kernel
# averages of the TSDF kernels -> gpurun_out/pmc_rgbd.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_rgbd
rm -rf $OUT; mkdir -p $OUT
cd /tmp
run() { name=$1; shift
  STEPS=${STEPS:-8} WARMUP=2 timeout 500 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/scripts/rgbd_loop.py > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
python3 - <<PY
import csv, glob, collections, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"(tsdf_\w+|inverse_warp_kernel|depth_normalise\w*)", k)
        if m:
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
# (the first launches see a nearly empty volume: averages over the last half of the run)
out = {k: {c: round(sum(v[len(v) // 2:]) / len(v[len(v) // 2:]), 1) for c, v in cs.items()} | {"launches": max(len(v) for v in cs.values())} for k, cs in agg.items()}
json.dump(out, open("$R/gpurun_out/pmc_rgbd.json", "w"), indent=1)
for k, v in out.items():
    print(k, v)
PY
grep -il "error\|invalid" $OUT/*.log | head
find $OUT -name "*.csv" -size +4M -delete
