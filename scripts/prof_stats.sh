#!/bin/bash
# rocprofv3 kernel stats of the default bench (fp32 split, eager launches) and of the fp16 leg -> gpurun_out/{f32,fp16}_stats.csv
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_f32 $R/gpurun_out/prof_fp16
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f32 -o bench -- python $R/bench.py --steps 31 --warmup 3 --cpu-frames 0 --no-secondary --no-graph > $R/gpurun_out/prof_f32.log 2>&1); echo "prof f32 rc=$?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fp16 -o bench -- python $R/bench.py --steps 31 --warmup 3 --cpu-frames 0 --no-secondary --no-graph --dtype fp16 > $R/gpurun_out/prof_fp16.log 2>&1); echo "prof fp16 rc=$?"
grep -h "^{" $R/gpurun_out/prof_f32.log | cut -c1-200; grep -h "^{" $R/gpurun_out/prof_fp16.log | cut -c1-200
for m in f32 fp16; do
  db=$(find $R/gpurun_out/prof_$m -name "*.db" | head -1)
  # 31 timed + 3 warm-up + 2 timeline frames (one eager warm-up + the bracketed one)
  [ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $R/gpurun_out/${m}_stats.csv 36
  find $R/gpurun_out/prof_$m -name "*.db" -delete
done
