#!/bin/bash
# rocprofv3 kernel stats of the default bench (fp32 split) and of the fp16 leg, then the PMC passes.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_f32 $R/gpurun_out/prof_fp16
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f32 -o bench -- python $R/bench.py --steps 31 --warmup 3 --cpu-frames 0 --no-secondary --no-graph > $R/gpurun_out/prof_f32.log 2>&1); echo "prof f32 rc=$?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fp16 -o bench -- python $R/bench.py --steps 31 --warmup 3 --cpu-frames 0 --no-secondary --no-graph --dtype fp16 > $R/gpurun_out/prof_fp16.log 2>&1); echo "prof fp16 rc=$?"
tail -1 $R/gpurun_out/prof_f32.log; tail -1 $R/gpurun_out/prof_fp16.log
for m in f32 fp16; do
  db=$(find $R/gpurun_out/prof_$m -name "*.db" | head -1)
  [ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $R/gpurun_out/${m}_stats.csv 34
  find $R/gpurun_out/prof_$m -name "*.db" -size +30M -delete
done
bash $R/scripts/pmc_conv.sh > $R/gpurun_out/pmc.log 2>&1; tail -12 $R/gpurun_out/pmc.log
timeout 200 python $R/scripts/conv_micro.py --shape 1,128,128,256,256,3 --reps 50
