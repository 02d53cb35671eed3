#!/bin/bash
# rocprofv3 kernel stats of the rgbd_integration branch -> gpurun_out/rgbd_stats.csv
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/scripts/rgbd_step_breakdown.py > $R/gpurun_out/rgbd_breakdown.log 2>&1; cat $R/gpurun_out/rgbd_breakdown.log | tail -12
python $R/scripts/rgbd_loop.py 2>&1 | tail -1
rm -rf $R/gpurun_out/prof_rgbd
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_rgbd -o bench -- python $R/scripts/rgbd_loop.py > $R/gpurun_out/prof_rgbd.log 2>&1); echo "prof rgbd rc=$?"
tail -1 $R/gpurun_out/prof_rgbd.log
db=$(find $R/gpurun_out/prof_rgbd -name "*.db" | head -1)
[ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $R/gpurun_out/rgbd_stats.csv 34
find $R/gpurun_out/prof_rgbd -name "*.db" -delete
head -30 $R/gpurun_out/rgbd_stats.csv
