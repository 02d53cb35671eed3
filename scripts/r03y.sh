#!/bin/bash
# re-tune the 16-bit plans under the 1 x 4 wavefront layout, then old plans against new on the same box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m sgam_neurips22_amd.tune --dtypes fp16,bf16 --merge --configs 256x1,256x4,256x8,512x4 --out gpurun_out/plans_h16_1x4.json > gpurun_out/tune_h16_1x4.log 2>&1
tail -3 gpurun_out/tune_h16_1x4.log
for rep in 1 2; do for pf in sgam_neurips22_amd/tuned_plans_gfx950.json gpurun_out/plans_h16_1x4.json; do
  echo "== $pf"
  for dt in bf16 fp16; do SGAM_PLAN_FILE=$GRAFT_REPO_ROOT/$pf python bench.py --dtype $dt --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline 2>/dev/null | cut -c60-110; done
done; done
