#!/bin/bash
# plans for B = 16 (lock step of 16 scenes): tuner run, then heuristic plans against tuned on the same box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m sgam_neurips22_amd.tune --dtypes f32,fp16,bf16 --merge --configs 256x16 --out gpurun_out/plans_b16.json > gpurun_out/tune_b16.log 2>&1
tail -2 gpurun_out/tune_b16.log
for pf in sgam_neurips22_amd/tuned_plans_gfx950.json gpurun_out/plans_b16.json; do
  echo "== $pf"
  SGAM_PLAN_FILE=$GRAFT_REPO_ROOT/$pf python bench.py --steps 12 --warmup 3 --cpu-frames 0 --lockstep-scenes 16 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        for k,v in d['lockstep_scenes'].items():
            if isinstance(v,dict): print(k,v['value'],v['ms_per_round'],v['roofline']['frame']['frac'])
"
done
