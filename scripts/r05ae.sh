#!/bin/bash
# round 5, call 34: the two plan changes the in-frame search found for the 16-bit modes, A / B x 3 + what the 128^2 layer runs on + agreement tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
fr() { n=$1; m=$2; shift; shift; echo -n "$n $m: "; env "$@" timeout 300 python scripts/h16_frame.py $m 2>&1 | tail -9 | head -${HEADN:-1}; }
for r in 1 2 3; do
for m in bf16 fp16; do
fr old $m
fr planA $m SGAM_PLAN_FILE=$GRAFT_REPO_ROOT/ablib/plans_a.json
fr planAB $m SGAM_PLAN_FILE=$GRAFT_REPO_ROOT/ablib/plans_new.json
done; done
SGAM_PLAN_FILE=$GRAFT_REPO_ROOT/ablib/plans_new.json timeout 300 python scripts/frame_timeline.py bf16 1 2>&1 | grep "16384, 128, 1152\|1024, 512, 4608\|launches" | head
SGAM_PLAN_FILE=$GRAFT_REPO_ROOT/ablib/plans_new.json timeout 900 python -m pytest tests/test_gpu_h16.py tests/test_gpu_configs.py tests/test_gpu_vqgan.py -q -k "16bit or h16 or config2 or bf16 or fp16" 2>&1 | tail -4
