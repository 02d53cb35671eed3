#!/bin/bash
# round 5, call 33: the in-frame plan search again on the final launch structure (f32, bf16)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python scripts/plan_search.py f32 16 0.004 2>&1 | grep -v amdgpu.ids | tail -15
timeout 1500 python scripts/plan_search.py bf16 16 0.004 2>&1 | grep -v amdgpu.ids | tail -15
