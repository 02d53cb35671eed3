#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for m in bf16 f32; do timeout 1500 python scripts/plan_search.py $m 2>&1 | grep -v "amdgpu.ids\|Working with" | tail -40; done
