#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 400 python scripts/h16_pc_check.py 2>&1 | grep -v amdgpu.ids | tail -14 | cut -c1-260
