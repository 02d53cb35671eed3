#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for h in 0 1; do echo "== SGAM_HPC=$h"; SGAM_HPC=$h timeout 300 python scripts/h16_frame.py bf16 2>&1 | tail -9 | head -4; done
for h in 0 1; do echo "== SGAM_HPC=$h S=8"; SGAM_HPC=$h timeout 300 python scripts/h16_frame.py bf16 20 8 2>&1 | tail -9 | head -4; done
