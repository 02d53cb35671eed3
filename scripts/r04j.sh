#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
PC_CASES=1 SGAM_HPC=1 SGAM_HPC_DBG=1 timeout 200 python scripts/h16_pc_check.py dump 2>&1 | grep HPC_DBG | tail -2
timeout 300 python scripts/h16_pc_check.py 2>&1 | tail -3
for v in 0 1; do echo "== HPC=$v"; SGAM_HPC=$v timeout 300 python scripts/h16_frame.py bf16 40 2>/dev/null | head -3; SGAM_HPC=$v timeout 300 python scripts/h16_frame.py bf16 12 8 2>/dev/null | head -3;  SGAM_HPC=$v timeout 300 python scripts/h16_frame.py fp16 12 8 2>/dev/null | head -2; done
