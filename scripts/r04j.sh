#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for f in 0 1 2 3; do for c in 1 2; do echo "== DBGF=$f case $c"; PC_CASES=$c SGAM_HPC=1 SGAM_HPC_DBG=1 SGAM_HPC_DBGF=$f timeout 200 python scripts/h16_pc_check.py dump 2>&1 | grep HPC_DBG | tail -2 | cut -c1-700; done; done
