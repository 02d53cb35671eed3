#!/bin/bash
# round 4: producer/consumer kernel — bit-identity, cycle stamps, in-frame time against the one-role kernel (B = 1 and lock step S = 8)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
SGAM_HPC=1 timeout 300 python scripts/h16_pc_check.py 2>&1 | tail -3
for c in 1; do PC_CASES=$c SGAM_HPC=1 SGAM_HPC_DBG=1 timeout 200 python scripts/h16_pc_check.py dump 2>&1 | grep HPC_DBG | tail -2 | cut -c1-900; done
for h in 0 1; do echo "== SGAM_HPC=$h"; SGAM_HPC=$h timeout 300 python scripts/h16_frame.py bf16 2>&1 | tail -9 | head -4; done
for h in 0 1; do echo "== SGAM_HPC=$h S=8"; SGAM_HPC=$h timeout 300 python scripts/h16_frame.py bf16 20 8 2>&1 | tail -9 | head -4; done
