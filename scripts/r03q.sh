#!/bin/bash
# 16-bit halo kernel epilogue: direct 16-byte stores from the accumulator layout (SGAM_HDIRECT=1, default) vs LDS transpose + row stores, at B = 1 / 8
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in 1 0; do mkdir -p /tmp/ab/d$v; SGAM_HDIRECT=$v SGAM_LIB_DIR=/tmp/ab/d$v python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"; done
for rep in 1 2; do for v in 1 0; do
  export SGAM_HIP_LIB=/tmp/ab/d$v/libsgam_hip.so
  echo -n "HDIRECT=$v: "
  for k in "bfloat16|B1|256x256x128|256x256|N128|k3x3s1u0" "bfloat16|B8|256x256x128|256x256|N128|k3x3s1u0" "bfloat16|B8|128x128x128|128x128|N128|k3x3s1u0" "bfloat16|B8|64x64x256|64x64|N256|k3x3s1u0"; do
    python scripts/shape_time.py "$k" 128,128,1 2>/dev/null | grep plan | awk '{printf "%s us  ", $4}'
  done; echo
done; done
