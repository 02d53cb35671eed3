#!/bin/bash
# where the 128-row split-fp32 halo kernel's time goes, one tile wave (B = 1) vs eight (B = 8): SGAM_XABLATE builds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in 0 21 25 24 27 26 1; do
  mkdir -p /tmp/ab/x$v
  SGAM_XABLATE=$v SGAM_LIB_DIR=/tmp/ab/x$v python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"
done
for v in 0 21 25 24 27 26 1; do
  export SGAM_HIP_LIB=/tmp/ab/x$v/libsgam_hip.so
  echo "== XABLATE=$v"
  python scripts/shape_time.py "f32x|B1|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 2>/dev/null | grep plan
  python scripts/shape_time.py "f32x|B8|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 2>/dev/null | grep plan
  python scripts/shape_time.py "f32x|B8|128x128x128|128x128|N128|k3x3s1u0" 128,128,1 64,128,1 2>/dev/null | grep plan
done
