#!/bin/bash
# split-fp32 halo kernel at B = 8 (clock-throttled regime): 2 x 2 wavefront grid (half the LDS reads, twice the L1 fragment loads) against 1 x 4
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in 1 2; do mkdir -p /tmp/ab/x$v; SGAM_XWGM=$v SGAM_LIB_DIR=/tmp/ab/x$v python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"; done
for rep in 1 2; do for v in 1 2; do
  export SGAM_HIP_LIB=/tmp/ab/x$v/libsgam_hip.so
  echo "== XWGM=$v"
  python scripts/shape_time.py "f32x|B8|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 2>/dev/null | grep plan
  python scripts/shape_time.py "f32x|B8|128x128x128|128x128|N128|k3x3s1u0" 128,128,1 64,128,1 2>/dev/null | grep plan
  python scripts/shape_time.py "f32x|B1|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 2>/dev/null | grep plan
done; done
