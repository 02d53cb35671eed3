#!/bin/bash
# halo staging of the 128-row split-fp32 kernel: loads vs arithmetic (SGAM_XABLATE 28 / 29), B = 1 and B = 8
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in 0 25 28 29 21; do
  mkdir -p /tmp/ab/x$v
  SGAM_XABLATE=$v SGAM_LIB_DIR=/tmp/ab/x$v python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"
done
for rep in 1 2; do for v in 0 25 28 29 21; do
  export SGAM_HIP_LIB=/tmp/ab/x$v/libsgam_hip.so
  echo -n "XABLATE=$v: "
  python scripts/shape_time.py "f32x|B1|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 2>/dev/null | grep plan | awk '{printf "B1 %s us  ", $4}'
  python scripts/shape_time.py "f32x|B8|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 2>/dev/null | grep plan | awk '{printf "B8 %s us\n", $4}'
done; done
