#!/bin/bash
# where the 16-bit 128-row halo kernel's time goes at one tile wave (B = 1) and eight (B = 8): SGAM_HABLATE builds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in 0 1 2 4 8 16 24 32; do
  mkdir -p /tmp/ab/h$v
  SGAM_HABLATE=$v SGAM_LIB_DIR=/tmp/ab/h$v python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"
done
for v in 0 1 2 4 8 16 24 32; do
  export SGAM_HIP_LIB=/tmp/ab/h$v/libsgam_hip.so
  echo -n "HABLATE=$v: "
  python scripts/shape_time.py "bfloat16|B1|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 2>/dev/null | grep plan | awk '{printf "B1 %s us  ", $4}'
  python scripts/shape_time.py "bfloat16|B8|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 2>/dev/null | grep plan | awk '{printf "B8 %s us  ", $4}'
  python scripts/shape_time.py "bfloat16|B8|256x256x128|256x256|N128|k3x3s1u0" 256,128,1 2>/dev/null | grep plan | awk '{printf "B8/256-row %s us\n", $4}'
done
