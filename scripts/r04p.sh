#!/bin/bash
# round 4: A-fragment read-ahead depth of the one-role 16-bit halo kernel (variant builds under ablib/fd<FD2><FD4>): layers + frame
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for shape in "64 256 64" "128 128 64" "32 256 64" "256 128 128"; do
  for v in 11 21 31 41 32; do
    L=$GRAFT_REPO_ROOT/ablib/fd$v/libsgam_hip.so
    echo -n "FD=$v  "; SGAM_HIP_LIB=$L timeout 200 python scripts/h16_layer_time.py 1 bf16 $shape 2>&1 | tail -1 | cut -c40-260
  done
done
for v in 11 31 32 41 11 31; do echo -n "FD=$v frame: "; SGAM_HIP_LIB=$GRAFT_REPO_ROOT/ablib/fd$v/libsgam_hip.so timeout 300 python scripts/h16_frame.py bf16 2>&1 | tail -9 | head -1; done
