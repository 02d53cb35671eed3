#!/bin/bash
# where the 16-bit halo kernel's time goes: SGAM_HABLATE builds (1 no MFMAs, 2 no epilogue, 4 no main loop), same box
cd $GRAFT_REPO_ROOT
for a in ${ABL:-0 1 2 3 4 6}; do
  mkdir -p /tmp/ab/h$a
  SGAM_HABLATE=$a SGAM_LIB_DIR=/tmp/ab/h$a python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"
done
for a in ${ABL:-0 1 2 3 4 6} 0; do
  echo "== HABLATE=$a"
  export SGAM_HIP_LIB=/tmp/ab/h$a/libsgam_hip.so
  python scripts/shape_time.py "float16|B1|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 64,128,1 2>/dev/null | grep plan
  python scripts/shape_time.py "float16|B1|64x64x256|64x64|N256|k3x3s1u0" 64,128,1 2>/dev/null | grep plan
done
