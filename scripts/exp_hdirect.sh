#!/bin/bash
# same-box A/B of the 16-bit halo kernel's epilogue: transposed product + direct stores (SGAM_HDIRECT=1) vs LDS transpose (0)
cd $GRAFT_REPO_ROOT
for a in 1 0; do
  mkdir -p /tmp/ab/d$a
  SGAM_HDIRECT=$a SGAM_LIB_DIR=/tmp/ab/d$a python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"
done
for a in 1 0 1 0; do
  echo "== HDIRECT=$a"
  export SGAM_HIP_LIB=/tmp/ab/d$a/libsgam_hip.so
  python scripts/shape_time.py "float16|B1|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 256,128,1 2>/dev/null | grep plan
  python scripts/shape_time.py "float16|B1|64x64x256|64x64|N256|k3x3s1u0" 64,128,1 2>/dev/null | grep plan
  python scripts/shape_time.py "float16|B1|128x128x128|128x128|N128|k3x3s1u0" 64,128,1 128,128,1 2>/dev/null | grep plan
  for dt in fp16 bf16; do python bench.py --dtype $dt --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline 2>/dev/null | cut -c60-110; done
done
