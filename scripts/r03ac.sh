#!/bin/bash
# 16-bit 128-row halo tile, GroupNorm variants, at three workgroups per CU (168 registers + 36 - 84 bytes of scratch) under the 1 x 4 layout
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in 2 3; do mkdir -p /tmp/ab/o$v; SGAM_HGN_OCC=$v SGAM_LIB_DIR=/tmp/ab/o$v python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"; done
for rep in 1 2; do for v in 2 3; do
  export SGAM_HIP_LIB=/tmp/ab/o$v/libsgam_hip.so
  echo "== OCC=$v"
  for sh in 1,128,128,256,256,3 8,128,128,256,256,3 8,128,128,128,128,3; do python scripts/conv_micro.py --shape $sh --reps 20 --norm --dtype bf16 2>/dev/null | grep shape | cut -c1-110; done
  python bench.py --dtype bf16 --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline 2>/dev/null | cut -c60-110
done; done
