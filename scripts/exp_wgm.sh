#!/bin/bash
# halo kernel: wavefront layout 1x4 vs 2x2 on the frame's main 3x3 shapes (rebuilds conv_f32x.hip on the box)
cd $GRAFT_REPO_ROOT
for wgm in 1 2; do
  SGAM_XWGM=$wgm python -m sgam_neurips22_amd.build 2>&1 | grep -E "error|warning: v"
  echo "== SGAM_XWGM=$wgm"
  python scripts/shape_time.py "f32x|B1|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 64,128,1 | grep plan
  python scripts/shape_time.py "f32x|B1|128x128x128|128x128|N128|k3x3s1u0" 64,128,1 128,128,1 | grep plan
  python scripts/shape_time.py "f32x|B1|64x64x256|64x64|N256|k3x3s1u0" 64,128,2 64,128,1 | grep plan
  python scripts/shape_time.py "f32x|B1|32x32x256|32x32|N256|k3x3s1u0" 64,128,8 64,128,4 | grep plan
  python scripts/shape_time.py "f32x|B1|16x16x512|16x16|N512|k3x3s1u0" 64,128,16 64,128,8 | grep plan
done
python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"
