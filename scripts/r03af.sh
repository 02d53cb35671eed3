#!/bin/bash
# halo kernel on a 64-channel tile (2 x 2 wavefronts, whole-K workgroups) for the 64 x 64 maps against the (64, 128) split-K plans
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "tile_plans or 64_channel_tile" 2>&1 | tail -3
for sh in "f32x|B1|64x64x256|64x64|N256|k3x3s1u0" "f32x|B1|64x64x128|64x64|N256|k3x3s1u0" "f32x|B1|32x32x256|32x32|N256|k3x3s1u0" "f32x|B1|128x128x128|128x128|N128|k3x3s1u0" "f32x|B1|16x16x512|16x16|N512|k3x3s1u0"; do
  python scripts/shape_time.py "$sh" 64,128,1 64,128,2 64,128,4 64,64,1 64,64,2 64,64,4 64,64,8 2>/dev/null | grep plan
  python scripts/cold_time.py "$sh" 64,128,2 64,64,1 64,64,2 2>/dev/null | grep -i "plan\|cold" | head -6
done
