#!/bin/bash
# round 5, call 22: the whole-K-panel GEMM as the kernel of the other 1 x 1 convolutions, now that proj_out (the shape it lost) is fused
# into the attention's merge: SGAM_PANEL_GEMM=1 with the minimum grid at 64 / 16 / 1 tiles, f32 frames
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
fr() { n=$1; m=$2; shift; shift; echo -n "$n $m: "; env "$@" timeout 300 python scripts/h16_frame.py $m 2>&1 | tail -9 | head -${HEADN:-1}; }
for r in 1 2 3; do
fr off f32 SGAM_PANEL_GEMM=0
fr p64 f32 SGAM_PANEL_GEMM=1 SGAM_PANEL_MIN_WGS=64
fr p16 f32 SGAM_PANEL_GEMM=1 SGAM_PANEL_MIN_WGS=16
fr p1 f32 SGAM_PANEL_GEMM=1 SGAM_PANEL_MIN_WGS=1
done
SGAM_PANEL_GEMM=1 SGAM_PANEL_MIN_WGS=1 timeout 600 python -m pytest tests/test_gpu_vqgan.py -q -k "full_model" 2>&1 | tail -3
