#!/bin/bash
# A/B of the fused GroupNorm + q|k|v GEMM (SGAM_FUSE_NORM_QKV=1, default) against normalise pass + generic GEMM (=0), same box
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in 1 0; do
    echo "== SGAM_FUSE_NORM_QKV=$v (rep $rep)"
    SGAM_FUSE_NORM_QKV=$v python bench.py --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline 2>/dev/null | cut -c60-110
  done
done
