#!/bin/bash
# A/B of the whole-K-panel GEMM behind the 1x1 convs (SGAM_PANEL_GEMM=1, default) against the generic kernel (=0), same box
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in 1 0; do
    echo "== SGAM_PANEL_GEMM=$v (rep $rep)"
    SGAM_PANEL_GEMM=$v python bench.py --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline 2>/dev/null | cut -c60-110
  done
done
