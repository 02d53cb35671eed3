#!/bin/bash
# round 5, call 4: the whole-K-panel GEMM after its latency fix (all panel loads in flight, weights four k-steps ahead): tests, the
# fused q|k|v projection in the f32 frame, then the panel kernel as the kernel of the other 1x1 convolutions (SGAM_PANEL_GEMM=1) for
# three size thresholds; deeper prefetch of the generic kernel's 64 x 64 tile (SGAM_XPF_SMALL variant builds)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/ablib
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm or panel or attn or qkv or conv1x1 or 1x1" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_vqgan.py -x -q -k "parity" 2>&1 | tail -3
fr() { echo -n "$1: "; shift; env "$@" timeout 300 python scripts/h16_frame.py f32 2>&1 | tail -9 | head -1; }
MAIN=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib/libsgam_hip.so
for r in 1 2; do
fr ref SGAM_HIP_LIB=$A/ref/libsgam_hip.so
fr main SGAM_HIP_LIB=$MAIN
fr panel64 SGAM_HIP_LIB=$MAIN SGAM_PANEL_GEMM=1
fr panel16 SGAM_HIP_LIB=$MAIN SGAM_PANEL_GEMM=1 SGAM_PANEL_MIN_WGS=16
fr panel1 SGAM_HIP_LIB=$MAIN SGAM_PANEL_GEMM=1 SGAM_PANEL_MIN_WGS=1
fr xpf3 SGAM_HIP_LIB=$A/xpf3/libsgam_hip.so
fr xpf4 SGAM_HIP_LIB=$A/xpf4/libsgam_hip.so
done
SGAM_PANEL_GEMM=1 SGAM_PANEL_MIN_WGS=1 timeout 300 python scripts/frame_timeline.py f32 1 > gpurun_out/r05d_timeline_f32_panel1.txt 2>&1; grep -c . gpurun_out/r05d_timeline_f32_panel1.txt
timeout 300 python scripts/frame_timeline.py f32 1 > gpurun_out/r05d_timeline_f32_main.txt 2>&1; head -4 gpurun_out/r05d_timeline_f32_main.txt | tail -2
