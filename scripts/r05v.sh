#!/bin/bash
# round 5, call 23: the small AttnBlocks' attention in one launch (attn_small_f32x_kernel): tests, AttnBlock(512) and full-model parity,
# f32 frames with the chain / the one-launch kernel; the panel GEMM default
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "small_attention" 2>&1 | tail -25
timeout 1200 python -m pytest tests/test_gpu_vqgan.py tests/test_gpu_lockstep.py tests/test_gpu_configs.py -q -k "full_model or attention or lockstep or trajectory or parity" 2>&1 | tail -6
fr() { n=$1; m=$2; shift; shift; echo -n "$n $m: "; env "$@" timeout 300 python scripts/h16_frame.py $m 2>&1 | tail -9 | head -${HEADN:-1}; }
for r in 1 2 3; do
fr chain f32 SGAM_ATTN_SMALL=0
fr small f32 SGAM_ATTN_SMALL=1
done
timeout 300 python scripts/frame_timeline.py f32 1 2>&1 | grep -i "attn_small\|launches" | head
