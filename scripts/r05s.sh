#!/bin/bash
# round 5, call 19: the whole 16-bit AttnBlock in four launches (merge of the key ranges fused into proj_out + residual): parity, suites,
# bf16 / fp16 frames with the projection fusion off and on; the split-fp32 three-launch block's bit-identity test again
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_h16.py -q -k "three_launches or front_end" 2>&1 | tail -25
timeout 1500 python -m pytest tests/test_gpu_h16.py tests/test_gpu_configs.py tests/test_gpu_vqgan.py tests/test_gpu_lockstep.py -q -k "16bit or h16 or bf16 or fp16 or lockstep" 2>&1 | tail -6
fr() { n=$1; m=$2; shift; shift; echo -n "$n $m: "; env "$@" timeout 300 python scripts/h16_frame.py $m 2>&1 | tail -9 | head -${HEADN:-1}; }
for r in 1 2 3; do
for m in bf16 fp16; do
fr front $m SGAM_ATTN_BLOCK_H16_PROJ=0
fr whole $m SGAM_ATTN_BLOCK_H16_PROJ=1
done; done
timeout 300 python scripts/frame_timeline.py bf16 1 2>&1 | grep -i "attn\|gn_finalize_kernel\|launches" | head -12
