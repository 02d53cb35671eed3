#!/bin/bash
# round 5, call 36: the 16-bit flash kernel with 256-query workgroups (eight wavefronts on ONE K / V stream, SGAM_ATTN_H8=1): tests, kernel time, frames
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
SGAM_ATTN_H8=1 timeout 900 python -m pytest tests/test_gpu_h16.py tests/test_gpu_configs.py tests/test_gpu_ops.py -q -k "attention or attn or front_end or 16bit" 2>&1 | tail -4
fr() { n=$1; m=$2; shift; shift; echo -n "$n $m: "; env "$@" timeout 300 python scripts/h16_frame.py $m 2>&1 | tail -9 | head -${HEADN:-1}; }
for r in 1 2 3; do
for m in bf16 fp16; do
fr h4 $m SGAM_ATTN_H8=0
fr h8 $m SGAM_ATTN_H8=1
done; done
for v in 0 1; do SGAM_ATTN_H8=$v timeout 300 python scripts/frame_timeline.py bf16 1 2>&1 | grep -i "attn_flash" | head -2; done
