#!/bin/bash
# round 5, call 24: small attention kernels with the K ring + up-front V gathers: tests (f32x, 16-bit), f32 / bf16 frames with the chain
# and the one-launch kernel, per-launch time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_h16.py -q -k "small_attention" 2>&1 | tail -12
timeout 1200 python -m pytest tests/test_gpu_vqgan.py tests/test_gpu_h16.py tests/test_gpu_lockstep.py tests/test_gpu_configs.py -q -k "full_model or attention or lockstep or 16bit or h16_vs" 2>&1 | tail -6
fr() { n=$1; m=$2; shift; shift; echo -n "$n $m: "; env "$@" timeout 300 python scripts/h16_frame.py $m 2>&1 | tail -9 | head -${HEADN:-1}; }
for r in 1 2 3; do
for m in f32 bf16; do
fr chain $m SGAM_ATTN_SMALL=0
fr small $m SGAM_ATTN_SMALL=1
done; done
timeout 300 python scripts/frame_timeline.py f32 1 2>&1 | grep -i "attn_small\|launches" | head -4
timeout 300 python scripts/frame_timeline.py bf16 1 2>&1 | grep -i "attn_small\|launches" | head -4
