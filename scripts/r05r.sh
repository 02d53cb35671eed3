#!/bin/bash
# round 5, call 18: the split-fp32 AttnBlock in three launches (fused front end writes K / V^T in fragment order): bit identity, attention +
# full-model parity tests, f32 frames with the switch off and on
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "three_launches" 2>&1 | tail -25
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vqgan.py tests/test_gpu_lockstep.py -q -k "attention or attn or full_model or lockstep" 2>&1 | tail -6
fr() { n=$1; m=$2; shift; shift; echo -n "$n $m: "; env "$@" timeout 300 python scripts/h16_frame.py $m 2>&1 | tail -9 | head -${HEADN:-1}; }
for r in 1 2 3; do
fr sep f32 SGAM_ATTN_BLOCK_F32X=0
fr fused f32 SGAM_ATTN_BLOCK_F32X=1
done
timeout 300 python scripts/frame_timeline.py f32 1 2>&1 | grep -i "attn\|gemm_gn\|launches" | head -12
